#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-32}
rocprofv3 --kernel-trace --stats -d /tmp/pn_s -o p -- python $R/tools/bench_normals.py --k $K --no-cpu --repeat 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pn_s/p_results.db $R/gpurun_out/r2_normals_k${K}_kernel_stats.txt "" > /dev/null
head -16 $R/gpurun_out/r2_normals_k${K}_kernel_stats.txt | awk -F', ' '{print substr($1,1,70), $(NF-4), $(NF-3), $(NF-2)}'
