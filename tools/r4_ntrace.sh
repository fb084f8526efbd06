#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O
for k in 32 8; do
rm -rf /tmp/nt$k
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/nt$k -o t -- python $R/tools/bench_normals.py --k $k --no-cpu --angular --repeat 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/nt$k/t_results.db $O/normals_angular_k${k}_kernel_stats.txt "" > /dev/null 2>&1
echo "== k=$k"; head -14 $O/normals_angular_k${k}_kernel_stats.txt | cut -c1-50,200-260
done
