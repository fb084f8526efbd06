#!/bin/bash
# kernel trace of one e3d_normals_knn call at 20 M points: NMODE="" (uniform scan) or "--angular"; NK="32 8"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${R4TAG:-r4nt}; mkdir -p $O
for k in ${NK:-32 8}; do
rm -rf /tmp/nt$k
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/nt$k -o t -- python $R/tools/bench_normals.py --k $k --no-cpu $NMODE --repeat 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/nt$k/t_results.db $O/normals${NMODE}_k${k}_kernel_stats.txt "" > /dev/null 2>&1
echo "== k=$k"; grep -E "e3d::|rocprim|fillBuffer|copyBuffer" $O/normals${NMODE}_k${k}_kernel_stats.txt | head -16 | sed -E 's/\(.*\)//' | cut -c1-150
done
