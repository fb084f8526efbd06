#!/bin/bash
# kNN normals: kernel trace of one call (uniform scan, k = 32 and 8) and a sweep of the starting cell size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${R4TAG:-r4ncell}; mkdir -p $O
for k in 32 8; do
rm -rf /tmp/nt$k
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/nt$k -o t -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/nt$k/t_results.db $O/normals_k${k}_kernel_stats.txt "" > /dev/null 2>&1
echo "== k=$k"; head -16 $O/normals_k${k}_kernel_stats.txt | cut -c1-60,200-270
done
cd $R
for f in 1.3 2.0 3.0 4.5; do echo "k=32 factor $f"; E3D_KNN_STATS=1 E3D_KNN_CELL_FACTOR=$f timeout 200 python tools/bench_normals.py --k 32 --no-cpu --repeat 2 2>&1 | grep -E "knn\]|ms_per_call" | tail -4 | cut -c1-200; done
for f in 0.45 0.8 1.3 2.0 3.0; do echo "k=8 factor $f"; E3D_KNN_STATS=1 E3D_KNN_CELL_FACTOR=$f timeout 200 python tools/bench_normals.py --k 8 --no-cpu --repeat 2 2>&1 | grep -E "knn\]|ms_per_call" | tail -4 | cut -c1-200; done
