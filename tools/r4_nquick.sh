#!/bin/bash
# quick timing of e3d_normals_knn at 20 M points: NK (default "32 8"), NMODES (default uniform only), extra env passes through
for k in ${NK:-32 8}; do for mode in ${NMODES:-uniform}; do m=""; [ $mode = angular ] && m="--angular"
echo "== k=$k $mode"; E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k $k --no-cpu --repeat 3 $m 2>&1 | grep -E "single pass|level 0 cell|ms_per_call" | tail -3 | cut -c1-150; done; done
