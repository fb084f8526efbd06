#!/bin/bash
for ct in "64 48" "52 42" "52 44" "44 38" "44 36"; do set -- $ct; echo "== k=32 cap $1 target $2"; E3D_KNN_CAP1=$1 E3D_KNN_REP_TARGET=$2 E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k 32 --no-cpu --repeat 3 2>&1 | grep -E "single pass|ms_per_call" | tail -2 | cut -c1-150; done
for ct in "36 22" "36 18" "20 14" "28 16"; do set -- $ct; echo "== k=8 cap $1 target $2"; E3D_KNN_CAP1=$1 E3D_KNN_REP_TARGET=$2 E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k 8 --no-cpu --repeat 3 2>&1 | grep -E "single pass|ms_per_call" | tail -2 | cut -c1-150; done
