#!/bin/bash
run() { echo "== $*"; env "$@" E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k $K --no-cpu --repeat 3 2>&1 | grep -E "single pass|level 0 cell|ms_per_call" | tail -3 | cut -c1-140; }
K=32
run E3D_KNN_REP_STRIDE=16
run E3D_KNN_REP_STRIDE=16 E3D_KNN_REP_AVG=1
run E3D_KNN_REP_STRIDE=32 E3D_KNN_REP_AVG=1
run E3D_KNN_CELL_FACTOR=1.0
run E3D_KNN_CELL_FACTOR=1.15
run E3D_KNN_CELL_FACTOR=1.5
run E3D_KNN_REP_TARGET=44
run E3D_KNN_REP_TARGET=52
K=8
run E3D_KNN_REP_STRIDE=16
run E3D_KNN_REP_STRIDE=32 E3D_KNN_REP_AVG=1
run E3D_KNN_CELL_FACTOR=0.35
run E3D_KNN_CELL_FACTOR=0.6
run E3D_KNN_REP_TARGET=16
