#!/bin/bash
run() { echo "== K=$K $*"; env "$@" E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k $K --no-cpu --repeat 3 2>&1 | grep -E "single pass|level 0 cell|ms_per_call" | tail -3 | cut -c1-140; }
K=32
run E3D_KNN_CELL_FACTOR=1.15
run E3D_KNN_CELL_FACTOR=1.5
run E3D_KNN_CELL_FACTOR=1.8
K=8
run E3D_KNN_CELL_FACTOR=0.35
run E3D_KNN_CELL_FACTOR=0.55
run E3D_KNN_CELL_FACTOR=0.7
run E3D_KNN_CELL_FACTOR=0.9
run E3D_KNN_CAP1=28 E3D_KNN_REP_TARGET=17
K=16
run E3D_KNN_SINGLE=1
run E3D_KNN_SINGLE=0
