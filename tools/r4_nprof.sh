#!/bin/bash
# kNN normals: section stop-watch of the two scan kernels (library built with -DE3D_KNN_PROF=1)
for k in 32 8; do echo "== k=$k"; E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k $k --no-cpu --repeat 2 2>&1 | grep -E "knn|ms_per_call" | tail -5 | cut -c1-220; done
