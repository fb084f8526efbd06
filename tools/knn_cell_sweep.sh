#!/bin/bash
# E3D_KNN_CELL_FACTOR sweep of the kNN normal estimator (tools/bench_normals.py, 20 M points, uniform and scanner-sampled): ms per call
cd $GRAFT_REPO_ROOT
O=gpurun_out/knn_cell; mkdir -p $O
for K in 32 8; do
  if [ $K = 32 ]; then FS="default 0.9 1.1 1.3 1.6 2.0 2.5"; else FS="default 0.3 0.38 0.45 0.55 0.7 0.9"; fi
  for F in $FS; do
    for MODE in "" "--angular"; do
      if [ $F = default ]; then unset E3D_KNN_CELL_FACTOR; else export E3D_KNN_CELL_FACTOR=$F; fi
      R=$(timeout 120 python tools/bench_normals.py --k $K --no-cpu --repeat 3 $MODE 2>/dev/null | tail -1)
      echo "k=$K factor=$F mode=${MODE:-uniform} $R" | cut -c1-400
    done
  done
done | tee $O/sweep.txt
