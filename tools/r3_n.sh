#!/bin/bash
# adaptive-reach kNN: parity tests, then a sweep of the cell factor and the block-population thresholds
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_multires.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for k in 32 8; do
  for cf in 1.3 2.6 5.2 8 12; do
    for need in "3.2,2.3,2.0" "2.5,2.0,1.8"; do
      r=$(E3D_KNN_STATS=1 E3D_KNN_CELL_FACTOR=$cf E3D_KNN_NEED=$need timeout 120 python tools/bench_normals.py --k $k --no-cpu --repeat 3 2> $O/stats_${k}_${cf}.txt | python -c "import json,sys; d=json.load(sys.stdin); print(round(d.get('ms_per_call', d.get('ms',0)),2))" 2>/dev/null)
      echo "k=$k cell_factor=$cf need=$need ms=$r  $(grep -m2 '\[knn\]' $O/stats_${k}_${cf}.txt | tr '\n' ' ')"
    done
  done
done
