#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_multires.py -x -q -m gpu 2>&1 | tail -4
for sel in 3 2; do
  for k in 32 8; do
    E3D_KNN_SELECT=$sel python tools/bench_normals.py --k $k --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('sel $sel k', d['k'], 'ms', round(d['ms_per_call'],2), 'Mn/s', round(d['value']/1e6,1))"
  done
done
for f in ${FACTORS:-}; do E3D_KNN_CELL_FACTOR=$f python tools/bench_normals.py --k 32 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('factor $f k32 ms', round(d['ms_per_call'],2))"; E3D_KNN_CELL_FACTOR=$f python tools/bench_normals.py --k 8 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('factor $f k8 ms', round(d['ms_per_call'],2))"; done
