#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_multires.py -x -q -m gpu 2>&1 | tail -4
run() { python tools/bench_normals.py --k $1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$2 k', d['k'], 'ms', round(d['ms_per_call'],2))"; }
run 32 base; run 8 base; run 48 base; run 60 base
E3D_KNN_DENSE_LOG2=10 run 32 hash
for f in ${FACTORS:-}; do E3D_KNN_CELL_FACTOR=$f run 32 f$f; done
