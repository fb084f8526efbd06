#!/bin/bash
O=gpurun_out/${R4TAG:-r4o}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_normals.py tests/test_gpu_multires.py tests/test_gpu_cli.py tests/test_gpu_at_size.py -x -q -m gpu -k "not c5 and not c3" > $O/pytest_normals.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_normals.txt
for k in 32 8; do
  for mode in angular uniform; do
    fl=""; [ $mode = angular ] && fl="--angular"
    E3D_KNN_STATS=1 timeout 300 python tools/bench_normals.py --k $k --no-cpu $fl --repeat 3 > $O/n_${mode}_k$k.json 2> $O/n_${mode}_k$k.err
    python -c "
import json
d=json.loads(open('$O/n_${mode}_k$k.json').read().strip().splitlines()[-1]); print('k=$k $mode', round(d['ms_per_call'],2), 'ms')
"
    grep "^\[knn\]" $O/n_${mode}_k$k.err | tail -4 | cut -c1-120
  done
done
