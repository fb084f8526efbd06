#!/bin/bash
# kNN estimator: parity tests, then the bench scans (20 M points) with statistics, with and without the single-scan variants
O=gpurun_out/${R4TAG:-r4ns}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_normals.py tests/test_gpu_multires.py tests/test_gpu_switches.py -q -m gpu -x -k "not icp_data" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for k in 32 8; do for mode in "" "--angular"; do echo "== k=$k $mode"; E3D_KNN_STATS=1 timeout 200 python tools/bench_normals.py --k $k --no-cpu --repeat 3 $mode 2>&1 | grep -E "knn\]|ms_per_call" | tail -6 | cut -c1-200; done; done
echo "== old path"; for k in 32 8; do E3D_KNN_SINGLE=0 timeout 200 python tools/bench_normals.py --k $k --no-cpu --repeat 3 2>&1 | grep -E "ms_per_call" | cut -c1-160; done
