#!/bin/bash
# quick loop: ICP parity tests, certificate counters, timing at 2 x 50 M
mkdir -p gpurun_out
if [ "${SKIPTESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r2_icp_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_icp_tests.log
tail -5 gpurun_out/r2_icp_tests.log
fi
N=${1:-50000000}
MODE=${2:-0}
IT=${3:-25}
E3D_NN_STATS=1 E3D_PROF_INNER=150 python tools/prof_nn.py $N $MODE $IT 2>&1 | grep -v amdgpu.ids | cut -c1-420 | grep -v "1->0"
