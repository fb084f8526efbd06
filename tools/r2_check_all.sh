#!/bin/bash
# whole GPU suite, log to gpurun_out/r2_gpu_tests.log
mkdir -p gpurun_out
timeout ${TMO:-2400} python -m pytest tests -q -m gpu ${PYTEST_ARGS:-} > gpurun_out/r2_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r2_gpu_tests.log | head -60
