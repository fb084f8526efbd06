#!/bin/bash
# round-2 GPU check of the ICP path: parity tests, then the headline bench without the CPU legs
mkdir -p gpurun_out
if [ "${SKIPTESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r2_icp_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_icp_tests.log
tail -6 gpurun_out/r2_icp_tests.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reg > gpurun_out/r2_bench_quick.json 2> gpurun_out/r2_bench_quick.err
echo "bench rc=$?"
cat gpurun_out/r2_bench_quick.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['breakdown_ms_per_iter'], d['roofline']['frac'], d['lm_passes_per_iter'])"
tail -3 gpurun_out/r2_bench_quick.err
