#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 300 python tools/r3_trend.py 50000000 25 0 > $O/trend_full.txt 2>&1; tail -28 $O/trend_full.txt
timeout 300 python tools/r3_trend.py 50000000 25 1 > $O/trend_partial.txt 2>&1; tail -28 $O/trend_partial.txt
timeout 900 python bench.py --no-cpu-baseline --no-reg --no-normals --no-partial --steps 20 --warmup 5 > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f/bench_icp.json"))
print("terrace ms/step", d["ms_per_step"], "value", d["value"])
a=d.get("allpairs",{})
print("allpairs ms/iter", a.get("ms_per_iter"), a.get("rank0_ms_per_iter"), a.get("lm_passes_per_iter"))
for k,v in d["roofline"].get("kernels",{}).items(): print(k, v.get("avg_launch_ms"), v.get("summed_ms_per_iter"))
PY
