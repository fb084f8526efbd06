#!/bin/bash
# round 3, GPU call 1: LM loop shapes (micro), ICP parity with the new LM kernels, the ICP legs of the bench
O=gpurun_out/r3a; mkdir -p $O
timeout 300 ./tools/micro/lm_variants 100 8 > $O/lm_variants.txt 2>&1; echo "micro rc=$?"
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-reg --no-normals --steps 20 --warmup 5 > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_icp.json"))
print("terrace ms/step", d["ms_per_step"], "value", d["value"])
a=d.get("allpairs",{})
print("allpairs", {k:(round(v,1) if isinstance(v,float) else v) for k,v in a.items() if not isinstance(v,(dict,list))})
print(a.get("rank0_ms_per_iter"))
for k in d["roofline"].get("kernels",[]): print(k)
PY
