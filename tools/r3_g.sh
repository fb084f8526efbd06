#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-normals --no-allpairs --no-partial --steps 5 --warmup 5 > $O/bench_reg.json 2> $O/bench_reg.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/bench_reg.json"))["image_registrator"]
print({k:v for k,v in d.items() if k not in ("roofline","run_phase_profile","config")})
print(d["run_phase_profile"]["note"])
for k,v in sorted(d["run_phase_profile"]["ms_total"].items()): print("  %-45s %8.2f ms total  %7.2f per iteration" % (k, v, v/d["run_phase_profile"]["iterations"]))
for k,v in d["roofline"].items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a not in ("note","frac_basis")})
PY
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
