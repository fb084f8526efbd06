#!/bin/bash
O=gpurun_out/${R4TAG:-r4j}; mkdir -p $O
R4TAG=${R4TAG:-r4j} bash tools/r4_reg.sh
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s/bench_default.json" % os.environ.get("R4TAG", "r4j")).read().strip().splitlines()[-1])
print("headline ms/step %.3f value %.4g settling %s steady %.3f frac %.3f (moved %.3f)" % (d["ms_per_step"], d["value"], d["ms_per_step_settling"], d["ms_per_step_steady"], d["roofline"]["frac"], d["roofline"].get("frac_of_bytes_moved") or -1))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("start"), d.get("speedup_vs_cpu_iteration_rate"))
p = d["partial_overlap"]; print("partial ms/step %.3f steady %.3f" % (p["ms_per_step"], p["ms_per_step_steady"]))
a = d["allpairs"]; print("allpairs", {k: a[k] for k in a if k in ("value", "ms_per_iter", "steps", "warmup")})
r = d["image_registrator"]; print("reg", r["value"], r["ms_per_run_iteration"])
n = d["normal_estimation"]; print("normals k32 %.2f ms k8 %.2f ms" % (n["k32"]["ms_per_call"], n["k8"]["ms_per_call"]))
PY
