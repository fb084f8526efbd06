#!/bin/bash
# ICP parity tests (all modes) + the two ICP bench legs
O=gpurun_out/${R4TAG:-r4iq}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json, os
tag = os.environ.get("R4TAG", "r4iq")
d = json.loads(open("gpurun_out/%s/bench_icp.json" % tag).read().strip().splitlines()[-1])
def show(d, name):
    print(name, "ms/step %.3f settling %s steady %.3f" % (d["ms_per_step"], d.get("ms_per_step_settling"), d["ms_per_step_steady"]))
    print("   each", " ".join("%.2f" % v for v in d["ms_per_step_each"]))
    for k, v in d["roofline"]["kernels"].items():
        print("   %-20s %8.3f ms/iter  avg %s" % (k, v["summed_ms_per_iter"] or 0, v.get("avg_launch_ms")))
show(d, "terrace")
if d.get("partial_overlap"): show(d["partial_overlap"], "partial")
PY
