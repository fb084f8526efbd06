#!/bin/bash
# the round's closing run: whole GPU test suite, default bench line, normals on the scanner-sampled scan
O=gpurun_out/${R4TAG:-r4z}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for k in 32 8; do
  timeout 300 python tools/bench_normals.py --k $k --no-cpu > $O/normals_uniform_k$k.json 2>/dev/null
  timeout 300 python tools/bench_normals.py --k $k --no-cpu --angular > $O/normals_angular_k$k.json 2>/dev/null
  python -c "
import json
for t in ('uniform','angular'):
    d=json.loads(open('$O/normals_%s_k$k.json'%t).read().strip().splitlines()[-1]); print('normals k=$k', t, round(d['ms_per_call'],2), 'ms', d['finite_fraction'])
"
done
python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s/bench_default.json" % os.environ.get("R4TAG", "r4z")).read().strip().splitlines()[-1])
print("headline ms/step %.3f value %.4g settling %s steady %.3f frac %.3f" % (d["ms_per_step"], d["value"], d["ms_per_step_settling"], d["ms_per_step_steady"], d["roofline"]["frac"]))
a = d["allpairs"]; print("allpairs ms/iter %.1f settling %s steady %.1f value %.4g" % (a["ms_per_iter"], a["ms_per_iter_settling"], a["ms_per_iter_steady"], a["value"]))
r = d["image_registrator"]; print("reg", r["value"], r["ms_per_run_iteration"])
PY
