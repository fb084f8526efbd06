#!/bin/bash
# batched pair search: ICP parity tests + the multi-process test + all-pairs leg with and without batching
O=gpurun_out/${R4TAG:-r4batch}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_multiprocess.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python -m pytest tests/test_gpu_at_size.py -q -m gpu -x -k "c3 or all_pairs" > $O/pytest_c3.txt 2>&1; echo "pytest c3 rc=$?"; tail -2 $O/pytest_c3.txt
timeout 600 python bench.py --no-cpu-baseline --only allpairs > $O/bench_ap.json 2> $O/bench_ap.err; echo "bench rc=$?"
E3D_ICP_BATCH=0 timeout 600 python bench.py --no-cpu-baseline --only allpairs > $O/bench_ap_nobatch.json 2> $O/bench_ap_nobatch.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json, os
tag = os.environ.get("R4TAG", "r4batch")
for f in ("bench_ap", "bench_ap_nobatch"):
    d = json.loads(open("gpurun_out/%s/%s.json" % (tag, f)).read().strip().splitlines()[-1])
    print(f, "ms/iter %.1f settling %s steady %.1f" % (d["ms_per_iter"], d["ms_per_iter_settling"], d["ms_per_iter_steady"]))
    print("   each", " ".join("%.0f" % v for v in d["ms_per_iter_each"]))
    print("   nn  ", " ".join("%.0f" % v for v in d["nn_ms_per_iter_each"]))
d = json.loads(open("gpurun_out/%s/bench_icp.json" % tag).read().strip().splitlines()[-1])
print("terrace ms/step", d["ms_per_step"], "steady", d["ms_per_step_steady"], "partial", d.get("partial_overlap", {}).get("ms_per_step"))
PY
