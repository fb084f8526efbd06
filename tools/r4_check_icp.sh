#!/bin/bash
# Round 4, first GPU pass: ICP parity tests, the ICP bench legs with resident and with compacted rows, LM loop micro-benchmark.
O=gpurun_out/${R4TAG:-r4a}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $O/pytest_icp.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_icp.txt | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
E3D_ICP_RESIDENT=0 timeout 600 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-partial > $O/bench_icp_compacted.json 2> $O/bench_icp_compacted.err; echo "bench compacted rc=$?"
timeout 300 tools/micro/lm_variants 100 2 > $O/lm_variants_2sets.txt 2>&1; echo "lm_variants rc=$?"
tail -36 $O/lm_variants_2sets.txt
python - <<'PY'
import json
for f in ("bench_icp", "bench_icp_compacted"):
    try:
        d = json.loads(open("gpurun_out/" + __import__("os").environ.get("R4TAG", "r4a") + "/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/step %.3f settling %s steady %.3f" % (d["ms_per_step"], d.get("ms_per_step_settling"), d["ms_per_step_steady"]))
    for k, v in d["roofline"]["kernels"].items():
        print("   %-20s %8.3f ms/iter  avg %s" % (k, v["summed_ms_per_iter"] or 0, v.get("avg_launch_ms")))
    p = d.get("partial_overlap")
    if p:
        print("  partial: ms/step %.3f steady %.3f" % (p["ms_per_step"], p["ms_per_step_steady"]))
        for k, v in p["roofline"]["kernels"].items():
            print("   %-20s %8.3f ms/iter" % (k, v["summed_ms_per_iter"] or 0))
PY
