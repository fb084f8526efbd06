#!/bin/bash
O=gpurun_out/${R4TAG:-r4m}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_multiprocess.py -x -q -m gpu > $O/pytest_icp.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_icp.txt
for bal in 1 0; do
  E3D_NN_BALANCE=$bal timeout 600 python tools/icp_trend.py 50000000 14 0 0.01 2 > $O/trend_bal$bal.txt 2>&1
  echo "== terrace balance=$bal (bounded ms (Mq))"; tail -14 $O/trend_bal$bal.txt | cut -c1-75
done
E3D_NN_BALANCE=1 timeout 600 python tools/icp_trend.py 50000000 14 1 0.01 2 > $O/trend_partial_bal1.txt 2>&1; echo "== partial balance=1"; tail -10 $O/trend_partial_bal1.txt | cut -c1-75
E3D_NN_BALANCE=0 timeout 600 python tools/icp_trend.py 50000000 14 1 0.01 2 > $O/trend_partial_bal0.txt 2>&1; echo "== partial balance=0"; tail -10 $O/trend_partial_bal0.txt | cut -c1-75
timeout 900 python bench.py --only allpairs > $O/bench_allpairs.json 2> $O/bench_allpairs.err; echo "allpairs rc=$?"
python - <<'PY'
import json, os
a = json.loads(open("gpurun_out/%s/bench_allpairs.json" % os.environ.get("R4TAG", "r4m")).read().strip().splitlines()[-1])
print("allpairs value %.4g ms/iter %.1f settling %s steady %.1f" % (a["value"], a["ms_per_iter"], a["ms_per_iter_settling"], a["ms_per_iter_steady"]))
print(" each", [round(v) for v in a["ms_per_iter_each"]])
for k, v in a["roofline"]["kernels"].items():
    print("   %-20s %8.2f ms/iter avg %s" % (k, v["summed_ms_per_iter"] or 0, v.get("avg_launch_ms")))
PY
