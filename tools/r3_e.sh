#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 900 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -s --durations=5 > $O/at_size.txt 2>&1; tail -30 $O/at_size.txt
timeout 900 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --steps 20 --warmup 5 > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3e/bench_icp.json"))
for name,x in (("terrace",d),("partial",d.get("partial_overlap"))):
    if not x: continue
    print(name,"ms/step",round(x["ms_per_step"],2),"settling",x["ms_per_step_settling"],"steady",round(x["ms_per_step_steady"],2),"from",x["steady_from_timed_step"],"value",x["value"], "matched", x["config"]["matched_fraction"])
    print("  accounted",round(x["breakdown_ms_per_iter"]["kernels_accounted"],2),"rest",round(x["breakdown_ms_per_iter"]["host_sync_and_small_kernels"],2))
    for k,v in x["roofline"]["kernels"].items(): print("   ",k, "avg", v.get("avg_launch_ms"), "sum/iter", v.get("summed_ms_per_iter"), "GB/s", v.get("GBs"))
PY
