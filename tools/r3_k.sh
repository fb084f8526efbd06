#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py tests/test_gpu_reg.py -x -q -m gpu -k "not full_size" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2; grep -E "apart|worst deviation" $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_reg.py -q -m gpu -s -k "insensitive" 2>&1 | grep -E "apart|passed|failed"
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3k/bench_default.json"))
print("terrace", round(d["ms_per_step"],2), d["value"], "steady", round(d["ms_per_step_steady"],2), "settling", d["ms_per_step_settling"], "frac", round(d["roofline"]["frac"],3), d["roofline"]["kernel"][:20])
print("  bounded", d["roofline"]["kernels"]["k_nn_bounded"]["avg_launch_ms"], "speedup", d.get("speedup_vs_cpu_iteration_rate"))
p=d["partial_overlap"]; print("partial", round(p["ms_per_step"],2), p["value"], round(p["ms_per_step_steady"],2))
a=d["allpairs"]; print("allpairs", round(a["ms_per_iter"],1), a["value"], a["rank0_ms_per_iter"])
r=d["image_registrator"]; print("reg", r["value"], r["accumulate_ms_all_images"], r["ms_per_run_iteration"], r.get("speedup_vs_cpu"))
n=d["normal_estimation"]; print("normals k32", n["k32"]["ms_per_call"], "k8", n["k8"]["ms_per_call"])
PY
