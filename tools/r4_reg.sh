#!/bin/bash
# ImageRegistrator: parity tests + the bench leg with the kernel-group record
O=gpurun_out/${R4TAG:-r4g}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reg.py tests/test_gpu_cli_reg.py tests/test_gpu_multires.py -x -q -m gpu > $O/pytest_reg.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_reg.txt
timeout 900 python bench.py --only reg --no-cpu-baseline > $O/bench_reg.json 2> $O/bench_reg.err; echo "bench rc=$?"; tail -3 $O/bench_reg.err
python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s/bench_reg.json" % os.environ.get("R4TAG", "r4g")).read().strip().splitlines()[-1])
print("ms per run iteration %.2f  accumulate all images %.2f ms  obs refresh %.2f ms  residuals/s %.3g" % (d["ms_per_run_iteration"], d["accumulate_ms_all_images"], d["observation_refresh_ms_all_images"], d["value"]))
pp = d["run_phase_profile"]
print({k: round(v, 2) for k, v in pp["ms_total"].items()}, pp["iterations"])
for k, g in pp["kernel_groups"].items():
    print("  %-26s %7.2f ms/iter  %6.1f launches  avg %.4f ms  %8.3g units  frac %s" % (k, g["ms_per_iteration"], g["launches_per_iteration"], g["avg_launch_ms"], g["units_per_launch"], None if g["frac"] is None else round(g["frac"], 3)))
print("kernel groups sum %.2f ms/iter" % pp["kernel_groups_ms_per_iteration"])
PY
