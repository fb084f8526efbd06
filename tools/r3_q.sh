#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --steps 20 --warmup 5 > $O/bench_icp.json 2> $O/bench_icp.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3q/bench_icp.json"))
for name,x in (("terrace",d),("partial",d["partial_overlap"])):
    k=x["roofline"]["kernels"]
    print(name, round(x["ms_per_step"],2), "steady", round(x["ms_per_step_steady"],2), "certify", round(k["k_nn_certify"]["avg_launch_ms"],3), "transform/iter", round(k["k_transform_bbox"]["summed_ms_per_iter"],3), "lm", round(k["k_lm_pass"]["avg_launch_ms"],3))
PY
