#!/bin/bash
# A/B: whole library built with -fno-slp-vectorize (lib_noslp) against the default build (only e3d_icp_kernels.hip without SLP)
O=gpurun_out/r3i; mkdir -p $O
./tools/micro/lm_variants 100 8 > $O/lm_variants.txt 2>&1
run() {  # tag
  timeout 900 python bench.py --no-cpu-baseline --no-partial --no-allpairs --steps 20 --warmup 5 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
  python - $1 <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3i/bench_%s.json"%sys.argv[1]))
print(sys.argv[1],"terrace",round(d["ms_per_step"],2),"steady",round(d["ms_per_step_steady"],2),"lm_pass",round(d["roofline"]["kernels"]["k_lm_pass"]["avg_launch_ms"],3),"multi",round(d["roofline"]["kernels"]["k_lm_cost_multi"]["avg_launch_ms"],3),"bounded",round(d["roofline"]["kernels"]["k_nn_bounded"]["avg_launch_ms"],3),"compact",round(d["roofline"]["kernels"]["k_compact_corr"]["avg_launch_ms"],3))
r=d["image_registrator"]; print("  reg accumulate ms",round(r["accumulate_ms_all_images"],2),"p1",round(r["roofline"]["k_reg_pass1"]["avg_launch_ms"],3),"p2",round(r["roofline"]["k_reg_pass2_tile32"]["avg_launch_ms"],3),"obs",round(r["observation_refresh_ms_all_images"],2),"run iter",round(r["ms_per_run_iteration"],2))
n=d["normal_estimation"]; print("  normals", {k:(round(v,3) if isinstance(v,float) else v) for k,v in n.items() if "ms" in k or k=="value"})
PY
}
run default
cp dataset-pipeline_amd/lib_noslp/libe3dhip.so dataset-pipeline_amd/lib/libe3dhip.so
run noslp
grep -E "pftrue|pf1|cost_multi" $O/lm_variants.txt | head -40
