#!/bin/bash
# round 5, GPU call 3: CLI tests (rebuilt tools), all-pairs leg (batched kernels) with the scale model, far-list routing on the headline scene
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_switches.py -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
timeout 600 python bench.py --only allpairs > $O/allpairs.json 2> $O/allpairs.err; echo "allpairs rc=$?"; tail -3 $O/allpairs.err
python - $O/allpairs.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("allpairs: ms_per_iter %.1f settling %s steady %.1f launches %s" % (d["ms_per_iter"], d["ms_per_iter_settling"], d["ms_per_iter_steady"], d["nn_launches_per_iter"]))
print("each", [round(v, 1) for v in d["ms_per_iter_each"]])
sm = d.get("scale_model")
if sm:
    print("scale model:", {k: sm[k] for k in ("ms_per_iter_n1", "ms_per_iter_as_rank0_of_world", "modelled_speedup", "non_dividing_ms_per_iter", "poses_equal_single_gpu_run", "nn_kernel_launches_per_iter_as_rank0")}, "steady", sm["steady"])
    print("rank0 each", [round(v, 1) for v in sm["ms_per_iter_each_as_rank0"]])
    print("last as rank0", sm["last_iteration_as_rank0"]); print("last n1", sm["last_iteration_n1"])
P
E3D_ICP_BATCH=1 timeout 300 python bench.py --only allpairs --no-scale-model > $O/allpairs_batch1.json 2> /dev/null
python -c "
import json; d=json.load(open('$O/allpairs_batch1.json')); print('allpairs E3D_ICP_BATCH=1: ms_per_iter %.1f steady %.1f' % (d['ms_per_iter'], d['ms_per_iter_steady'])); print('each', [round(v,1) for v in d['ms_per_iter_each']])"
for div in 32 4 0; do
  E3D_NN_FAR_DIV=$div timeout 200 python tools/icp_trend.py 50000000 40 0 0.01 2 3.0 > $O/trend_p3_div$div.txt 2>&1
  echo "FAR_DIV=$div"; grep -E "wall ms|converged" $O/trend_p3_div$div.txt
done
