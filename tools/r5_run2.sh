#!/bin/bash
# round 5, GPU call 2: the batched pair kernels (tests + all-pairs leg with the recorded-reduction scale model), the as-written configs[3] / [4] tests, kNN block mapping counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_icp.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_at_size.py -x -q -s > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?"; grep -E "c5|c4|boundary|passed|failed|Error" $O/pytest_b.log | tail -12
timeout 600 python bench.py --only allpairs > $O/allpairs.json 2> $O/allpairs.err; echo "allpairs rc=$?"
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
cd /tmp
for x in 1 0; do
  rm -rf /tmp/pmc_$x
  E3D_KNN_XCD=$x timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$x -o p -- python $GRAFT_REPO_ROOT/tools/bench_normals.py --k 8 --no-cpu --repeat 1 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pmc_$x/p_results.db $GRAFT_REPO_ROOT/$O/normals_k8_fetch_xcd$x.txt "" > /dev/null 2>&1
  grep -E "k_knn_hist|k_knn_normals<4>|k_permute" $GRAFT_REPO_ROOT/$O/normals_k8_fetch_xcd$x.txt | grep "FETCH_SIZE," | cut -c1-40,150-230
done
