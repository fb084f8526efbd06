#!/bin/bash
# round 5, GPU call 4: pruning key kernel (tests + trend), kNN block mapping chunk sizes on both scans, default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_icp.py -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
timeout 200 python tools/icp_trend.py 50000000 40 0 0.01 2 3.0 > $O/trend_p3.txt 2>&1; grep -E "wall ms|converged" $O/trend_p3.txt
E3D_NN_PRUNE=0 timeout 200 python tools/icp_trend.py 50000000 40 0 0.01 2 3.0 > $O/trend_p3_noprune.txt 2>&1; echo "PRUNE=0"; grep -E "wall ms|converged" $O/trend_p3_noprune.txt
timeout 200 python tools/icp_trend.py 50000000 30 1 0.01 2 1.0 > $O/trend_partial.txt 2>&1; echo "partial"; grep -E "wall ms|converged" $O/trend_partial.txt
for k in 32 8; do
  for x in 0 64; do
    E3D_KNN_XCD=$x timeout 120 python tools/bench_normals.py --k $k --no-cpu > $O/n_u_k${k}_x$x.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/n_u_k${k}_x$x.json')); print('normals uniform k=$k G=$x: %.3f ms' % d['ms_per_call'])"
  done
  for x in 0 16 64 256; do
    E3D_KNN_XCD=$x timeout 120 python tools/bench_normals.py --k $k --no-cpu --angular > $O/n_a_k${k}_x$x.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/n_a_k${k}_x$x.json')); print('normals scanner-sampled k=$k G=$x: %.3f ms' % d['ms_per_call'])"
  done
done
cd /tmp
for x in 64; do
  rm -rf /tmp/pmc_$x
  E3D_KNN_XCD=$x timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$x -o p -- python $GRAFT_REPO_ROOT/tools/bench_normals.py --k 8 --no-cpu --repeat 1 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pmc_$x/p_results.db $GRAFT_REPO_ROOT/$O/normals_k8_fetch_xcd$x.txt "" > /dev/null 2>&1
  grep -E "k_knn_hist|k_knn_normals<4>|k_permute" $GRAFT_REPO_ROOT/$O/normals_k8_fetch_xcd$x.txt | grep "FETCH_SIZE," | cut -c1-40,150-230
done
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; tail -c 3800 $O/bench_stdout.txt; cp bench_detail.json $O/ 2>/dev/null
