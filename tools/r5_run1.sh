#!/bin/bash
# round 5, GPU call 1: the changed paths' tests, the convergence sweep of the headline scene, pass-2 variants, kNN block mapping, default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_switches.py tests/test_gpu_normals.py -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_reg.py -x -q -k "pass2 or accumul or whole_problem" > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?"; tail -3 $O/pytest_b.log
timeout 300 python tools/icp_converge.py 50000000 80 1.0 1.5 2.0 3.0 > $O/converge.txt 2>&1; echo "converge rc=$?"; grep scale $O/converge.txt
for v in default tile32 mfma64p4 mfma64p16 mfma64l2 mfma64l2p8 mfma64l3; do
  if [ $v = default ]; then unset E3D_REG_PASS2; else export E3D_REG_PASS2=$v; fi
  timeout 200 python bench.py --only reg --reg-images 4 --no-cpu-baseline > $O/reg_$v.json 2> $O/reg_$v.err
  python - $O/reg_$v.json $v <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]; k = [x for x in r if x.startswith("k_reg_pass2")][0]
    print("pass2 %-10s %.4f ms  pass1 %.4f ms  value %.3g residuals/s  ms_per_run_iteration %.1f" % (sys.argv[2], r[k]["avg_launch_ms"], r["k_reg_pass1"]["avg_launch_ms"], d["value"], d["ms_per_run_iteration"]))
except Exception as e:
    print("pass2", sys.argv[2], "FAILED", e)
P
done
unset E3D_REG_PASS2
for k in 32 8; do for x in 1 0; do
  E3D_KNN_XCD=$x timeout 120 python tools/bench_normals.py --k $k --no-cpu > $O/normals_k${k}_xcd$x.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$O/normals_k${k}_xcd$x.json')); print('normals k=$k xcd=$x: %.3f ms' % d['ms_per_call'])"
done; done
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; tail -c 4000 $O/bench_stdout.txt
