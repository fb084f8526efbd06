#!/bin/bash
# round 5, GPU call 6: (B) f64 pass 2 with XCD-contiguous chunks (grid sizes), kNN leftovers (list capacity of the two-pass variant, sampling stride)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_reg.py -x -q -k "pass2 or accumul or whole_problem or run_on_current" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -2 $O/pytest_a.log
for b in 0 256 1024 2048; do
  if [ $b = 0 ]; then unset E3D_REG_PASS2_BLOCKS; else export E3D_REG_PASS2_BLOCKS=$b; fi
  timeout 200 python bench.py --only reg --reg-images 4 --no-cpu-baseline > $O/reg_b$b.json 2> /dev/null
  python - $O/reg_b$b.json $b <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; k = [x for x in r if x.startswith("k_reg_pass2")][0]
print("pass2 blocks %-5s %.4f ms  pass1 %.4f ms" % (sys.argv[2], r[k]["avg_launch_ms"], r["k_reg_pass1"]["avg_launch_ms"]))
P
done
unset E3D_REG_PASS2_BLOCKS
for k in 8 32; do for ce in 4 8 12 20; do for st in 8 16; do
  E3D_KNN_CAP_EXTRA=$ce E3D_KNN_REP_STRIDE=$st timeout 120 python tools/bench_normals.py --k $k --no-cpu > $O/n_k${k}_ce${ce}_st$st.json 2>/dev/null
  E3D_KNN_CAP_EXTRA=$ce E3D_KNN_REP_STRIDE=$st timeout 120 python tools/bench_normals.py --k $k --no-cpu --angular > $O/na_k${k}_ce${ce}_st$st.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/n_k${k}_ce${ce}_st$st.json')); a=json.load(open('$O/na_k${k}_ce${ce}_st$st.json')); print('normals k=$k cap_extra=$ce stride=$st: uniform %.3f ms  scanner-sampled %.3f ms' % (d['ms_per_call'], a['ms_per_call']))"
done; done; done
