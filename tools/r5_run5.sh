#!/bin/bash
# round 5, GPU call 5: coarse distance field (tests + trend on the headline and partial-overlap scenes, grid build time), all-pairs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_icp.py -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
for f in 8 0 12; do
  E3D_NN_FIELD=$f timeout 200 python tools/icp_trend.py 50000000 40 0 0.01 2 3.0 > $O/trend_p3_field$f.txt 2>&1; echo "FIELD=$f"; grep -E "wall ms|converged" $O/trend_p3_field$f.txt
done
for f in 8 0; do
  E3D_NN_FIELD=$f timeout 200 python tools/icp_trend.py 50000000 30 1 0.01 2 1.0 > $O/trend_partial_field$f.txt 2>&1; echo "partial FIELD=$f"; grep -E "wall ms|converged" $O/trend_partial_field$f.txt
done
timeout 900 python -m pytest tests/test_gpu_at_size.py -x -q -k "c3_all_pairs" > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?"; tail -3 $O/pytest_b.log
timeout 600 python bench.py --only allpairs --no-scale-model > $O/allpairs.json 2> $O/allpairs.err; echo "allpairs rc=$?"
python -c "
import json; d=json.load(open('$O/allpairs.json')); print('allpairs: ms_per_iter %.1f steady %.1f' % (d['ms_per_iter'], d['ms_per_iter_steady'])); print('each', [round(v,1) for v in d['ms_per_iter_each']])"
