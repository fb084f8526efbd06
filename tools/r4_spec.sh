#!/bin/bash
O=gpurun_out/${R4TAG:-r4p}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_multiprocess.py tests/test_gpu_distributed.py tests/test_gpu_cli.py -x -q -m gpu > $O/pytest_icp.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_icp.txt
for sp in 1 0; do
  E3D_LM_SPECULATE=$sp timeout 600 python tools/icp_trend.py 50000000 25 0 0.01 2 > $O/trend_spec$sp.txt 2>&1
  echo "== speculate=$sp"; tail -12 $O/trend_spec$sp.txt | cut -c150-260
done
E3D_LM_SPECULATE=1 timeout 600 python tools/icp_trend.py 10000000 12 0 0.02 16 > $O/trend_ap_spec1.txt 2>&1; echo "== allpairs speculate=1"; tail -6 $O/trend_ap_spec1.txt | cut -c150-260
timeout 600 python bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-partial > $O/bench_terrace.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_terrace.json').read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['ms_per_step_steady'], [round(x,2) for x in d['ms_per_step_each'][-6:]])"
