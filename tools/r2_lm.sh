#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_icp.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -2
for m in 1 0; do
E3D_LM_PAIR=$m python bench.py --no-cpu-baseline --no-reg --no-normals --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
a=d['allpairs']; print('pair=$m allpairs ms/iter', round(a['ms_per_iter'],1), {k:round(v,1) for k,v in a['rank0_ms_per_iter'].items()}, 'passes', round(a['lm_passes_per_iter'],2))"
done
