#!/bin/bash
O=gpurun_out/${R4TAG:-r4o}; mkdir -p $O
for k in 32 8; do
  for f in default 3 6 12 24; do
    if [ $f = default ]; then unset E3D_KNN_CELL_FACTOR; else export E3D_KNN_CELL_FACTOR=$f; fi
    E3D_KNN_STATS=1 timeout 300 python tools/bench_normals.py --k $k --no-cpu --angular --repeat 2 > $O/na_k${k}_f$f.json 2> $O/na_k${k}_f$f.err
    python -c "
import json
d=json.loads(open('$O/na_k${k}_f$f.json').read().strip().splitlines()[-1]); print('k=$k factor=$f', round(d['ms_per_call'],2), 'ms')
"
    grep "^\[knn\]" $O/na_k${k}_f$f.err | tail -8 | cut -c1-110
  done
done
