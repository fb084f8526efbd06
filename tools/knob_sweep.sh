#!/bin/bash
# wall ms per outer iteration of the headline scene (tools/icp_trend.py) under one environment setting per run:
#   bash tools/knob_sweep.sh NAME=VALUE [NAME=VALUE ...]     (each argument is one run; "A=1" = the defaults)
cd $GRAFT_REPO_ROOT
O=gpurun_out/knob_sweep; mkdir -p $O
for kv in "$@"; do
  name=$(echo "$kv" | tr '= ,' '___')
  env $(echo "$kv" | tr ',' ' ') python tools/icp_trend.py 50000000 26 0 0.01 2 2.5 > $O/$name.txt 2>&1
  python - "$O/$name.txt" "$kv" <<'PY'
import sys
for ln in open(sys.argv[1]):
    if ln.startswith("wall ms per iteration:"):
        v=[float(x) for x in ln.split(":")[1].split()]
        print("%-34s sum(5..24) %.1f  ramp 5..7 %.1f  transition 8..12 %.1f  settling 13..24 %.1f" % (sys.argv[2], sum(v[5:25]), sum(v[5:8]), sum(v[8:13]), sum(v[13:25])))
PY
done
