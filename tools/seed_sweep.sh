#!/bin/bash
# sweep of the seeded far lists' knobs on the headline scene (wall ms per outer iteration; sum over the bench's timed window, iterations 5..24)
cd $GRAFT_REPO_ROOT
O=gpurun_out/seed_sweep; mkdir -p $O
run() { name=$1; shift; env "$@" python tools/icp_trend.py 50000000 26 0 0.01 2 2.5 > $O/$name.txt 2>&1; python - "$O/$name.txt" "$name" <<'PY'
import sys
for ln in open(sys.argv[1]):
    if ln.startswith("wall ms per iteration:"):
        v=[float(x) for x in ln.split(":")[1].split()]
        print("%-22s sum(5..24) %.1f  it 7..12: %s" % (sys.argv[2], sum(v[5:25]), " ".join("%.1f"%x for x in v[7:13])))
PY
}
run default A=1
run default_again A=1
run seed0 E3D_NN_SEED=0
run probes4 E3D_NN_SEED_PROBES=4
run near035 E3D_NN_SEED_NEAR=0.35
run near045 E3D_NN_SEED_NEAR=0.45
run near05 E3D_NN_SEED_NEAR=0.5
run span2 E3D_ROW_SPAN=2
run span3 E3D_ROW_SPAN=3
run span6 E3D_ROW_SPAN=6
run frac05 E3D_NN_SEED_FRAC=0.5
