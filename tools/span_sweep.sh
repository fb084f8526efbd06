R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -3
for sp in 0 1 2 4 8; do echo span $sp; E3D_ROW_SPAN=$sp python $R/bench.py --no-cpu-baseline --steps 3 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_iter'],2), {k:round(v,2) for k,v in d['breakdown_ms_per_iter'].items()})"; done
