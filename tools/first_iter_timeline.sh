cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/first_iter; mkdir -p $O
rm -rf /tmp/fi
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fi -o t -- python $R/tools/icp_trend.py 50000000 1 0 0.01 2 2.5 > $O/trend.txt 2>&1
python - <<'PY' > $O/timeline.txt
import sqlite3,glob
db=sqlite3.connect(glob.glob('/tmp/fi/*results.db')[0])
rows=list(db.execute("select name,start,end from kernels order by start"))
# find the run() window: last big gap; print kernels with gaps
t0=rows[0][1]
prev=None
busy=0
out=[]
for n,s,e in rows:
    gap=(s-prev)/1e6 if prev else 0
    out.append("%9.3f ms  +gap %7.3f  dur %7.3f  %s"%((s-t0)/1e6,gap,(e-s)/1e6,n[:70]))
    prev=e
print("\n".join(out[-400:]))
PY
tail -3 $O/trend.txt | cut -c1-200; wc -l $O/timeline.txt
