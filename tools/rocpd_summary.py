"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel stats and PMC sums -> small text files.
usage: python tools/rocpd_summary.py <results.db> <out.txt> [name_filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); out = open(sys.argv[2], "w"); flt = sys.argv[3] if len(sys.argv) > 3 else ""
out.write("# kernel stats (durations in ns): name, calls, total_ns, avg_ns, min_ns, max_ns\n")
for r in db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"):
    if flt in r[0]:
        out.write("%s, %d, %d, %.0f, %d, %d\n" % (r[0][:160], r[1], r[2], r[3], r[4], r[5]))
try:
    rows = list(db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name order by 1,2"))
    if rows:
        out.write("# PMC: kernel, counter, sum over dispatches, dispatches\n")
        for r in rows:
            if flt in r[0]:
                out.write("%s, %s, %.6g, %d\n" % (r[0][:80], r[1], r[2], r[3]))
        # the same restricted to a kernel's LARGEST dispatches (counter value >= 0.8 x its maximum): a run that ramps up -- the first
        # ICP iterations hold fewer correspondences -- otherwise averages launches of different sizes
        out.write("# PMC_FULL: kernel, counter_FULL, sum over the dispatches with value >= 0.8 max, dispatches\n")
        per = {}
        for kn, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
            per.setdefault((kn, cn), []).append(v)
        for (kn, cn), vs in sorted(per.items()):
            if flt in kn:
                big = [v for v in vs if v >= 0.8 * max(vs)]
                out.write("%s, %s_FULL, %.6g, %d\n" % (kn[:80], cn, sum(big), len(big)))
except Exception as e:  # noqa
    out.write("# no counters: %s\n" % e)
out.close()
print(open(sys.argv[2]).read()[:6000])
