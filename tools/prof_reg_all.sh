cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pr2 -o r -- python $R/tools/prof_reg_run.py 0 4 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr2/r_results.db $R/gpurun_out/reg_run_all_kernels.txt "" > /dev/null
python - <<PY
import re
for line in open("$R/gpurun_out/reg_run_all_kernels.txt").read().split("\n")[:16]:
    m=re.match(r"(.*), (\d+), (\d+), (\d+), (\d+), (\d+)", line.strip())
    if m:
        name=re.sub(r"\(.*","",m.group(1))[-70:]
        print("%-72s calls %4s total %8.3f ms avg %8.1f us"%(name,m.group(2),int(m.group(3))/1e6,int(m.group(4))/1e3))
PY
