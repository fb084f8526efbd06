# kernel table of the (B) optimisation loop for one camera model: bash tools/prof_reg_model.sh <model 0|1|2>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-2}
rocprofv3 --kernel-trace --stats -d /tmp/prm$M -o r -- python $R/tools/prof_reg_run.py $M 4 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/prm$M/r_results.db $R/gpurun_out/reg_run_model${M}_kernels.txt "" > /dev/null
python - <<PY
import re
for line in open("$R/gpurun_out/reg_run_model${M}_kernels.txt").read().split("\n")[:14]:
    m=re.match(r"(.*), (\d+), (\d+), (\d+), (\d+), (\d+)", line.strip())
    if m:
        name=re.sub(r"\(.*","",m.group(1))[-70:]
        print("%-72s calls %4s total %8.3f ms avg %8.1f us"%(name,m.group(2),int(m.group(3))/1e6,int(m.group(4))/1e3))
PY
