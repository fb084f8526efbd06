# usage: bash tools/prof_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.txt (+ bench json)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_bench.log 2>&1
python $R/tools/rocpd_summary.py /tmp/prof_$TAG/bench_results.db $R/gpurun_out/${TAG}_kernel_stats.txt "" > /dev/null
grep -E "e3d|rocprim" $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-60,160-260 | head -30
grep "^{" $R/gpurun_out/${TAG}_bench.log | cut -c1-400
