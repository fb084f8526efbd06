#!/bin/bash
# Round-6 evidence (one MI355X): kernel-trace stats of the default bench line, HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes)
# and SQ counters of the kernels of the three paths.  Summaries land in gpurun_out/r6prof/ ; the ones to judge are copied to profiles/.
# usage: bash tools/prof_round6.sh [stage ...]   stages: trace terrace partial allpairs icp reg normals sq nsq nta   (default: terrace allpairs icp reg normals)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof
mkdir -p $O
STAGES=${@:-terrace allpairs icp reg normals}
SUM="python $R/tools/rocpd_summary.py"
# (the headline leg alone, iterations 0 .. 15: the ramp, the transition and the first settling iterations of the default scene)
ICP="python $R/bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-partial --no-regression --no-scanner --no-whole-run --steps 6 --warmup 10"
REG="python $R/bench.py --only reg --no-cpu-baseline --reg-images 4"
pmc() {  # pmc <tag> <counters...> -- <command>
  local tag=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/r6p_$tag
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/r6p_$tag -o p -- "$@" > /dev/null 2>&1
  echo "[$tag] rc=$?"
  $SUM /tmp/r6p_$tag/p_results.db $O/$tag.txt "${FILTER:-e3d}" > /dev/null 2>&1
}
for st in $STAGES; do
  case $st in
    trace)
      rm -rf /tmp/r6p_trace
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r6p_trace -o b -- python $R/bench.py > $O/bench_traced.json 2> $O/bench_traced.err
      echo "[trace] rc=$?"
      $SUM /tmp/r6p_trace/b_results.db $O/bench_kernel_stats.txt "" > /dev/null 2>&1
      head -25 $O/bench_kernel_stats.txt | cut -c1-60,150-230 ;;
    terrace)   # the headline leg alone: its kernel averages are the ones bench.py's live HIP-event figures must agree with
      rm -rf /tmp/r6p_terrace
      timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r6p_terrace -o b -- python $R/bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-partial --no-regression --no-scanner --no-whole-run > $O/terrace_traced.json 2> /dev/null
      echo "[terrace] rc=$?"
      $SUM /tmp/r6p_terrace/b_results.db $O/terrace_kernel_stats.txt e3d > /dev/null 2>&1
      head -12 $O/terrace_kernel_stats.txt | cut -c1-60,150-230 ;;
    partial)   # the partial-overlap leg alone (bench.py --partial-only): the same kernels where half of the queries find no partner
      rm -rf /tmp/r6p_partial
      timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/r6p_partial -o b -- python $R/bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-whole-run --partial-only > $O/partial_traced.json 2> /dev/null
      echo "[partial] rc=$?"
      $SUM /tmp/r6p_partial/b_results.db $O/partial_kernel_stats.txt e3d > /dev/null 2>&1
      head -12 $O/partial_kernel_stats.txt | cut -c1-60,150-230 ;;
    allpairs)   # the all-pairs leg alone: kernel trace of its ten timed iterations
      rm -rf /tmp/r6p_ap
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r6p_ap -o b -- python $R/bench.py --only allpairs --no-scale-model > $O/allpairs_traced.json 2> /dev/null
      echo "[allpairs] rc=$?"
      $SUM /tmp/r6p_ap/b_results.db $O/allpairs_kernel_stats.txt e3d > /dev/null 2>&1
      head -14 $O/allpairs_kernel_stats.txt | cut -c1-60,150-230 ;;
    icp)
      pmc icp_fetch FETCH_SIZE -- $ICP
      pmc icp_write WRITE_SIZE -- $ICP ;;
    reg)
      pmc reg_fetch FETCH_SIZE -- $REG
      pmc reg_write WRITE_SIZE -- $REG ;;
    normals)
      for k in 32 8; do
        FILTER="" pmc normals_k${k}_fetch FETCH_SIZE -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
        FILTER="" pmc normals_k${k}_write WRITE_SIZE -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
      done ;;
    sq)
      A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
      B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
      FILTER=k_nn pmc icp_sq_a $A -- $ICP
      FILTER=k_nn pmc icp_sq_b $B -- $ICP
      ;;
    nsq)     # SQ counters of the kNN normal kernels, k = 32 and k = 8
      A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"
      B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
      for k in ${NK:-32 8}; do
        FILTER="" pmc normals_k${k}_sq_a $A -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
        FILTER="" pmc normals_k${k}_sq_b $B -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
      done ;;
    nta)     # texture-addresser / vector-L1 counters of the kNN normal kernels (is the scan bound by its loads' address processing?)
      A="TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
      B="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
      rocprofv3 --list-avail 2>/dev/null | grep -E "TA_|TCP_" | head -80 > $O/avail_ta_tcp.txt
      for k in ${NK:-32 8}; do
        FILTER="" pmc normals_k${k}_ta_a $A -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
        FILTER="" pmc normals_k${k}_ta_b $B -- python $R/tools/bench_normals.py --k $k --no-cpu --repeat 1
      done ;;
  esac
done
ls -la $O
