#!/bin/bash
# round 6, closing session after the (B) changes (cost from the next accumulation, per-block counts from k_obs_eval): whole -m gpu suite,
# default bench, counter passes of the (B) leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6close2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; tail -c 3900 $O/bench_stdout.txt; cp bench_detail.json $O/ 2>/dev/null
timeout 900 bash tools/prof_round6.sh reg > $O/prof.log 2>&1; echo "prof rc=$?"; grep "rc=" $O/prof.log
