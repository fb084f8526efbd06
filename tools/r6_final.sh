#!/bin/bash
# round 6, closing GPU session: whole -m gpu suite, kernel traces + counter passes of the bench legs (tools/prof_round6.sh), default bench, tool end to end
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"; tail -c 3900 $O/bench_stdout.txt; cp bench_detail.json $O/ 2>/dev/null
timeout 1500 bash tools/prof_round6.sh terrace allpairs icp reg normals > $O/prof.log 2>&1; echo "prof rc=$?"; grep "rc=" $O/prof.log
timeout 400 python tools/bench_tool_icp.py --points 50000000 --iterations 100 > $O/tool_icp.json 2> $O/tool_icp.err; echo "tool rc=$?"; cat $O/tool_icp.json | cut -c1-600
