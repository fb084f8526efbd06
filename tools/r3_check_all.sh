#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.txt | tail -3; grep -A10 "slowest" $O/pytest.txt | head -12
bash tools/prof_round3.sh trace terrace icp reg normals 2>&1 | grep -E "^\[|rc=" 
ls gpurun_out/r3prof
