#!/bin/bash
O=gpurun_out/${R4TAG:-r4e}; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -s --durations=5 > $O/pytest_at_size.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|c5 at size|Error|assert" $O/pytest_at_size.txt | tail -15
timeout 600 python -m pytest tests/test_gpu_icp.py -q -m gpu -k "resident or lm_tries or plane or identical" > $O/pytest_icp.txt 2>&1; echo "pytest icp rc=$?"; tail -2 $O/pytest_icp.txt
