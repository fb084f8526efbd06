#!/bin/bash
O=gpurun_out/${R4TAG:-r4e}; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -s --durations=5 ${R4K:+-k "$R4K"} > $O/pytest_at_size.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|c5 at size|Error|^E  " $O/pytest_at_size.txt | tail -15
