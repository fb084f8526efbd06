#!/bin/bash
run() { echo "$1: $(python tools/bench_c4.py --images 3 --accumulate-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('p1', round(d['pass1_ms_per_image'],3), 'p2', round(d['pass2_ms_per_image'],3), 'acc', round(d['accumulate_ms_all_images'],2))")"; }
E3D_REG_PASS2_OCC=2 E3D_REG_PASS2_BLOCKS=1024 run occ2_b1024
E3D_REG_PASS2_OCC=2 run occ2_b512
E3D_REG_PASS2_OCC=3 run occ3_b768
E3D_REG_PASS2_OCC=3 E3D_REG_PASS2_BLOCKS=1024 run occ3_b1024
timeout 600 python -m pytest tests/test_gpu_reg.py -x -q -m gpu -k "accumulate or whole_problem or rig_image" 2>&1 | tail -3
