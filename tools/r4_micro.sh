#!/bin/bash
O=gpurun_out/${R4TAG:-r4d}; mkdir -p $O
timeout 300 tools/micro/lm_variants 100 2 > $O/lm_variants_2sets.txt 2>&1; echo "lm_variants rc=$?"
grep -E "cost_multi rows|dense outer=src|mode1 compacted" $O/lm_variants_2sets.txt
timeout 300 python tools/icp_trend.py 50000000 25 0 > $O/trend_full.txt 2>&1; tail -27 $O/trend_full.txt | cut -c1-230
bash tools/prof_round4.sh nsq 2>&1 | grep -E "^\[|rc="
