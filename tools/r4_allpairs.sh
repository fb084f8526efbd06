#!/bin/bash
O=gpurun_out/${R4TAG:-r4k}; mkdir -p $O
E3D_NN_STATS=0 timeout 900 python tools/icp_trend.py 10000000 ${R4ITERS:-14} 0 0.02 16 > $O/trend_allpairs.txt 2>&1; echo "rc=$?"; tail -$((${R4ITERS:-14}+3)) $O/trend_allpairs.txt | cut -c1-250
