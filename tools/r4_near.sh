#!/bin/bash
O=gpurun_out/${R4TAG:-r4l}; mkdir -p $O
for near in 0.4 0.7 1.0; do
  E3D_NN_NEAR=$near timeout 600 python tools/icp_trend.py 10000000 6 0 0.02 16 > $O/ap_near_$near.txt 2>&1
  echo "== all-pairs near $near"; tail -6 $O/ap_near_$near.txt | cut -c1-150
  E3D_NN_NEAR=$near timeout 600 python tools/icp_trend.py 50000000 10 0 0.01 2 > $O/t_near_$near.txt 2>&1
  echo "== terrace near $near"; tail -10 $O/t_near_$near.txt | cut -c1-150
done
