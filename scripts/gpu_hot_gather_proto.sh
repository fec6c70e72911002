#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/hot_gather_proto2.log; : > $L
export MFX_HOT_S_GAIN=1 MFX_HOT_S_N0=2 MFX_HOT_S_POW=0.5
for sd in 0 250 1000; do
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2 12 $sd 2 >> $L 2>&1 || exit 1
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c3shard 8 $sd 2 >> $L 2>&1 || exit 1
done
