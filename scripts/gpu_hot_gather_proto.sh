#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/hot_gather_proto3.log; : > $L
for hl in 128 48 24; do
  export PROTO_HOT_LEN=$hl TAG="hot_len=$hl"
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2s 12 1000 2 >> $L 2>&1 || exit 1
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2 12 1000 2 >> $L 2>&1 || exit 1
done
