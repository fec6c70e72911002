#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/hot_gather_proto4.log; : > $L
for cc in 0 1; do
  export PROTO_CONCURRENT=$cc TAG="concurrent=$cc"
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2s 12 1000 2 >> $L 2>&1 || exit 1
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2 12 1000 2 >> $L 2>&1 || exit 1
  timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c1 20 1000 5 >> $L 2>&1 || exit 1
done
timeout -k 10 400 python scripts/gpu_hot_gather_proto.py c2 12 0 2 >> $L 2>&1
