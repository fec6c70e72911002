#!/bin/bash
# Fold gain g0 * (n / n0 + 1)^p: the floor (rows cut into a few chains) against the head rows.  gpurun_out/gain_shape.log
# usage: gpu_gain_shape.sh "g0 n0 p" ...
mkdir -p gpurun_out
L=gpurun_out/gain_shape.log; : > $L
for gn in "$@"; do
  set -- $gn
  export MFX_HOT_S_GAIN=$1 MFX_HOT_S_N0=$2 MFX_HOT_S_POW=$3 TAG="$1,$2,$3"
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 20 30 >> $L 2>&1 || exit 1
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 12 15 >> $L 2>&1 || exit 1
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c2s 12 5 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c3shard 8 5 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 12 3 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 8 3 >> $L 2>&1 || exit 1
done
