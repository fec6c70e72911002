#!/bin/bash
# How much of the GPU-vs-emulation difference is concurrency: the same data run with few waves per XCD.
mkdir -p gpurun_out
L=gpurun_out/conc_diag.log; : > $L
export MFX_HOT_S_GAIN=1 MFX_HOT_S_N0=2 MFX_HOT_S_POW=0.5
for w in 1 4 16 0; do
  export MFX_WGS_PER_XCD=$w TAG="wgs/xcd=$w"
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2s 12 2 >> $L 2>&1 || exit 1
done
