#!/bin/bash
# One task per wave everywhere (MFX_ONE_TASK=1) against the default rule (2 from 128 ratings per wave and launch): time over shapes, parity spreads
mkdir -p gpurun_out
L=gpurun_out/onetask.log; : > $L
timeout -k 10 500 python scripts/gpu_ab.py lib lib:MFX_ONE_TASK=1 >> $L 2>&1 || exit 1
timeout -k 10 600 python scripts/gpu_ab.py --dense lib lib:MFX_ONE_TASK=1 >> $L 2>&1 || exit 1
for ot in "" 1; do
  export MFX_ONE_TASK=$ot TAG="one_task=$ot"
  [ -z "$ot" ] && unset MFX_ONE_TASK
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 20 20 >> $L 2>&1 || exit 1
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 12 10 >> $L 2>&1 || exit 1
  timeout -k 10 200 python scripts/gpu_rmse_spread.py c2s 12 4 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c3shard 8 4 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 12 3 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 8 3 >> $L 2>&1 || exit 1
  timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 20 3 >> $L 2>&1 || exit 1
done
