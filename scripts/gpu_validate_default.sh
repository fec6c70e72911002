#!/bin/bash
# Parity spreads of the default build over all full-size fixtures, then the GPU suite twice.  gpurun_out/validate_default.log
mkdir -p gpurun_out
L=gpurun_out/validate_default.log; : > $L
export TAG="default"
timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 20 30 >> $L 2>&1 || exit 1
timeout -k 10 200 python scripts/gpu_rmse_spread.py c1 12 10 >> $L 2>&1 || exit 1
timeout -k 10 200 python scripts/gpu_rmse_spread.py c2s 12 4 >> $L 2>&1 || exit 1
timeout -k 10 300 python scripts/gpu_rmse_spread.py c3shard 8 4 >> $L 2>&1 || exit 1
timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 12 5 >> $L 2>&1 || exit 1
timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 8 3 >> $L 2>&1 || exit 1
timeout -k 10 300 python scripts/gpu_rmse_spread.py c2 20 3 >> $L 2>&1 || exit 1
cat $L
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_a.log 2>&1 || { tail -n 30 gpurun_out/pytest_gpu_a.log; exit 1; }
tail -n 2 gpurun_out/pytest_gpu_a.log
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_b.log 2>&1 || { tail -n 30 gpurun_out/pytest_gpu_b.log; exit 1; }
tail -n 2 gpurun_out/pytest_gpu_b.log
