#!/bin/bash
# Phases of the device plan builder (MFX_PLAN_TIMING=1) at configs[1] and configs[2] size.  gpurun_out/plan_timing.log
mkdir -p gpurun_out
MFX_PLAN_TIMING=1 timeout -k 10 600 python scripts/gpu_facade_time.py c1 2 2>&1 | grep -v "^ *[0-9]\+ \|^iter" > gpurun_out/plan_timing.log
MFX_PLAN_TIMING=1 timeout -k 10 600 python scripts/gpu_facade_time.py c2 2 2>&1 | grep -v "^ *[0-9]\+ \|^iter" >> gpurun_out/plan_timing.log
