# round 3: one rank's compute of the strong split (N = 2, 4, 8), then the fixtures quickly
set -o pipefail
L=gpurun_out/r3_one_rank.log; : > $L
for w in 2 4 8; do timeout -k 10 300 python scripts/gpu_one_rank.py $w c2 2>&1 | grep -v amdgpu.ids >> $L || exit 1; done
RANK_SIM=7 timeout -k 10 300 python scripts/gpu_one_rank.py 8 c2 2>&1 | grep -v amdgpu.ids >> $L || exit 1
timeout -k 10 300 python scripts/gpu_one_rank.py 8 c2 conflict_div=12 2>&1 | grep -v amdgpu.ids >> $L || exit 1
cat $L
Q=gpurun_out/r3_quick_after_cut.log; : > $Q
for c in "c1 12" "c1 20" "c2 12" "c2s 12" "c3shard 8" "c4shard 4"; do timeout -k 10 300 python scripts/gpu_quick.py $c 2 2>&1 | grep -v amdgpu.ids >> $Q || exit 1; done
cat $Q
