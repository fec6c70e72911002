# round 3, final code: every full-size fixture and every held-out law, two single runs each (profiles/experiments/r03_fixtures_every_run.log)
set -o pipefail
Q=gpurun_out/r3_final_parity.log; : > $Q
for c in "c1 12" "c1 20" "c2 8" "c2 12" "c2 20" "c2s 12" "c3shard 8" "c4shard 4"; do timeout -k 10 300 python scripts/gpu_quick.py $c 2 2>&1 | grep -v amdgpu.ids >> $Q || exit 1; done
timeout -k 10 600 python scripts/gpu_heldout_quick.py uniform zipf11 zipf11dup rect eta005_lam001 eta005_lam05 eta02_lam001 eta02_lam05 zipf11_k64 2>&1 | grep -v amdgpu.ids >> $Q || exit 1
echo "== wide=1" >> $Q
for c in "c1 12" "c1 20"; do timeout -k 10 300 python scripts/gpu_quick.py $c 2 wide=1 2>&1 | grep -v amdgpu.ids >> $Q || exit 1; done
timeout -k 10 600 python scripts/gpu_heldout_quick.py uniform zipf11 rect eta005_lam001 eta02_lam001 zipf11_k64 wide=1 2>&1 | grep -v amdgpu.ids >> $Q || exit 1
cat $Q
