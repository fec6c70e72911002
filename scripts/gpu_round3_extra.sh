# round 3: same-box A/B of the committed library (lib_head) against the working tree (default plan and wide=1)
set -o pipefail
Q=gpurun_out/r3_wide_ab.log; : > $Q
H=/root/repo/question-recommendation-system_amd/lib_head/libmf.so
run() { # label, env lib, opts
  echo "== $1" >> $Q
  for c in "c1 12" "c1 20" "c2s 12" "c3shard 8" "c2 12" "c4shard 4"; do MFX_LIB=$2 timeout -k 10 300 python scripts/gpu_quick.py $c 1 $3 2>&1 | grep -v amdgpu.ids >> $Q || exit 1; done
  MFX_LIB=$2 timeout -k 10 400 python scripts/gpu_heldout_quick.py zipf11 rect zipf11_k64 eta02_lam001 $3 2>&1 | grep -v amdgpu.ids >> $Q || exit 1
}
run "HEAD library" $H ""
run "working tree, default" "" ""
run "working tree, wide=1" "" "wide=1"
cat $Q
