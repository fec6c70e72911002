set -o pipefail
Q=gpurun_out/r3_${VARIANT:-lib_shfl}_ab.log; : > $Q
S=/root/repo/question-recommendation-system_amd/${VARIANT:-lib_shfl}/libmf.so
for lib in "$S" ""; do
  echo "== ${lib:-default build}" >> $Q
  for c in "c2 12" "c1 12" "c2s 12" "c4shard 4"; do MFX_LIB=$lib timeout -k 10 300 python scripts/gpu_quick.py $c 2 2>&1 | grep -v amdgpu.ids >> $Q || exit 1; done
  MFX_LIB=$lib timeout -k 10 300 python scripts/gpu_quick.py c1 12 1 wide=1 2>&1 | grep -v amdgpu.ids >> $Q || exit 1
  MFX_LIB=$lib timeout -k 10 400 python scripts/gpu_k_sweep.py 8 32 128 256 2>&1 | grep -v amdgpu.ids >> $Q || exit 1
done
cat $Q
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "single_pass or workgroup_visit or split_rows or duplicate" 2>&1 | tail -3
