# A/B of an experiment library (make variant) against the shipped one: gpu_variant_ab.sh <libdir> <tag> [divs...]
set -o pipefail
V=$1; TAGN=$2; shift 2
L=gpurun_out/r3_${TAGN}.log
A=/root/repo/question-recommendation-system_amd/$V/libmf.so
: > $L
for cd in "$@"; do
  echo "== $V, conflict_div=$cd" >> $L
  CHECK_ACC=1 MFX_LIB=$A timeout -k 10 120 python scripts/gpu_quick.py c1 12 1 conflict_div=$cd 2>&1 | grep -v amdgpu.ids >> $L || exit 1
  MFX_LIB=$A timeout -k 10 120 python scripts/gpu_quick.py c1 20 2 conflict_div=$cd 2>&1 | grep -v amdgpu.ids >> $L || exit 1
  MFX_LIB=$A TAG=$TAGN timeout -k 10 300 python scripts/gpu_heldout_quick.py uniform zipf11 rect eta005_lam001 eta02_lam001 zipf11_k64 conflict_div=$cd 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done
echo "== $V c2, c2s, c3shard (default cap)" >> $L
MFX_LIB=$A timeout -k 10 200 python scripts/gpu_quick.py c2 12 1 2>&1 | grep -v amdgpu.ids >> $L || exit 1
MFX_LIB=$A timeout -k 10 200 python scripts/gpu_quick.py c2s 12 1 2>&1 | grep -v amdgpu.ids >> $L || exit 1
MFX_LIB=$A timeout -k 10 200 python scripts/gpu_quick.py c3shard 8 1 2>&1 | grep -v amdgpu.ids >> $L || exit 1
