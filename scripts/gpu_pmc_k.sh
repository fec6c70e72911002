# Fabric traffic of sgd_round at another width on the bench stream: bash scripts/gpu_pmc_k.sh <k> [opt=v ...]
# (separate --pmc passes, no trace options; summarize_pmc.py reads the pmc_* directories)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
K=$1; shift
TAG=k${K}$(echo "$*" | tr -d ' =')
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$T -- python3 scripts/gpu_k_sweep.py $K "$@" > $OUT/run_$T.log 2> $OUT/pmc_$T.err; echo "pmc $T rc=$?"
done
python3 scripts/summarize_pmc.py $OUT $TAG > $OUT/pmc.json; cat $OUT/pmc.json | python3 -c "import json,sys; d=json.load(sys.stdin); print('$TAG', {k:d[k] for k in d if 'per_launch' in k or 'rate' in k})"
find $OUT -name "*counter_collection.csv" -size +1M -delete 2>/dev/null; rm -rf $OUT/*/*/*.db 2>/dev/null
