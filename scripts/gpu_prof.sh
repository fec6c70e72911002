# rocprofv3 kernel trace + traffic counters for the bench workload (run on the GPU box).
#   bash scripts/gpu_prof.sh c2|c1 [tag]
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFG=${1:-c2}
TAG=${2:-r03}
OUT=gpurun_out/prof_${TAG}_$CFG
mkdir -p $OUT
ARGS="--config $CFG --no-cpu-baseline --no-secondary"
# A. kernel trace + stats (the program itself after --, no wrappers)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err; echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv
# B. counters, each group in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2); no trace options with --pmc
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$T -- python3 bench.py --steps 3 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_$T.err; echo "pmc $T rc=$?"
done
python3 scripts/summarize_pmc.py $OUT $CFG > $OUT/pmc_$CFG.json 2> $OUT/summarize.err; cat $OUT/pmc_$CFG.json
rm -rf $OUT/trace/*/*.db 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
du -sh $OUT
