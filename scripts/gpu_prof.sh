# rocprofv3 kernel trace + HBM traffic counters for the bench workload (run on the GPU box).
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01
mkdir -p $OUT
# A. variants: launch stream
python scripts/gpu_streams.py > gpurun_out/streams.log 2>&1; tail -6 gpurun_out/streams.log
# B. kernel trace + stats (the program itself after --, no wrappers)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err; echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
# C. counters, each in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$T -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$T.err; echo "pmc $T rc=$?"
done
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
rm -rf $OUT/trace/*/*.db 2>/dev/null
du -sh $OUT
