# End-of-milestone GPU pass: parity tests, smoke, bench line, rocprofv3 kernel trace + HBM counters.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/round_$R; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err; echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -4 $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$T -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$T.err; echo "pmc $T rc=$?"
done
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
rm -rf $OUT/trace/*/*.db $OUT/trace/*/*kernel_trace.csv 2>/dev/null
