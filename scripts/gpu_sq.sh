# NOTE: a pass with TA_*_sum counters (TA_ADDR_STALLED_BY_TC_CYCLES_sum ...) hung rocprofv3 on this pool; do not add one.
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/sq_r01; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err; echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/sq_r01/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sgd_round<8, true, false>" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-36s n=%d mean=%.4g" % (k, len(v), sum(v)/len(v)))
PY
