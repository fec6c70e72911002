# SQ counters of the shipped sgd_round, a few per pass (round 1: a pass with 8 counters incl. SQ_INST_LEVEL_VMEM / SQ_LEVEL_WAVES
# exceeded the hardware's budget and aborted rocprofv3; TA_* counters hung it -- none of either here).
#   bash scripts/gpu_sq_r02.sh c2|c1
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CFG=${1:-c2}
OUT=gpurun_out/sq_r02_$CFG; mkdir -p $OUT
ARGS="--config $CFG --no-cpu-baseline --no-secondary"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 bench.py --steps 3 --warmup 1 $ARGS > /dev/null 2> $OUT/p$i.err; echo "pass $i rc=$?"
done
python3 - "$OUT" "$CFG" <<'PY' > $OUT/sq_$CFG.txt
import csv, glob, collections, sys
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "sgd_round" in name and "false>" in name.replace(" ", "").split("(")[0][-8:]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# SQ counters per launch of sgd_round (full-k launches), bench.py --config %s, one rocprofv3 --pmc pass per group of four" % cfg)
for k, v in sorted(acc.items()):
    print("%-28s launches=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
g = lambda k: sum(acc[k]) / len(acc[k]) if acc.get(k) else float("nan")
print("# derived: VALU instructions per wave %.0f, SALU %.0f, VMEM rd %.1f wr %.1f, LDS %.1f; busy fraction of wave cycles: active %.3f wait_any %.3f wait_inst %.3f"
      % (g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU") / g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"), g("SQ_INSTS_VMEM_WR") / g("SQ_WAVES"),
         g("SQ_INSTS_LDS") / g("SQ_WAVES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
PY
cat $OUT/sq_$CFG.txt
find $OUT -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
