"""Per-launch HBM bytes of sgd_round from rocprofv3 --pmc CSVs (MI355X_MICROARCH.md HBM section:
FETCH_SIZE counts 64-B units of 128-B reads -> x2 on gfx950; both counters are in KiB)."""
import csv, glob, os, sys
out = sys.argv[1]
def per_kernel(tag, counter):
    vals = []
    for f in glob.glob(os.path.join(out, "pmc_%s" % tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "sgd_round" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return vals
for tag, names in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum_TCC_MISS_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    for c in names:
        v = per_kernel(tag, c)
        if v:
            # the first launches of a run are the slow_only epoch; report the full-k ones (last half)
            w = v[len(v)//2:]
            print("%s: %d launches, mean %.1f (all), mean of last half %.1f" % (c, len(v), sum(v)/len(v), sum(w)/len(w)))
        else:
            print("%s: no rows found" % c)
