# traffic of L2-sized gathered stripes (exploration)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/exp12; mkdir -p $O; rm -rf $O/*
i=0
for V in "" "owner_side=1 stripes=32" "owner_side=1 stripes=64" "stripes=32"; do
  i=$((i+1))
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $O/v${i}_$T -- python3 scripts/gpu_case.py c2 4 $V > $O/v${i}_$T.out 2> $O/v${i}_$T.err
  done
  grep CASE $O/v${i}_FETCH_SIZE.out | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[5:]); print('variant $i %-30s %9.3f ms/epoch %8.1f us/launch stripes %d wg/cu %d' % (d['opts'], d['ms_epoch'], d['us_launch'], d['stripes'], d['wg_per_cu']))
"
  python3 - $O $i <<'PY'
import csv, glob, sys
out, i = sys.argv[1], sys.argv[2]
def per(counter):
    v = []
    for f in glob.glob("%s/v%s_*/**/*counter_collection.csv" % (out, i), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "sgd_round" in n and "false>" in n.replace(" ", "").split("(")[0][-8:] and row["Counter_Name"] == counter:
                v.append(float(row["Counter_Value"]))
    return v
f, w, h, m = per("FETCH_SIZE"), per("WRITE_SIZE"), per("TCC_HIT_sum"), per("TCC_MISS_sum")
if f and w:
    n = len(f)
    print("   launches %d: read %.1f MB + write %.1f MB per launch; per epoch %.2f GB; L2 hit %.3f" % (n, 2 * sum(f) / n / 1024, sum(w) / n / 1024, (2 * sum(f) + sum(w)) / 1024 / 1024 / 3, sum(h) / (sum(h) + sum(m)) if h else -1))
PY
done
find $O -name "*.csv" -delete
