"""Per-launch counters of sgd_round (full-k launches) from rocprofv3 --pmc CSVs -> JSON for bench.py's roofline block.

  python3 scripts/summarize_pmc.py <dir with pmc_*/ sub-directories> <config name> > profiles/rNN_pmc_<config>.json

MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64-B units of 128-B reads -> x2 on gfx950; FETCH_SIZE and
WRITE_SIZE are in KiB; both count traffic between the L2s and the fabric (Infinity-Cache hits included).
Each counter group comes from its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)."""
import csv, glob, json, os, sys

out, cfg = sys.argv[1], sys.argv[2]


def per_kernel(counter):
    vals = []
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            # full-k launches only: sgd_round<LANES, FULL, SLOW=false>
            if "sgd_round" in name and "false>" in name.replace(" ", "").split("(")[0][-8:] and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


res = {"config": cfg, "kernel": "sgd_round (full-k launches)", "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch"}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum",
          "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum"):
    v = per_kernel(c)
    if v:
        res[c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["read_bytes_per_launch"] = 2.0 * res["FETCH_SIZE"]["mean"] * 1024.0   # x2: gfx950 correction
    res["write_bytes_per_launch"] = res["WRITE_SIZE"]["mean"] * 1024.0
    res["traffic_bytes_per_launch"] = res["read_bytes_per_launch"] + res["write_bytes_per_launch"]
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
    h, m = res["TCC_HIT_sum"]["mean"], res["TCC_MISS_sum"]["mean"]
    res["l2_requests_per_launch"] = h + m
    res["l2_hit_rate"] = h / (h + m) if h + m else None
print(json.dumps(res, indent=1))
