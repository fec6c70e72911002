# Full check of the round on the GPU box: tests, smoke, bench line.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r2/bench.json").read().strip().splitlines()[-1])
print("value %.3e  ms/step %.3f  frac %.3f  traffic_frac %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_frac")))
print("matched_rmse", {k: d["matched_rmse"][k] for k in ("gpu", "oracle", "rel_diff_vs_oracle", "within_rtol")})
print("matched_rmse_sample", d.get("matched_rmse_sample"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "usable_cores", "facade_12_threads_20_bins")})
c = d["configs1"]; print("c1 value %.3e ms %.3f" % (c["value"], c["ms_per_step"]), c["matched_rmse"])
PY
