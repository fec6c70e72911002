cd $GRAFT_REPO_ROOT; O=gpurun_out/exp6; mkdir -p $O; rm -f $O/log.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep CASE >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for C in "c1 12" "c2s 12" "c2 12" "c2 8"; do
  set -- $C
  run $1 $2
  run $1 $2
  MFX_FOLD_MODE=1 run $1 $2
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-22s %-50s %9.3f ms/epoch rmse %.4f wg/cu %d hot %d tasks %d' % (d['case'], d['epochs'], d['opts'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['hot'], d['tasks']))
"
timeout -k 10 300 python3 tests/tools/gpu_hot_rows.py > $O/hot_rows.log 2>&1; grep -A1 "gpu fold" $O/hot_rows.log | head -4
