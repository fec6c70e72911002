cd $GRAFT_REPO_ROOT; O=gpurun_out/exp24; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for V in "6 0" "3 16" "2 4" "4 32"; do
  set -- $V
  export MFX_HOT_S_GAIN=$1 MFX_HOT_S_N0=$2
  run c2 12; run c2 12; run c2 20; run c2 8
  run c1 12; run c1 20; run c1 20; run c1 20
  run c2s 12
  run 20000,10000,2000000,32 10
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-26s ep%2d %-44s %9.3f ms/epoch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse']))
"
