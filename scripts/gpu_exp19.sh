cd $GRAFT_REPO_ROOT; O=gpurun_out/exp19; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
export MFX_LIST_ORDER=2
for G in 12 16 24; do
  MFX_HOT_S_GAIN=$G run c1 12
  MFX_HOT_S_GAIN=$G run c1 20
  MFX_HOT_S_GAIN=$G run c2 12
  MFX_HOT_S_GAIN=$G run c2 20
  MFX_HOT_S_GAIN=$G run c2 8
  MFX_HOT_S_GAIN=$G run c2s 12
  MFX_HOT_S_GAIN=$G run 20000,10000,2000000,32 10
  MFX_HOT_S_GAIN=$G run 60000,30000,6000000,32 8
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-26s ep%2d %-48s %9.3f ms/epoch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse']))
"
