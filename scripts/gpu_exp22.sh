cd $GRAFT_REPO_ROOT; O=gpurun_out/exp22; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for G in 2 4 6 8; do
  for i in 1 2 3; do MFX_HOT_S_GAIN=$G run c1 20; done
  MFX_HOT_S_GAIN=$G run c1 12
  for i in 1 2; do MFX_HOT_S_GAIN=$G run c2 12; done
  MFX_HOT_S_GAIN=$G run c2 20
  MFX_HOT_S_GAIN=$G run c2s 12
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-24s %9.3f ms/epoch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse']))
"
