cd $GRAFT_REPO_ROOT; O=gpurun_out/exp17; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for G in 1 4 16 32 128; do
  MFX_HOT_S_GAIN=$G run c1 20
  MFX_HOT_S_GAIN=$G run c2 20
  MFX_HOT_S_GAIN=$G run c2s 20
done
MFX_HOT_LWW=1 run c1 20
MFX_HOT_LWW=1 run c2 20
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-28s %9.3f ms/epoch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse']))
"
