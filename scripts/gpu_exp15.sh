cd $GRAFT_REPO_ROOT; O=gpurun_out/exp15; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for F in 1 0; do
  MFX_FOLD_MODE=$F run c1 12
  MFX_FOLD_MODE=$F run c1 12
  MFX_FOLD_MODE=$F run c2s 12
  MFX_FOLD_MODE=$F run c2 12
  MFX_FOLD_MODE=$F run c2 8
  MFX_FOLD_MODE=$F run 60000,30000,6000000,32 8
  MFX_FOLD_MODE=$F run 20000,10000,2000000,32 10
  MFX_FOLD_MODE=$F run 20000,10000,2000000,64 6
  MFX_FOLD_MODE=$F run 5000,4000,400000,128 5
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-26s ep%2d %-24s %9.3f ms/epoch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse']))
"
