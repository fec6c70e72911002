cd $GRAFT_REPO_ROOT; O=gpurun_out/exp4; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep CASE >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for C in "c2s 12" "c2 8" "c1 12"; do
  set -- $C
  for H in 64 128 256 512 1024 4096; do
    MFX_HOT_LEN=$H run $1 $2
    MFX_HOT_LEN=$H MFX_HOT_LWW=1 run $1 $2
  done
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-50s %9.3f ms/epoch rmse %.4f wg/cu %d hot %d tasks %d' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['hot'], d['tasks']))
"
