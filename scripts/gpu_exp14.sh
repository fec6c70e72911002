cd $GRAFT_REPO_ROOT; O=gpurun_out/exp14; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
C=20000,10000,2000000,32
for G in 4 16 128; do
  MFX_HOT_S_GAIN=$G run $C 10
  MFX_HOT_S_GAIN=$G run $C 10
  MFX_HOT_S_GAIN=$G MFX_WGS_PER_XCD=1 run $C 10
done
MFX_CONFLICT_DIV=24 run $C 10
MFX_CONFLICT_DIV=8 run $C 10
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-26s ep%2d %-50s %9.3f ms/epoch rmse %.4f wg/cu %d tasks %d hot %d' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['tasks'], d['hot']))
"
