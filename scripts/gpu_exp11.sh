cd $GRAFT_REPO_ROOT; O=gpurun_out/exp11; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
C=1000000,31256,6250000,64
run $C 8
run $C 8 wg_per_cu=2
run $C 8 wg_per_cu=3
MFX_ONE_TASK=2 run $C 8
MFX_HOT_LEN=128 run $C 8
MFX_HOT_LWW=1 run $C 8
run $C 8 stripes=4
run $C 8 owner_side=1
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-22s %-26s %9.3f ms/epoch %8.1f us/launch rmse %.4f wg/cu %d stripes %d tasks %d hot %d' % ('sub8', d['epochs'], d['opts'], d['env'], d['ms_epoch'], d['us_launch'], d['rmse'], d['wg_per_cu'], d['stripes'], d['tasks'], d['hot']))
"
