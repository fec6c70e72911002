cd $GRAFT_REPO_ROOT; O=gpurun_out/exp8; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for W in 2 3 4 5; do run c2 8 wg_per_cu=$W; done
for D in 8 10 12 16; do MFX_CONFLICT_DIV=$D run c1 12; done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-18s %-30s %9.3f ms/epoch rmse %.4f wg/cu %d hot %d tasks %d' % (d['case'], d['epochs'], d['opts'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['hot'], d['tasks']))
"
