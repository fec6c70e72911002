cd $GRAFT_REPO_ROOT; O=gpurun_out/exp10; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
run c2 8
for S in 16 32 48 64; do run c2 8 owner_side=1 stripes=$S; done
for S in 32 64; do run c2 8 stripes=$S; done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-34s %9.3f ms/epoch %8.1f us/launch rmse %.4f wg/cu %d tasks %d hot %d' % (d['case'], d['epochs'], d['opts'], d['ms_epoch'], d['us_launch'], d['rmse'], d['wg_per_cu'], d['tasks'], d['hot']))
"
