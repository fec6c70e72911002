cd $GRAFT_REPO_ROOT; O=gpurun_out/exp13; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for T in 1 2 3 4 6 8; do MFX_ONE_TASK=$T run c2 6; done
MFX_ONE_TASK=0 run c2 6
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-26s %9.3f ms/epoch %8.1f us/launch rmse %.4f tasks %d pad %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['us_launch'], d['rmse'], d['tasks'], d['pad']))
"
