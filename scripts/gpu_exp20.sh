cd $GRAFT_REPO_ROOT; O=gpurun_out/exp20; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for i in 1 2; do for L in 0 2 1; do MFX_LIST_ORDER=$L run c2 6; MFX_LIST_ORDER=$L run c1 8; done; done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-28s %9.3f ms/epoch %8.1f us/launch rmse %.4f' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['us_launch'], d['rmse']))
"
