cd $GRAFT_REPO_ROOT; O=gpurun_out/exp16; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
run c2 6
CASE_DROP_HOT=100000 CASE_DROP_SIDE=u run c2 6
CASE_DROP_HOT=12000 CASE_DROP_SIDE=u run c2 6
CASE_DROP_HOT=12000 CASE_DROP_SIDE=v run c2 6
CASE_DROP_HOT=12000 CASE_DROP_SIDE=uv run c2 6
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %9.3f ms/epoch  %.3e ratings/s rmse %.4f hot %d' % (d['case'], d['epochs'], d['ms_epoch'], d['ratings_per_s'], d['rmse'], d['hot']))
"
