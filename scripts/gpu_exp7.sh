cd $GRAFT_REPO_ROOT; O=gpurun_out/exp7; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
# c1: users are the gathered side; hot threshold ~1600 ratings (multiplicity 2)
for S in u v uv; do
CASE_DROP_HOT=1600 CASE_DROP_SIDE=$S run c1 12
CASE_DROP_HOT=1600 CASE_DROP_SIDE=$S MFX_WGS_PER_XCD=1 run c1 12
done
for S in u v uv; do
CASE_DROP_HOT=2400 CASE_DROP_SIDE=$S run c2s 12
CASE_DROP_HOT=2400 CASE_DROP_SIDE=$S MFX_WGS_PER_XCD=1 run c2s 12
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-50s %9.3f ms/epoch rmse %.4f wg/cu %d hot %d tasks %d' % (d['case'], d['epochs'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['hot'], d['tasks']))
"
