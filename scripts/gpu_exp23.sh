cd $GRAFT_REPO_ROOT; O=gpurun_out/exp23; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep -E "CASE|dropped" >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for S in 8 16; do
  run c2 12 stripes=$S
  run c2 20 stripes=$S
  run c1 12 stripes=$S
  run c1 20 stripes=$S
  run c1 20 stripes=$S
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-6s ep%2d %-18s %9.3f ms/epoch rmse %.4f wg/cu %d' % (d['case'], d['epochs'], d['opts'], d['ms_epoch'], d['rmse'], d['wg_per_cu']))
"
