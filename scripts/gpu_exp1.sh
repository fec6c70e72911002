# RMSE / time vs stripe count and owner side (round 2, experiment 1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/exp1; mkdir -p $O
for C in "c1 12" "c2s 12" "c2 8"; do
  set -- $C
  for V in "" "stripes=16" "stripes=24" "owner_side=1" "owner_side=1 stripes=16"; do
    timeout -k 10 300 python3 scripts/gpu_case.py $1 $2 $V 2>&1 | grep CASE >> $O/log.txt || echo "FAILED $C $V" >> $O/log.txt
  done
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-28s %8.3f ms/epoch %7.1f us/launch rmse %.4f stripes %d wg/cu %d hot %d pad %.3f ownerQ %d' % (d['case'], d['epochs'], d['opts'], d['ms_epoch'], d['us_launch'], d['rmse'], d['stripes'], d['wg_per_cu'], d['hot'], d['pad'], d['owner_is_q']))
"
