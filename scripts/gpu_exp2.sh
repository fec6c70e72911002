# What makes the GPU path fit faster than the oracle at configs[2] density?  Concurrency vs order (round 2, experiment 2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/exp2; mkdir -p $O; rm -f $O/log.txt
run() { timeout -k 10 300 python3 scripts/gpu_case.py "$@" 2>&1 | grep CASE >> $O/log.txt || echo "FAILED $*" >> $O/log.txt; }
for C in "c2s 12" "c1 12"; do
  set -- $C
  run $1 $2
  MFX_WGS_PER_XCD=1 run $1 $2
  MFX_WGS_PER_XCD=4 run $1 $2
  MFX_WGS_PER_XCD=16 run $1 $2
  MFX_HOT_LEN=100000000 run $1 $2
  MFX_HOT_LEN=100000000 MFX_WGS_PER_XCD=1 run $1 $2
  MFX_WGS_PER_XCD=1 run $1 $2 identity_maps=2
  MFX_WGS_PER_XCD=1 run $1 $2 rk_mode=1
done
cat $O/log.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('CASE'): print(l.strip()); continue
    d = json.loads(l[5:]); print('%-5s ep%2d %-22s %-50s %9.3f ms/epoch rmse %.4f wg/cu %d hot %d tasks %d' % (d['case'], d['epochs'], d['opts'], d['env'], d['ms_epoch'], d['rmse'], d['wg_per_cu'], d['hot'], d['tasks']))
"
