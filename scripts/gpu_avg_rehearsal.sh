# Four ranks on ONE GPU over gloo, configs[1] per rank, 20 epochs: replicas of Q combined by mean (avg), by rating-count-weighted
# mean (wavg), and the slot rotation, 1 and 8 exchanges per epoch.  gpurun_out/avg_rehearsal.log
cd $GRAFT_REPO_ROOT
L=gpurun_out/avg_rehearsal.log; : > $L
for C in "avg 1" "wavg 1" "avg 8" "wavg 8" "rotate 1"; do
  set -- $C
  timeout -k 10 400 python bench.py --gpus 4 --same-device --backend gloo --config c1 --steps 16 --warmup 3 --combine $1 --syncs-per-epoch $2 2>gpurun_out/avg_$1_$2.err | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('N=%d combine %-6s exchanges/epoch %s final_rmse %.4f after %d epochs, %.2f ms/step' % (d['n_gpus'], d['config']['combine'], d['config']['exchanges_per_epoch'], d['final_rmse'], d['epochs_trained'], d['ms_per_step']))
" >> $L || exit 1
done
cat $L
