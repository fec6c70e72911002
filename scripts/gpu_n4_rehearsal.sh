# Four ranks on ONE GPU over gloo (the rehearsal path of the ring), configs[1] per rank: final RMSE vs slots per rank.
cd $GRAFT_REPO_ROOT
for C in 2 1; do
  timeout -k 10 400 python bench.py --gpus 4 --same-device --backend gloo --config c1 --steps 16 --warmup 3 --slots-per-rank $C 2>gpurun_out/n4_c$C.err | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('N=%d slots/rank %s stripes %s final_rmse %.4f after %d epochs, %.2f ms/step' % (d['n_gpus'], d['config']['slots_per_rank'], d['config']['stripes'], d['final_rmse'], d['epochs_trained'], d['ms_per_step']))
"
done
