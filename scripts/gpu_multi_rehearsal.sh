# N=4 rehearsal of bench.py on ONE GPU (gloo, host-staged exchange): rotation vs averaging, final RMSE.
# Throughput here means nothing (4 processes share one GPU and stage through the host).
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
PARSE='import sys,json
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("N=%d combine %s syncs/epoch %s rmse %.4f epochs %d ms/step %.2f" % (d["n_gpus"], d["config"].get("combine"), d["config"].get("syncs_per_epoch"), d["final_rmse"], d["epochs_trained"], d["ms_per_step"]))'
python bench.py --steps 16 --warmup 3 --no-cpu-baseline | python -c "$PARSE"
P=29700
for N in 2 4; do P=$((P+1))
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P bench.py --gpus $N --steps 16 --warmup 3 --backend gloo --same-device --combine rotate 2> gpurun_out/multi_rot$N.err | python -c "$PARSE"
done
tail -3 gpurun_out/multi_rot4.err
