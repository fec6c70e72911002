# N=4 rehearsal of bench.py on ONE GPU (gloo, host-staged exchange): which way of combining the replicas of Q keeps
# the final RMSE?  Throughput here means nothing.
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
PARSE='import sys,json
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("N=%d combine %s syncs/epoch %s rmse %.4f epochs %d" % (d["n_gpus"], d["config"].get("combine"), d["config"].get("syncs_per_epoch"), d["final_rmse"], d["epochs_trained"]))'
python bench.py --steps 16 --warmup 3 --no-cpu-baseline | python -c "$PARSE"
P=29600
for C in sum hybrid; do for S in 1 2 8; do P=$((P+1))
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 4 --steps 16 --warmup 3 --backend gloo --same-device --syncs-per-epoch $S --combine $C 2> gpurun_out/multi4_$C$S.err | python -c "$PARSE"
done; done
