cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29523 bench.py --gpus 4 --steps 3 --warmup 1 > gpurun_out/r2b_bench_g4.json 2> gpurun_out/r2b_bench_g4.err; echo "rc $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r2b_bench_g4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['n_gpus'], d['ranks_share_device'], d['config']['halo_transport'][:300], d['fields_ok'])
"
