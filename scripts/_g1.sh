cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NXS_BENCH_TRY_RESIDENT=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 20 --warmup 3 --mesh 10km > gpurun_out/r2b_bench_g2_10km.json 2> gpurun_out/r2b_bench_g2_10km.err; echo "rc $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r2b_bench_g2_10km.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['halo_transport'], d['phases_ms'], d['fields_ok'])
"
timeout -k 10 300 python3 bench.py --mesh 10km --no-cpu-baseline > gpurun_out/r2b_bench_10km.json 2>/dev/null; echo "rc $?"; python3 -c "
import json
d=json.loads(open('gpurun_out/r2b_bench_10km.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['traffic_source'])
"
