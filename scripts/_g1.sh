cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NXS_BENCH_TRY_RESIDENT=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 --mesh 10km > gpurun_out/r2_bench_g2c.json 2> gpurun_out/r2_bench_g2c.err; echo "g2 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_g2c.json')); print(d['ms_per_step'], d['config']['halo_transport']); print(d['roofline']['kernel'][:60], d['phases_ms'])"
tail -3 gpurun_out/r2_bench_g2c.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r2_bench_g2d.json 2> gpurun_out/r2_bench_g2d.err; echo "g2 2km rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_g2d.json')); print(d['ms_per_step'], d['config']['halo_transport'], d['phases_ms'])"
