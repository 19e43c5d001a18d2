cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r2_mr3.log 2>&1; echo "mr rc=$?" ; tail -5 gpurun_out/r2_mr3.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r2_bench_g2b.json 2> gpurun_out/r2_bench_g2b.err; echo "g2 rc=$?"; cat gpurun_out/r2_bench_g2b.json | cut -c1-300
