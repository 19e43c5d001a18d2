cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_suite3.log 2>&1; echo "suite rc=$?" ; tail -5 gpurun_out/r2_suite3.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r2_bench3.err
