cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_suite4.log 2>&1; echo "suite rc=$?" ; tail -3 gpurun_out/r2_suite4.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err; echo "bench rc=$?"
