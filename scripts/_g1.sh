cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_golden.py tests/test_regrid_cycle.py -m gpu -x -q > gpurun_out/r2_parity5.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_parity5.log
for m in 10km 2km; do python scripts/run_steps.py --mesh $m --steps 60 --torch-first 2>&1 | tail -1 | cut -c1-120; done
