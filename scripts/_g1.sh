cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_interp.py tests/test_remap.py tests/test_regrid_cycle.py -m gpu -x -q -s > gpurun_out/r2_interp1.log 2>&1; echo "rc=$?" ; tail -25 gpurun_out/r2_interp1.log
