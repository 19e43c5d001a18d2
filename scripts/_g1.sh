cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident" > gpurun_out/r2_res1.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r2_res1.log
