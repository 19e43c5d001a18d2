cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "resident" > gpurun_out/r2_res2.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r2_res2.log
