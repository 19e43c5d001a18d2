cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "registers" 2>&1 | grep -v amdgpu.ids | tail -2
