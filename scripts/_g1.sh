cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_remap.py tests/test_interp.py tests/test_regrid_cycle.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
