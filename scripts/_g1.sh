cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 scripts/soak_resident.py 100 2>&1 | grep -v amdgpu.ids | tail -6
