cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 scripts/soak_multirank.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "^[0-9]" | cut -c1-330
