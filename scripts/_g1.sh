cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 scripts/_soak_mr.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -6 | cut -c1-400
