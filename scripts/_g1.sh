cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 scripts/rehearse_rank_of_eight.py 12 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -8
