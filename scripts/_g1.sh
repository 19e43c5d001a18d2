cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 800 python3 scripts/rehearse_rank_of_eight.py 12 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "^overlap" | cut -c1-200
