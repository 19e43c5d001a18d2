cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 800 python3 scripts/rehearse_rank_of_eight.py 12 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "^overlap 0" | cut -c1-160
done
timeout -k 10 200 python3 scripts/run_steps.py --h 15600 --steps 200 --fused 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-130
