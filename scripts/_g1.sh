cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -m gpu -k "resident" 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do
echo "2 WG per CU:"; timeout -k 10 800 python3 scripts/rehearse_rank_of_eight.py 12 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "^overlap . rank 0" | cut -c1-160
done
for i in 1 2; do timeout -k 10 200 python3 scripts/run_steps.py --h 15600 --steps 200 --fused 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-130; done
timeout -k 10 200 python3 scripts/run_steps.py --mesh 10km --steps 200 --fused 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-130
