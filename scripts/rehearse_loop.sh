# the rehearsal of a rank of eight, repeated: how often does the shared-device harness fail, and with which code?   bash scripts/rehearse_loop.sh [mesh] [runs]
KIND=${1:-h16000}; RUNS=${2:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq 1 $RUNS); do timeout -k 10 200 python3 scripts/rehearse_rank_of_eight.py 8 $KIND 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "^overlap\|gave up\|did not arrive" | cut -c1-330; rc=${PIPESTATUS[0]}; if [ $rc -eq 124 ]; then exit 1; fi; echo "run $i done"; done
