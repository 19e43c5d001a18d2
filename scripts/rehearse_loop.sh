cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout -k 10 200 python3 scripts/rehearse_rank_of_eight.py 8 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep "ok False\|error\|Error" | cut -c1-900; rc=${PIPESTATUS[0]}; if [ $rc -eq 124 ]; then exit 1; fi; echo "run $i done"; done
