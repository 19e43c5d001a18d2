# scratch runner for one-off GPU experiments: bash scripts/gpu_cmd.sh <tag> <command...>   (stdout+stderr -> gpurun_out/<tag>.log)
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 "$@" > gpurun_out/${TAG}.log 2>&1
rc=$?
tail -25 gpurun_out/${TAG}.log
exit $rc
