# GPU parity suite on the GPU box, C-level stderr kept in the log (--capture=sys leaves fd 2 alone, so a message from the C++ runtime,
# glibc or the HSA runtime in front of an abort is not swallowed with the dying process).   bash scripts/gpu_suite.sh <tag> [pytest args]
set -o pipefail
TAG=${1:-suite}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --capture=sys -p no:cacheprovider "$@" > gpurun_out/${TAG}.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}.log
exit $rc
