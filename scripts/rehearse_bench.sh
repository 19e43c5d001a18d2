# bench.py as the driver launches it at N = 2 (torch.distributed.run, one rank per "GPU"), rehearsed on a one-GPU box: both ranks share
# device 0, so the numbers only bound the protocol overhead -- what is checked is that the line carries every structured field and that no
# failing transport or variant can lose it.   bash scripts/rehearse_bench.sh <tag> [mesh] [ranks]
set -o pipefail
TAG=${1:-rehearse}; MESH=${2:-10km}; N=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 5 --warmup 2 --mesh $MESH > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err
rc=$?
tail -c 3000 gpurun_out/${TAG}.json; echo; tail -5 gpurun_out/${TAG}.err
exit $rc
