# Round profile of the hot path on the GPU box (run from the repo root through gpurun): the bench line, rocprofv3 kernel statistics of the
# same workload, and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no tracing beside the counters) that profiles/*_pmc_traffic.json
# is reduced from.   bash scripts/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && echo "bench done" &&
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $OUT/bench_under_prof.json 2> $OUT/bench_under_prof.err && echo "stats done" &&
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/FETCH_SIZE -o f -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/fetch.log 2>&1 && echo "fetch done" &&
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/WRITE_SIZE -o w -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/write.log 2>&1 && echo "write done" &&
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats182 -o s -- python3 scripts/run_steps.py --h 15600 --steps 20 --fused 4 > $OUT/res182.log 2>&1 && echo "resident stats done"
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT $OUT/stats | head -30
