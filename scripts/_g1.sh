cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r2_prof
rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $R/bench_under_prof.json 2> $R/bench_under_prof.err; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc/FETCH_SIZE -o f -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $R/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc/WRITE_SIZE -o w -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $R/write.log 2>&1; echo "write rc=$?"
find $R -name "*.csv" | head -20
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err; echo "bench rc=$?"
