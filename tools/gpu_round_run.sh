# The round's measurement run on the GPU box (everything under profiles/<round>/ comes from here):
#   full -m gpu suite, bench line, rocprofv3 kernel statistics of the bench, the two PMC passes (program directly after --)
set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2_gputests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r2_stats gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write
rocprofv3 --kernel-trace --stats -d gpurun_out/r2_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_under_rocprof.json 2> gpurun_out/r2_b2.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2_pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2_pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/r2_pmc_fetch > gpurun_out/r2_pmc_fetch.json
python tools/pmc_summary.py gpurun_out/r2_pmc_write > gpurun_out/r2_pmc_write.json
cp $(find gpurun_out/r2_stats -name "*kernel_stats.csv") gpurun_out/r2_bench_kernel_stats.csv
rm -rf gpurun_out/r2_stats gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write
python bench.py --steps 50 --warmup 2 > gpurun_out/r2_bench.json 2>gpurun_out/r2_bench.err; cat gpurun_out/r2_bench.json
python -c "import __graft_entry__ as g; g.smoke()"
