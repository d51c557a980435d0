set -x
python -m pytest tests -m gpu -x -q -k "not bench_window" > gpurun_out/r2_t2.log 2>&1; tail -5 gpurun_out/r2_t2.log
python tools/stern_schedule.py gpurun_out/stern_schedule.json > gpurun_out/stern_schedule.log 2>&1; tail -40 gpurun_out/stern_schedule.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r2_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/r2_b2_under_rocprof.json 2> gpurun_out/r2_b2.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2_pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2_pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/r2_pmc_fetch > gpurun_out/r2_pmc_fetch.json
python tools/pmc_summary.py gpurun_out/r2_pmc_write > gpurun_out/r2_pmc_write.json
find gpurun_out/r2_stats -name "*kernel_stats.csv" | head; find gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write -name "*.csv" -size +1M -delete; find gpurun_out/r2_stats -name "*kernel_trace.csv" -delete
python bench.py --steps 50 --warmup 2 > gpurun_out/r2_b2.json 2>gpurun_out/r2_b2b.err; cat gpurun_out/r2_b2.json
