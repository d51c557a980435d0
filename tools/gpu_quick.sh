# quick GPU check of a kernel change: Krylov / Newton parity tests, bench line, kernel statistics of the bench
set -x
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_window or newton or linear or krylov or spmv or variants or handover or two_handles" > gpurun_out/q_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/q_tests.log
python bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/q_bench.json 2>gpurun_out/q_bench.err; python -c "import json; d=json.load(open('gpurun_out/q_bench.json')); print('bench', d['value'], d['config']['krylov_iterations'], d['roofline']['mean_launch_us'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/q_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/q_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/q_bench_rocprof.json 2> gpurun_out/q_b2.err
cp $(find gpurun_out/q_stats -name "*kernel_stats.csv") gpurun_out/q_kernel_stats.csv; rm -rf gpurun_out/q_stats
head -8 gpurun_out/q_kernel_stats.csv | cut -c1-150
