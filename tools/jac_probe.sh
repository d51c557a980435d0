# kernel durations + memory-side traffic of the assembly kernels (development probe)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/jac_st gpurun_out/jac_pmc
rocprofv3 --kernel-trace --stats -d gpurun_out/jac_st --output-format csv -- python3 tools/kernel_times.py > gpurun_out/kt.log 2>&1
grep -E "jac_gather|scale_columns|k_element" $(find gpurun_out/jac_st -name "*kernel_stats.csv") | cut -d, -f1-4
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/jac_pmc --output-format csv -- python3 tools/kernel_times.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/jac_pmc k_jac_gather | grep mean
rm -rf gpurun_out/jac_pmc gpurun_out/jac_st
python -m pytest tests -m gpu -x -q -k "assembly or golden_elements or bitwise or full_size or pore_time_loop" 2>&1 | tail -2
