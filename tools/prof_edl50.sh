#!/bin/bash
# rocprofv3 kernel statistics of the 1D case (bench.py --case edl50)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_edl50
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_edl50 -o e --output-format csv -- python3 bench.py --case edl50 --no-cpu-baseline > gpurun_out/prof_edl50.json 2> gpurun_out/prof_edl50.err
cut -d, -f1-4,6 gpurun_out/prof_edl50/e_kernel_stats.csv | head -16 | cut -c1-160
