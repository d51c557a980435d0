#!/bin/bash
# kernel statistics of the multilevel bench at two refinements (tools/README.md)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_ml2
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ml2 -o ml2 --output-format csv -- python3 bench.py --refine 2 --steps 5 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/prof_ml2/bench.json 2> gpurun_out/prof_ml2/bench.err
find gpurun_out/prof_ml2 -name "*kernel_stats.csv" | head -3
cat gpurun_out/prof_ml2/bench.json | cut -c1-600
