#!/bin/bash
# kernel statistics of one partition over the peer transport (exchange form $1, default 0)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
form=${1:-0}
export GMPNP_BENCH_BACKEND=gloo GMPNP_BENCH_TRANSPORTS=peer GMPNP_BENCH_EXCHANGE_FORM=$form
rm -rf gpurun_out/xch_prof_$form
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/xch_prof_$form -o x --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-edl50 --force-partitioned > gpurun_out/xch_prof_$form.json 2> gpurun_out/xch_prof_$form.err || { tail -5 gpurun_out/xch_prof_$form.err; exit 1; }
grep -E "k_half|k_dist_reduce_exchange" gpurun_out/xch_prof_$form/x_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
