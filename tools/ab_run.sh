# same-box A/B: for every abtest/lib_<name>.so given: bench line twice (interleaved), then rocprof kernel statistics once
# usage: bash tools/ab_run.sh base new ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    GMPNP_LIB=$PWD/abtest/lib_$v.so python bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$v.$rep.json 2>gpurun_out/ab_$v.$rep.err
    python -c "import json,sys; d=json.load(open('gpurun_out/ab_$v.$rep.json')); print('$v', $rep, 'its/s %.1f' % d['value'], 'krylov', d['config']['krylov_iterations'], 'event us %.2f' % d['roofline']['mean_launch_us'])"
  done
done
for v in "$@"; do
  rm -rf gpurun_out/ab_stats
  GMPNP_LIB=$PWD/abtest/lib_$v.so rocprofv3 --kernel-trace --stats -d gpurun_out/ab_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/ab_$v.rocprof.err
  cp $(find gpurun_out/ab_stats -name "*kernel_stats.csv") gpurun_out/ab_$v.kernel_stats.csv; rm -rf gpurun_out/ab_stats
  echo "== $v"; head -3 gpurun_out/ab_$v.kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
