# does the roofline timing cost anything on the refined mesh?  same box, sampling on (default) / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for ev in 4 1000000; do
    GMPNP_BENCH_SAMPLE_EVERY=$ev python bench.py --refine $1 --steps $2 --warmup 1 --no-cpu-baseline > gpurun_out/rsab_$ev.$rep.json 2> gpurun_out/rsab_$ev.$rep.err
    python -c "import json; d=json.load(open('gpurun_out/rsab_$ev.$rep.json')); r=d['roofline']; print('every $ev', $rep, 'its/s %.3f' % d['value'], 'mean us %.1f' % r['mean_launch_us'], 'sampled', r['launches_sampled'], 'of', r['launches_total'])"
  done
done
