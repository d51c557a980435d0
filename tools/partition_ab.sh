# same-box comparison of two libraries on the partitioned solve with ONE rank (whole mesh as one partition, peer transport)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    GMPNP_LIB=$PWD/abtest/lib_$v.so GMPNP_BENCH_TRANSPORTS=peer python bench.py --steps 50 --warmup 2 --no-cpu-baseline --force-partitioned > gpurun_out/pab_$v.$rep.json 2> gpurun_out/pab_$v.$rep.err
    python -c "import json; d=json.loads([l for l in open('gpurun_out/pab_$v.$rep.json') if l.startswith('{')][-1]); r=d['partitioned_rehearsal']; print('$v', $rep, 'single-GPU %.1f its/s |' % d['value'], 'one partition: %.1f its/s, newton %d krylov %d, %.1f us per BiCGStab iteration' % (r['value'], r['newton_iterations'], r['krylov_iterations'], 1e6*r['seconds']/r['krylov_iterations']) if 'value' in r else r)"
  done
done
