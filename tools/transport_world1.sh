# One rank (the whole mesh as ONE partition) on an uncontended GPU: what the partitioned iteration costs per transport when the
# collectives have nobody to wait for — launches and library calls only.  bench.py --force-partitioned, steps 0..19.
for tr in peer rccl host; do
  GMPNP_BENCH_TRANSPORTS=$tr python bench.py --steps 20 --warmup 2 --no-cpu-baseline --force-partitioned > gpurun_out/w1_$tr.json 2> gpurun_out/w1_$tr.err
  python -c "import json; d=json.loads([l for l in open('gpurun_out/w1_$tr.json') if l.startswith('{')][-1]); r=d['partitioned_rehearsal']; print('$tr', 'single-GPU solver %.1f its/s |' % d['value'], 'one partition over $tr: %.1f its/s, %.1f us per BiCGStab iteration' % (r['value'], 1e6*r['seconds']/r['krylov_iterations']) if 'value' in r else r)"
done
