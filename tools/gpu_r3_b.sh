set -x
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "partition or peer or bench or rccl or config3 or driver_outputs or steric" > gpurun_out/r3_parttests.log 2>&1; echo rc=$?; tail -5 gpurun_out/r3_parttests.log
for mode in 0 1; do
  GMPNP_BENCH_PEER_SEPARATE=$mode GMPNP_BENCH_TRANSPORTS=peer timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-edl50 --force-partitioned > gpurun_out/w1_peer_$mode.json 2> gpurun_out/w1_peer_$mode.err
  python -c "import json; d=json.loads([l for l in open('gpurun_out/w1_peer_$mode.json') if l.startswith('{')][-1]); r=d['partitioned_rehearsal']; print('separate=$mode', 'single-GPU solver %.1f its/s |' % d['value'], 'one partition over peer: %.1f its/s, %.1f us per BiCGStab iteration' % (r['value'], 1e6*r['seconds']/r['krylov_iterations']) if 'value' in r else r)"
done
