#!/bin/bash
# exchange form A/B of the peer transport: multi-process parity tests, then one partition on an uncontended GPU (launch cost only)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "peer_mailbox_transport_between_processes" -s 2>&1 | tail -15 > gpurun_out/xch_tests.log
cat gpurun_out/xch_tests.log
grep -q "passed" gpurun_out/xch_tests.log || exit 1
for form in 0 1; do
  GMPNP_BENCH_BACKEND=gloo GMPNP_BENCH_EXCHANGE_FORM=$form GMPNP_BENCH_TRANSPORTS=peer timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-edl50 --force-partitioned > gpurun_out/xch_w1_$form.json 2> gpurun_out/xch_w1_$form.err || { tail -5 gpurun_out/xch_w1_$form.err; exit 1; }
  python -c "import json; d=json.loads([l for l in open('gpurun_out/xch_w1_$form.json') if l.startswith('{')][-1]); r=d['partitioned_rehearsal']; print('form env $form ->', r.get('exchange_form'), 'single-GPU %.1f its/s |' % d['value'], 'one partition over peer: %.1f its/s, %.1f us per BiCGStab iteration' % (r['value'], 1e6*r['seconds']/r['krylov_iterations']) if 'value' in r else r)"
done
