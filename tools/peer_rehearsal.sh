# N ranks sharing the box's ONE GPU (gloo rendezvous): bench.py's partitioned phase over the peer-mailbox transport and over the
# host-staged one, 2 and 4 ranks.  (All ranks time-slice one card: the rates say what a collective costs, not how N GPUs scale.)
export GMPNP_BENCH_BACKEND=gloo
for n in 2 4; do for tr in peer host; do
  GMPNP_BENCH_TRANSPORTS=$tr python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --steps 10 --warmup 1 > gpurun_out/reh_${n}_$tr.json 2> gpurun_out/reh_${n}_$tr.err
  python -c "import json; d=json.loads([l for l in open('gpurun_out/reh_${n}_$tr.json') if l.startswith('{')][-1]); print($n, '$tr', 'partitioned its/s %.1f' % d['value'], d['partitioned'], 'krylov', d['config']['krylov_iterations'], 'replicas its/s %.1f' % d['replicas']['value'])"
done; done
