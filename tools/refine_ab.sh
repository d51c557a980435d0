# same-box comparison of libraries on the refined meshes: bash tools/refine_ab.sh <refine> <steps> name...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lev=$1; steps=$2; shift; shift
for rep in 1 2; do
  for v in "$@"; do
    GMPNP_BENCH_SAMPLE_EVERY=1000000 GMPNP_LIB=$PWD/abtest/lib_$v.so python bench.py --refine $lev --steps $steps --warmup 1 --no-cpu-baseline > gpurun_out/rab_$v.$rep.json 2> gpurun_out/rab_$v.$rep.err
    python -c "import json; d=json.load(open('gpurun_out/rab_$v.$rep.json')); print('$v', $rep, 'refine $lev its/s %.3f' % d['value'], 'krylov', d['config']['krylov_iterations'])"
  done
done
