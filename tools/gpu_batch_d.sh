# large-mesh paths after the workgroup-per-output reduction: tests, refined-mesh kernel statistics, bench on refined meshes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "refined or partition or rccl or bench_partitioned or config3" > gpurun_out/r2_t9.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2_t9.log
for lev in 1 2; do
  rm -rf gpurun_out/ref$lev
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ref$lev --output-format csv -- python3 tools/refined_roofline.py $lev > gpurun_out/refined_level${lev}_after.log 2>&1
  cp $(find gpurun_out/ref$lev -name "*kernel_stats.csv") gpurun_out/refined_level${lev}_kernel_stats_after.csv
  rm -rf gpurun_out/ref$lev
  grep -v "^W2026\|^E2026" gpurun_out/refined_level${lev}_after.log | head -4
  grep "k_dist_reduce\|k_minv_apply" gpurun_out/refined_level${lev}_kernel_stats_after.csv | cut -d, -f1-4 | cut -c1-120
done
python bench.py --refine 1 --steps 10 --warmup 1 --no-cpu-baseline > gpurun_out/bench_refine1.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/bench_refine1.json')); print('refine1', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['krylov_iterations'])"
python bench.py --refine 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_refine2.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/bench_refine2.json')); print('refine2', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['krylov_iterations'])"
