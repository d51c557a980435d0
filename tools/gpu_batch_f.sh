python -m pytest tests -m gpu -x -q -k "variants or refined" > gpurun_out/r2_t10.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2_t10.log
python tools/refined_roofline.py 2 > gpurun_out/refined_level2_mat.log 2>&1; cat gpurun_out/refined_level2_mat.log
python bench.py --refine 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_refine2_mat.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/bench_refine2_mat.json')); print('refine2', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['krylov_iterations'], d['roofline']['mean_launch_us'])"
