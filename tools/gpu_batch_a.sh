# refined-mesh kernel statistics (levels 0, 1, 2) + first half of the CPU window on the GPU box's host
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lev in 1 2; do
  rm -rf gpurun_out/ref$lev
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/ref$lev --output-format csv -- python3 tools/refined_roofline.py $lev > gpurun_out/refined_level$lev.log 2>&1
  cp $(find gpurun_out/ref$lev -name "*kernel_stats.csv") gpurun_out/refined_level${lev}_kernel_stats.csv
  rm -rf gpurun_out/ref$lev
  tail -12 gpurun_out/refined_level$lev.log
done
mkdir -p gpurun_out/ckpt
timeout -k 10 780 python tools/cpu_window.py --steps 25 --threads 1 --save gpurun_out/ckpt/cpu_window_half.npz --out gpurun_out/cpu_window_first25.json > gpurun_out/cpu_window_a.log 2>&1
tail -3 gpurun_out/cpu_window_a.log
