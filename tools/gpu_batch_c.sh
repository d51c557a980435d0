# second half of the CPU window, then the large-mesh paths after the pre-reduction / gather-run changes
bash tools/gpu_batch_b.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "refined or handover or partition or assembly or variants or window" > gpurun_out/r2_t8.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2_t8.log
for lev in 1 2; do
  rm -rf gpurun_out/ref$lev
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ref$lev --output-format csv -- python3 tools/refined_roofline.py $lev > gpurun_out/refined_level${lev}_after.log 2>&1
  cp $(find gpurun_out/ref$lev -name "*kernel_stats.csv") gpurun_out/refined_level${lev}_kernel_stats_after.csv
  rm -rf gpurun_out/ref$lev
  grep -v "^W2026\|^E2026" gpurun_out/refined_level${lev}_after.log | head -10
done
