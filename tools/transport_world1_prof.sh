cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tr in peer rccl; do
rm -rf gpurun_out/w1_stats
GMPNP_BENCH_TRANSPORTS=$tr rocprofv3 --kernel-trace --stats -d gpurun_out/w1_stats --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --force-partitioned > /dev/null 2> gpurun_out/w1_prof.err
cp $(find gpurun_out/w1_stats -name "*kernel_stats.csv") gpurun_out/w1_${tr}_kernel_stats.csv; rm -rf gpurun_out/w1_stats
echo "== $tr"; head -14 gpurun_out/w1_${tr}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-130
done
