# block-banded LU: old (abtest/lib_oldband.so, if present) against the tree's library, the band tests, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -f abtest/lib_oldband.so ]; then GMPNP_LIB=$PWD/abtest/lib_oldband.so timeout -k 10 300 python tools/band_lu_probe.py accuracy > gpurun_out/band_old.log 2>&1; fi
timeout -k 10 300 python tools/band_lu_probe.py accuracy newton > gpurun_out/band_new.log 2>&1 &&
timeout -k 10 600 python -m pytest tests -q -m gpu -k "band or direct or fall" -x > gpurun_out/band_tests.log 2>&1 &&
rm -rf gpurun_out/band_stats && rocprofv3 --kernel-trace --stats -d gpurun_out/band_stats --output-format csv -- python3 tools/band_lu_probe.py accuracy > gpurun_out/band_prof.log 2>&1 &&
cp $(find gpurun_out/band_stats -name "*kernel_stats.csv") gpurun_out/band_lu_probe_kernel_stats.csv; rm -rf gpurun_out/band_stats
for f in band_old band_new band_tests; do [ -f gpurun_out/$f.log ] && tail -n 4 gpurun_out/$f.log; done; true
