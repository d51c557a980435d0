# Round 3, final measurement run on the GPU box (everything under profiles/r03/ that names the final build comes from here):
# full -m gpu suite, smoke, bench line, rocprofv3 kernel statistics of the bench, the two PMC passes (program directly after --),
# refined-mesh lines with and without the multilevel term, the 1D case as a headline line.
set -x
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3f_gputests.log 2>&1; echo rc=$?; tail -4 gpurun_out/r3f_gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3f_stats gpurun_out/r3f_pmc_fetch gpurun_out/r3f_pmc_write
rocprofv3 --kernel-trace --stats -d gpurun_out/r3f_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_bench_under_rocprof.json 2> gpurun_out/r3f_b2.err
cp $(find gpurun_out/r3f_stats -name "*kernel_stats.csv") gpurun_out/r3f_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r3f_pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r3f_pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/r3f_pmc_fetch > gpurun_out/r3f_pmc_fetch.json
python tools/pmc_summary.py gpurun_out/r3f_pmc_write > gpurun_out/r3f_pmc_write.json
rm -rf gpurun_out/r3f_stats gpurun_out/r3f_pmc_fetch gpurun_out/r3f_pmc_write
# the traffic file of THIS build first (in the box's copy of the tree; tools/make_traffic_json.py is run again on the merged files
# at home), then the bench line that carries it
python tools/make_traffic_json.py gpurun_out/r3f_pmc_fetch.json gpurun_out/r3f_pmc_write.json r03
timeout -k 10 400 python bench.py --steps 50 --warmup 2 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err; echo bench rc=$?
timeout -k 10 300 python bench.py --refine 2 --no-multilevel --steps 10 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_bench_refine2.json 2>> gpurun_out/r3f_bench.err
timeout -k 10 300 python bench.py --refine 2 --multilevel --steps 10 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_bench_refine2_ml.json 2>> gpurun_out/r3f_bench.err
timeout -k 10 400 python bench.py --refine 3 --multilevel --steps 3 --warmup 1 --no-cpu-baseline --no-edl50 > gpurun_out/r3f_bench_refine3_ml.json 2>> gpurun_out/r3f_bench.err
timeout -k 10 200 python bench.py --case edl50 > gpurun_out/r3f_bench_edl50.json 2>> gpurun_out/r3f_bench.err
python - <<'PY'
import json
for f in ("r3f_bench", "r3f_bench_under_rocprof", "r3f_bench_refine2", "r3f_bench_refine2_ml", "r3f_bench_refine3_ml", "r3f_bench_edl50"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f)); r = d["roofline"]
        print(f, "its/s %.2f" % d["value"], "newton", d["config"]["newton_iterations"], "krylov", d["config"].get("krylov_iterations"), "frac %.3f" % r["frac"], r.get("mean_launch_us", r.get("mean_solve_us")))
    except Exception as e:
        print(f, "FAILED", e)
PY
head -6 gpurun_out/r3f_bench_kernel_stats.csv | cut -c1-120
