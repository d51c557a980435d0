# bench line (event-timed bursts) next to rocprofv3's kernel durations of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/bvr_bench.json 2> gpurun_out/bvr_bench.err &&
python bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/bvr_bench2.json 2>> gpurun_out/bvr_bench.err &&
rm -rf gpurun_out/bvr_stats && rocprofv3 --kernel-trace --stats -d gpurun_out/bvr_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > gpurun_out/bvr_under_rocprof.json 2> gpurun_out/bvr_rocprof.err &&
cp $(find gpurun_out/bvr_stats -name "*kernel_stats.csv") gpurun_out/bvr_kernel_stats.csv; rm -rf gpurun_out/bvr_stats
python - <<'PY'
import json, csv
for f in ("bvr_bench", "bvr_bench2", "bvr_under_rocprof"):
    d = json.load(open("gpurun_out/%s.json" % f)); r = d["roofline"]
    print(f, "its/s %.1f" % d["value"], "mean half-iteration %.2f us" % r["mean_launch_us"], "frac %.3f" % r["frac"], "sampled", r["launches_sampled"], "of", r["launches_total"])
rows = list(csv.DictReader(open("gpurun_out/bvr_kernel_stats.csv")))[:2]
print("rocprof:", [(x["Name"][13:21], round(float(x["AverageNs"]) / 1e3, 2)) for x in rows])
PY
