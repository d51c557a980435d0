# round 3, first measurement run on the GPU box: full -m gpu suite, then the bench line (with the new edl50 / banded legs)
set -x
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests.log 2>&1; echo rc=$?; tail -5 gpurun_out/r3_gputests.log
timeout -k 10 400 python bench.py --steps 50 --warmup 2 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo rc=$?; tail -c 600 gpurun_out/r3_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["mean_launch_us"])
print("edl50", {k: d.get("edl50", {}).get(k) for k in ("value", "error", "newton_iterations_equal_cpu")}, d.get("edl50", {}).get("roofline", {}).get("mean_solve_us"), d.get("edl50", {}).get("cpu_baseline", {}).get("value"))
cb = d.get("cpu_baseline", {})
print("cpu one", cb.get("one_thread", {}).get("value"), "banded", cb.get("banded_all_cores"))
PY
