"""How much of one MI355X K independent L_50_R_5 runs use when they share it from K host threads of ONE process (BASELINE
configs[4]: 35 independent jobs; `python -m gmpnp_amd.sweep --jobs_per_gpu K` runs them this way).  Every run is the bench
window (50 steps from t = 0 after 2 warm-up steps); the figure is the aggregate Newton iterations / s between two barriers
around the timed region, set-up excluded.

    python tools/concurrent_runs_probe.py [out.json]
"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from gmpnp_amd.pore3d import PoreRun
from gmpnp_amd.problem import pore_dirichlet

STEPS, WARM = 50, 2


def reset(r):
    r.sys.set_bcs(*pore_dirichlet(r.pp, r.bnd))
    r.sys.initialise([1.0] * 8 + [0.0])
    r.history = r.history[:1]
    r.newton_its, r.n, r.t = [], 0, 0.0
    r.sys.krylov_iterations = 0


def measure(K, **device_kwargs):
    runs = [PoreRun(num_steps=STEPS, concentration_elec=0.5, L=50e-9, R=5e-9, device_kwargs=dict(device_kwargs)) for _ in range(K)]
    bar = threading.Barrier(K + 1)
    times = [0.0] * K

    def work(k):
        r = runs[k]
        for _ in range(WARM):
            r.step(verbose=False)
        reset(r)
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            r.step(verbose=False)
        times[k] = time.perf_counter() - t0
        bar.wait()

    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    wall = time.perf_counter() - t0
    for t in th:
        t.join()
    its = sum(int(sum(r.newton_its)) for r in runs)
    same = all(r.newton_its == runs[0].newton_its for r in runs)
    for r in runs:
        r.sys.close()
    return {"runs": K, "options": device_kwargs, "newton_iterations": its, "wall_seconds": wall, "its_per_s": its / wall,
            "identical_newton_counts": same, "slowest_run_seconds": max(times)}


rows = []
CASES = ((1, {}), (2, {}), (3, {}), (4, {}), (2, {"shared_device": 1}), (4, {"shared_device": 1}), (6, {"shared_device": 1}))
if os.environ.get("PROBE_ONE_STREAM"):   # one stream per handle (coarse rebuild and warm-start test in the main stream), two-launch form kept
    kw1 = {"coarse_refresh": 3, "warm_in_stream": 1}
    CASES = ((1, {}), (1, kw1), (2, kw1), (3, kw1), (4, kw1), (5, kw1))
elif os.environ.get("PROBE_CASES"):   # e.g. "3,4,5,6" (default options) — with GPU_MAX_HW_QUEUES set by the caller
    CASES = ((1, {}),) + tuple((int(k), {}) for k in os.environ["PROBE_CASES"].split(","))
for K, kw in CASES:
    try:
        row = measure(K, **kw)
    except Exception as e:  # noqa: BLE001
        row = {"runs": K, "options": kw, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    rows.append(row)
    print(json.dumps(row), flush=True)
base = rows[0].get("its_per_s")
for r in rows:
    if base and "its_per_s" in r:
        r["vs_one_run"] = r["its_per_s"] / base
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump({"source": "tools/concurrent_runs_probe.py on 1 x MI355X", "rows": rows}, fh, indent=1)
print(json.dumps([(r["runs"], r["options"], r.get("vs_one_run")) for r in rows]))
