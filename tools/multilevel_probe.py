"""Geometric multilevel term of the preconditioner on the GPU (gmpnp_attach_coarse_level): BiCGStab iterations per linear solve and
time per Newton iteration with and without it on uniformly refined L_50_R_5 meshes; Newton counts must be identical and the
states must agree to solver accuracy.

    python tools/multilevel_probe.py [refine=1] [steps=2] [theta=1.0] [out.json] [sweeps=2]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd.pore3d import PoreRun

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
theta = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
out = {"refine": R, "steps": steps, "theta": theta, "sweeps": sweeps, "runs": {}}
if os.environ.get("ML_ONLY"):   # skip the two-level reference run (scans)
    pass
states = {}
cases = (("two-level", {}), ("multilevel", {"multilevel": True, "ml_theta": theta, "ml_sweeps": sweeps}))
if os.environ.get("ML_ONLY"):
    cases = cases[1:]
for name, kw in cases:
    run = PoreRun(num_steps=steps, concentration_elec=0.5, L=50e-9, R=5e-9, refine=R, **kw)
    try:
        t0 = time.perf_counter()
        per = []
        for _ in range(steps):
            st = run.step(verbose=False)
            per.append({"newton": st["iterations"], "krylov": list(st["krylov_per_iteration"][: st["iterations"]]), "direct_solves": st["direct_solves"]})
        wall = time.perf_counter() - t0
        its = int(sum(run.newton_its)); kry = int(run.sys.krylov_iterations)
        out["runs"][name] = {"n_vertices": run.mesh.num_vertices, "newton_iterations": its, "krylov_iterations": kry, "krylov_per_solve": kry / max(its, 1),
                             "seconds": wall, "ms_per_newton_iteration": 1e3 * wall / max(its, 1), "us_per_krylov_iteration": 1e6 * wall / max(kry, 1), "steps": per}
        states[name] = np.array(run.history[1:])
        print(name, json.dumps({k: v for k, v in out["runs"][name].items() if k != "steps"}), flush=True)
        print("   krylov per solve:", [p["krylov"] for p in per], flush=True)
    finally:
        run.sys.close()
if "two-level" in states:
    a, b = states["two-level"], states["multilevel"]
    out["state_relative_difference"] = float(np.linalg.norm(a - b) / np.linalg.norm(a))
    out["newton_counts_equal"] = [p["newton"] for p in out["runs"]["two-level"]["steps"]] == [p["newton"] for p in out["runs"]["multilevel"]["steps"]]
    print("state rel diff %.3e, Newton counts equal: %s" % (out["state_relative_difference"], out["newton_counts_equal"]))
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as fh:
        json.dump(out, fh, indent=1)
