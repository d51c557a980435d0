"""Probe of the V = -12.5 staged 1D run: where Newton stops converging (tools/stern_schedule.py reported step 5999)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gmpnp_amd.edl1d import EDLRun
run = EDLRun(voltage_multiplier=-12.5, dry_run=False)
last = None
try:
    for n in range(1, 20001):
        prev_u = run.history[-1].copy()
        try:
            st = run.step(verbose=False)
        except Exception as e:
            print("step", n, "failed:", e)
            s = run.sys.last_stats
            np.savez("gpurun_out/v125_fail.npz", un=prev_u, step=n)
            # the failed solve's statistics
            from gmpnp_amd import backend
            run.sys.dev.set_state(prev_u.ravel(), prev_u.ravel())
            st2 = run.sys.dev.newton_solve(backend.newton_options(run.solver_parameters, dim=1), error_on_nonconvergence=False)
            print("retry from u = u_n: its", st2["iterations"], "residuals", ["%.3e" % r for r in st2["residuals"]])
            break
        run.history = run.history[-1:]
        if n % 500 == 0 or n > 5990:
            print(n, "its", st["iterations"], "res", ["%.3e" % r for r in st["residuals"]], flush=True)
finally:
    run.sys.close()
