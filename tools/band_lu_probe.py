"""Block-banded LU (the 3D direct solver / Krylov fallback): accuracy against SciPy's sparse LU on the exported
Jacobian, time per factorisation, Newton with the LU instead of BiCGStab, and the L_50_R_1 run that BiCGStab loses.

  python tools/band_lu_probe.py [accuracy] [newton] [r1]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd import backend as B  # noqa: E402
from gmpnp_amd.pore3d import SOLVER_PARAMETERS, PoreRun  # noqa: E402

what = sys.argv[1:] or ["accuracy", "newton", "r1"]
BAND = {"nonlinear_solver": "newton", "newton_solver": dict(SOLVER_PARAMETERS["newton_solver"], linear_solver="band_lu")}


def relerr(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


if "accuracy" in what:
    for L, R in ((10e-9, 5e-9), (50e-9, 5e-9), (50e-9, 1e-9)):
        run = PoreRun(num_steps=2, concentration_elec=0.5, L=L, R=R)
        run.step(verbose=False)
        dev = run.sys.dev
        F, _ = dev.assemble(True)
        A = dev.jacobian_csr()
        xo = spla.splu(A.tocsc()).solve(F)
        t0 = time.perf_counter()
        x, st = dev.linear_solve(F, B.LINEAR_BAND_LU, 1e-10, 0.0, 10)
        t1 = time.perf_counter()
        x2, st2 = dev.linear_solve(F, B.LINEAR_BAND_LU, 1e-10, 0.0, 10)
        t2 = time.perf_counter()
        xk, stk = dev.linear_solve(F, B.LINEAR_TWOLEVEL, 1e-10, 0.0, 10000)
        print("L %g R %g  nv %d  band LU: |Ax-b|/|b| %.2e  vs splu %.2e  (BiCGStab vs splu %.2e, %d its)  first %.1f ms, second %.1f ms  bitwise repeat %s"
              % (L, R, run.mesh.num_vertices, relerr(A @ x, F), relerr(x, xo), relerr(xk, xo), stk["iterations"],
                 (t1 - t0) * 1e3, (t2 - t1) * 1e3, np.array_equal(x, x2)), flush=True)
        run.sys.close()

if "newton" in what:
    for sp, name in ((SOLVER_PARAMETERS, "two-level BiCGStab"), (BAND, "band LU")):
        run = PoreRun(num_steps=3, concentration_elec=0.5, L=10e-9, R=5e-9, solver_parameters=sp)
        t0 = time.perf_counter()
        its = []
        for n in range(3):
            st = run.step(verbose=False)
            its.append((st["iterations"], st["direct_solves"]))
        print(name, its, "%.2f s" % (time.perf_counter() - t0), "CO2_min", run.CO2_min, flush=True)
        states = run.sys.vertex_values().copy()
        run.sys.close()
        if name == "band LU":
            print("  states vs BiCGStab run: %.2e" % relerr(states, ref_states))
        ref_states = states

if "r1" in what:
    run = PoreRun(num_steps=8, concentration_elec=0.5, L=50e-9, R=1e-9)
    try:
        for n in range(8):
            t0 = time.perf_counter()
            st = run.step(verbose=False)
            print("R_1 step", n, "newton", st["iterations"], "direct", st["direct_solves"], "krylov", st["krylov_per_iteration"],
                  "%.2f s" % (time.perf_counter() - t0), "CO2_min %.6g" % run.CO2_min, flush=True)
    except RuntimeError as e:
        print("R_1 FAILED at step", run.n, str(e)[:300], flush=True)
    run.sys.close()
