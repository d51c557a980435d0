"""Where a time step's wall clock goes outside gmpnp_newton_solve (GPU box): the Python glue of PoreRun.step, piece by piece."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd.pore3d import PoreRun, column_medians
from gmpnp_amd.problem import pore_dirichlet

run = PoreRun(num_steps=60, concentration_elec=0.5, L=50e-9, R=5e-9)
for _ in range(2):
    run.step(verbose=False)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
T0 = time.perf_counter()
for _ in range(50):
    t = time.perf_counter()
    st = run.sys.solve(run.solver_parameters); t = tick("solve (python wrapper + library)", t)
    acc["inside library"] = acc.get("inside library", 0.0) + st["ms_total"] * 1e-3 if "ms_total" in st else 0.0
    vals = run.sys.vertex_values(); t = tick("vertex_values (D2H)", t)
    co2 = run.pp.sechenov_co2_scaled(*column_medians(vals, (1, 2, 3, 7))); t = tick("medians + Sechenov", t)
    d = pore_dirichlet(run.pp, run.bnd, co2); t = tick("pore_dirichlet", t)
    run.sys.set_bcs(*d); t = tick("set_bcs (H2D)", t)
    run.history.append(vals); run.CO2_min = float(np.amin(vals[:, 4])); t = tick("history + min", t)
    run.sys.assign_previous(); t = tick("assign_previous", t)
wall = time.perf_counter() - T0
print("50 steps: %.1f ms per step" % (wall / 50 * 1e3))
for k, v in acc.items():
    print("  %-36s %8.1f us per step" % (k, v / 50 * 1e6))
run.sys.close()
