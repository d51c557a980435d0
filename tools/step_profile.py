"""Where does a time step of the 3D driver spend host time? (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd.pore3d import PoreRun
from gmpnp_amd.problem import pore_dirichlet
from gmpnp_amd.solver import column_medians
run = PoreRun(num_steps=20, concentration_elec=0.5, L=50e-9, R=5e-9)
T = {}
def tic(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for n in range(20):
    t0 = time.perf_counter(); st = run.sys.solve(run.solver_parameters); tic("solve", t0)
    T["gpu_total_ms"] = T.get("gpu_total_ms", 0.0) + st["ms_total"]
    T["newton"] = T.get("newton", 0) + st["iterations"]
    t0 = time.perf_counter(); vals = run.sys.vertex_values(); tic("vertex_values", t0)
    t0 = time.perf_counter(); co2 = run.pp.sechenov_co2_scaled(*column_medians(vals, (1, 2, 3, 7))); tic("median+sechenov", t0)
    t0 = time.perf_counter(); d, v = pore_dirichlet(run.pp, run.bnd, co2); tic("pore_dirichlet", t0)
    t0 = time.perf_counter(); run.sys.set_bcs(d, v); tic("set_bcs", t0)
    t0 = time.perf_counter(); run.sys.assign_previous(); tic("assign_previous", t0)
for k, v in T.items():
    print("%-18s %s" % (k, ("%.2f ms" % (1e3 * v)) if k not in ("gpu_total_ms", "newton") else v))
