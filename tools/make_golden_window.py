#!/usr/bin/env python3
"""Golden fixture of the BENCH WINDOW: the first 50 time steps of BASELINE configs[2] (3D pore, L_50_R_5, 0.5 M, K,
V = -1) run with the CPU oracle (direct sparse LU per Newton iteration, like the reference's MUMPS, 3D:792).

bench.py's metric is a COUNT of Newton iterations over this window, so the GPU's BiCGStab@1e-10 solves have to give
the oracle's count at every one of the 50 steps, not only at the two steps `pore50_steps.npz` holds.  Stored (small):
Newton iterations and residual histories per step, the Sechenov CO2 Dirichlet value per step, the full state after
steps 0, 9, 24, 49, per-step per-field l2 norms and the values at 32 probe vertices for every step.

    python tools/make_golden_window.py [steps=50]      (about 25 min on one core)

ORACLE output, not FEniCS output (FEniCS is not installable here).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gmpnp_oracle as O
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, bnd = pore_problem(pp, mesh)
t0 = time.time()
out = O.pore_time_loop(pp, prob, bnd, steps, verbose=True)
nv, nf = mesh.num_vertices, prob.nf
S = out["states"].reshape(steps, nv, nf)
probes = np.unique(np.linspace(0, nv - 1, 32).astype(np.int64))
m = max(len(r) for r in out["residuals"])
res = np.array([r + [np.nan] * (m - len(r)) for r in out["residuals"]])
keep = [k for k in (0, 9, 24, 49) if k < steps]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pore50_window.npz"),
                    newton_its=np.array(out["newton_its"]), residuals=res, co2_bc=np.array(out["co2_bc"]),
                    probes=probes, probe_values=S[:, probes, :], field_norms=np.sqrt((S ** 2).sum(axis=1)),
                    full_steps=np.array(keep), full_states=out["states"][keep], args=np.array([0.5, 50e-9, 5e-9]))
print("done in %.0f s: %d Newton iterations over %d steps" % (time.time() - t0, int(np.sum(out["newton_its"])), steps))
