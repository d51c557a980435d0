"""Overhead of the partitioned algorithm itself (extra launches, reductions, packing), measured with every rank in ONE process
on one GPU: the ranks' kernels are serialised on one stream, so (time per BiCGStab iteration) x 1 GPU is an upper bound of the
per-rank compute on P GPUs times P, and the difference to the single-GPU solver is what partitioning adds before any RCCL cost.
    python tools/partition_scale_probe.py <refine> <P> [<P> ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import _pore
from gmpnp_amd import backend, dist
from gmpnp_amd.problem import pore_problem
MUMPS_09 = {"nonlinear_solver": "newton", "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50,
            "relative_tolerance": 1e-4, "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}}
refine = int(sys.argv[1]) if len(sys.argv) > 1 else 0
parts = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
pp, mesh, prob, bnd = _pore(50e-9, 5e-9)
if refine:
    prob, _ = pore_problem(pp, mesh, refine=refine)
nv = prob.coords.shape[0]
un = np.tile(np.r_[np.ones(8), 0.0], nv)
opts = backend.newton_options(MUMPS_09)
with backend.DeviceSolver(prob) as dev:
    dev.set_state(np.zeros(nv * 9), un)
    t0 = time.perf_counter(); st = dev.newton_solve(opts); dt = time.perf_counter() - t0
    u_ref = dev.get_state()
    print("refine %d, %d vertices | single GPU: newton %d, krylov %d, %.1f ms -> %.1f us per BiCGStab iteration (all-in)" % (
        refine, nv, st["iterations"], st["krylov_iterations"], 1e3 * dt, 1e6 * dt / st["krylov_iterations"]), flush=True)
for P in parts:
    t0 = time.perf_counter()
    with dist.PartitionedSolver(prob, P) as ps:
        t1 = time.perf_counter()
        ps.set_state(np.zeros(nv * 9), un)
        t2 = time.perf_counter(); st = ps.newton_solve(opts); t3 = time.perf_counter()
        u = ps.get_state()
    print("  %d partitions in one process: newton %d, krylov %d, plan+create %.1f s, solve %.1f ms -> %.1f us per iteration (serialised over the ranks), rel. diff %.1e" % (
        P, st["iterations"], st["krylov_iterations"], t1 - t0, 1e3 * (t3 - t2), 1e6 * (t3 - t2) / st["krylov_iterations"], np.linalg.norm(u - u_ref) / np.linalg.norm(u_ref)), flush=True)
