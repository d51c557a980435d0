"""GPU probe of the in-library partitioned solve: `nparts` ranks inside one process (device copies instead of RCCL) and the
RCCL transport at world size 1, against the serial golden step of L_10_R_5 / L_50_R_5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import _pore, GOLDEN
from gmpnp_amd import backend, dist
MUMPS_09 = {"nonlinear_solver": "newton", "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50,
            "relative_tolerance": 1e-4, "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}}
case = sys.argv[1] if len(sys.argv) > 1 else "pore10"
L = 10e-9 if case == "pore10" else 50e-9
pp, mesh, prob, bnd = _pore(L, 5e-9)
g = np.load(os.path.join(GOLDEN, case + "_steps.npz"))
nv = mesh.num_vertices
un = np.tile(np.r_[np.ones(8), 0.0], nv)
opts = backend.newton_options(MUMPS_09)
with backend.DeviceSolver(prob) as dev:
    dev.set_state(np.zeros(nv * 9), un)
    t0 = time.perf_counter(); st = dev.newton_solve(opts); dt = time.perf_counter() - t0
    print("serial: newton %d krylov %s  %.1f ms" % (st["iterations"], st["krylov_per_iteration"], 1e3 * dt), flush=True)
for nparts, rank in ((1, None), (2, None), (4, None), (8, None), (1, 0)):
    try:
        t0 = time.perf_counter()
        ps = dist.PartitionedSolver(prob, nparts, rank=rank, use_torch_dist=False)
        t1 = time.perf_counter()
        ps.set_state(np.zeros(nv * 9), un)
        st = ps.newton_solve(opts)
        t2 = time.perf_counter()
        u = ps.get_state()
        ps.close()
        err = np.linalg.norm(u - g["states"][0]) / np.linalg.norm(g["states"][0])
        print("%s nparts %d: newton %d (golden %d) krylov %s  rel.err %.2e  create %.2f s solve %.1f ms" % (
            "RCCL world 1" if rank is not None else "in-process", nparts, st["iterations"], int(g["newton_its"][0]), st["krylov_per_iteration"], err, t1 - t0, 1e3 * (t2 - t1)), flush=True)
    except Exception as e:  # noqa: BLE001
        print("nparts", nparts, "rank", rank, "FAILED:", e, flush=True)
