"""Exploratory GPU check (not a test): compare HIP assembly / SpMV / solves with the oracle and time them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd import backend
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem
import gmpnp_oracle as O

L, R = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (50e-9, 5e-9)
pp = pore_parameters(concentration_elec=0.5, L=L, R=R)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, bnd = pore_problem(pp, mesh)
nv = mesh.num_vertices
rng = np.random.default_rng(0)
u = np.concatenate([rng.uniform(.5, 1.5, (nv, 8)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
un = np.concatenate([rng.uniform(.5, 1.5, (nv, 8)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
t = time.time(); dev = backend.DeviceSolver(prob, phase_timing=1); print("create %.3fs nagg=%d nblocks=%d" % (time.time() - t, dev.n_aggregates, dev.n_blocks))
dev.set_state(u, un)
t = time.time(); F, nrm = dev.assemble(True); print("assemble %.4fs norm %.10e" % (time.time() - t, nrm))
Fo, Ao = O.assemble(prob, u, un)
print("oracle norm %.10e  F relerr %.3e" % (np.linalg.norm(Fo), np.linalg.norm(F - Fo) / np.linalg.norm(Fo)))
A = dev.jacobian_csr()
d = (A - Ao); print("J relerr (fro) %.3e  nnz %d vs %d" % (np.sqrt((d.data ** 2).sum()) / np.sqrt((Ao.data ** 2).sum()), A.nnz, Ao.nnz))
x = rng.standard_normal(prob.ndof)
y = dev.spmv(x); yo = Ao @ x
print("spmv relerr %.3e" % (np.linalg.norm(y - yo) / np.linalg.norm(yo)))
for k, name in ((0, "spmv"), (1, "element F+J"), (2, "jac gather"), (3, "res gather")):
    print("kernel %-12s %.2f us" % (name, dev.time_kernel(k, 50)))
for mode, name in ((backend.LINEAR_TWOLEVEL, "twolevel"), (backend.LINEAR_JACOBI, "jacobi")):
    t = time.time()
    try:
        xs, st = dev.linear_solve(Fo, mode, 1e-10, 0.0, 5000)
        print(name, st, "true relres %.3e  %.4fs" % (np.linalg.norm(Ao @ xs - Fo) / np.linalg.norm(Fo), time.time() - t))
    except Exception as e:
        print(name, "FAILED", e)
# Newton from zero
u0 = np.zeros(prob.ndof); un1 = np.tile(np.r_[np.ones(8), 0.0], nv)
opts = backend.newton_options({"newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50,
    "relative_tolerance": 1e-4, "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}})
for rep in range(3):
    dev.set_state(u0, un1)
    t = time.time(); st = dev.newton_solve(opts); dt = time.time() - t
    print("newton its %d krylov %s  %.4fs  -> %.1f its/s ; ms asm %.2f setup %.2f krylov %.2f" % (st["iterations"], st["krylov_per_iteration"], dt, st["iterations"] / dt, st["ms_assemble"], st["ms_setup"], st["ms_krylov"]))
print("residuals", st["residuals"])
np.save(os.path.join(ROOT, "gpurun_out", "u_step1_gpu.npy"), dev.get_state())
for rep in range(3):
    t = time.time(); xs, st = dev.linear_solve(Fo, backend.LINEAR_TWOLEVEL, 1e-10, 0.0, 5000); print("twolevel again", st["iterations"], "%.3f ms" % (1e3 * (time.time() - t)))
