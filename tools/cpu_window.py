"""CPU leg of the headline comparison over the WHOLE bench window: the first K time steps of
BASELINE configs[2] (L_50_R_5, 0.5 M, K+, V = -1) with the CPU oracle — NumPy P1 assembly of residual + exact Jacobian,
SciPy SuperLU per Newton iteration (the algorithm class of the reference's MUMPS solve, 3D:792), damped update, the
reference's stopping rule and per-step Sechenov feedback — timed on this box's host cores.

    python tools/cpu_window.py --steps 50 --threads 1 --out profiles/r02/cpu_window_1thread.json

Prints / stores Newton iterations per second over the solve phase, split into assembly and LU time, and the Newton
iteration count per step (the GPU run of the same window must give the same counts).  FEniCS itself is not installable.
"""
import argparse, json, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--threads", type=int, default=1, help="BLAS/OpenMP threads (0 = all cores)")
ap.add_argument("--out", type=str, default=None)
ap.add_argument("--first", type=int, default=0, help="first time step of this call (a gpurun call is limited to 20 min: the window is timed in two halves)")
ap.add_argument("--load", type=str, default=None, help="checkpoint (npz: u, un, co2, accumulated timings) of the previous half")
ap.add_argument("--save", type=str, default=None, help="checkpoint written after EVERY step")
ap.add_argument("--time-budget", type=float, default=0.0, help="stop cleanly after this many seconds of THIS call (0 = run to --steps)")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if a.threads > 0:
    nthreads = a.threads
else:
    import bench   # usable_cpus(): the cgroup quota, not os.cpu_count() (16 vs 256 on the one-GPU box)
    nthreads = bench.usable_cpus()
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = str(nthreads)
import numpy as np
import gmpnp_oracle as O
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem, pore_dirichlet

pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, bnd = pore_problem(pp, mesh)
nv, nf = mesh.num_vertices, prob.nf
u = np.zeros(prob.ndof)
un = np.tile(np.r_[np.ones(nf - 1), 0.0], nv)
O.assemble(prob, u, un)   # scatter pattern, once (DOLFIN builds its sparsity pattern once, too)
its, t_asm, t_lu, wall_before = [], 0.0, 0.0, 0.0
if a.load:
    ck = np.load(a.load)
    u, un = ck["u"], ck["un"]
    prob.bc_dofs, prob.bc_vals = pore_dirichlet(pp, bnd, float(ck["co2"]))
    its, t_asm, t_lu, wall_before = [int(v) for v in ck["its"]], float(ck["t_asm"]), float(ck["t_lu"]), float(ck["wall"])
    assert len(its) == a.first
t0 = time.perf_counter()
for n in range(a.first, a.steps):
    u, st = O.newton_solve(prob, u, un, maximum_iterations=50, relative_tolerance=1e-4, absolute_tolerance=1e-4, relaxation_parameter=0.9)
    u2 = u.reshape(nv, nf)
    co2 = pp.sechenov_co2_scaled(np.median(u2[:, 1]), np.median(u2[:, 2]), np.median(u2[:, 3]), np.median(u2[:, 7]))
    prob.bc_dofs, prob.bc_vals = pore_dirichlet(pp, bnd, co2)
    un = u.copy()
    its.append(st.iterations); t_asm += st.t_assemble; t_lu += st.t_linear
    wall = wall_before + time.perf_counter() - t0
    print("step %d: %d Newton iterations, %.1f s so far" % (n, st.iterations, wall), flush=True)
    if a.save:
        np.savez(a.save, u=u, un=un, co2=co2, its=np.array(its), t_asm=t_asm, t_lu=t_lu, wall=wall)
    if a.time_budget > 0 and time.perf_counter() - t0 > a.time_budget:
        break
wall = wall_before + time.perf_counter() - t0
out = {"workload": "3D MPNP_CO2ER_pore L_50_R_5, 0.5 M KHCO3, K+, V=-1: time steps 0..%d from t=0" % (a.steps - 1),
       "kind": "port (CPU oracle: NumPy assembly + SciPy SuperLU; FEniCS/MUMPS not installable)",
       "note": "solve phase only (mesh ingest and the one-off scatter pattern excluded); timed in calls of <= 20 min, state carried over",
       "threads": nthreads, "host_cpus": os.cpu_count(), "steps": len(its), "newton_iterations": int(sum(its)),
       "newton_per_step": its, "seconds": wall, "assembly_seconds": t_asm, "lu_seconds": t_lu,
       "value": sum(its) / wall, "unit": "Newton-iterations/s"}
print(json.dumps(out))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
