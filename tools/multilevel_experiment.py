"""CPU experiment (NumPy/SciPy on the oracle's Jacobians): does a geometric multilevel term make the BiCGStab iteration count of
the two-level solver independent of the refinement level?  L_50_R_5 refined `R` times by gmpnp_amd.mesh.refine_pore (nested P1
spaces: fine vertices = coarse vertices + edge midpoints), first Newton system of time step 0 (`zero`).  Variants:

  base        the library's scheme: M^-1 = Dinv (I + P Aci P^T), node-block Jacobi + 8 piecewise-constant slabs
  add-redisc  base + theta * sum_l P_l Dinv_l P_l^T, Dinv_l from the Jacobian REDISCRETISED on level l at the injected state
  add-galerk  the same with Dinv_l = inverse diagonal blocks of the Galerkin operator P_l^T J P_l
  vcycle      multiplicative V(1,1) cycle with damped block Jacobi, Galerkin operators, slab-corrected Jacobi on the coarsest level

    python tools/multilevel_experiment.py [R=1] [zero|step1]
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import gmpnp_oracle as O
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path, refine_pore, mark_pore_boundaries, pore_wall_tolerance
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import Problem, pore_dirichlet
from gmpnp_amd.backend import slab_permutation

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NF = 9
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
bnd = mark_pore_boundaries(mesh, pp.aspect_pore, pore_wall_tolerance(pp.L, pp.R))
levels = [(mesh, bnd)]
for _ in range(R):
    levels.append(refine_pore(*levels[-1]))
levels = levels[::-1]            # levels[0] = finest
probs = []
for m, b in levels:
    dofs, vals = pore_dirichlet(pp, b)
    probs.append(Problem(coords=m.coords, cells=m.cells, model=pp.model, wall_facets=b.ds_facets[2], exit_facets=b.ds_facets[3], bc_dofs=dofs, bc_vals=vals))
print("levels (vertices):", [p.coords.shape[0] for p in probs], flush=True)

def prolongation(nc, fine_mesh_coarse):   # fine <- coarse for one refinement of `fine_mesh_coarse` (the coarse mesh)
    from gmpnp_amd.mesh import refine_uniform
    c = fine_mesh_coarse.cells
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
    e = np.sort(c[:, pairs].reshape(-1, 2), axis=1).astype(np.int64)
    ukey = np.unique(e[:, 0] * nc + e[:, 1])
    edges = np.stack([ukey // nc, ukey % nc], 1)
    ne = len(edges)
    rows = np.concatenate([np.arange(nc), nc + np.arange(ne), nc + np.arange(ne)])
    cols = np.concatenate([np.arange(nc), edges[:, 0], edges[:, 1]])
    vals = np.concatenate([np.ones(nc), 0.5 * np.ones(2 * ne)])
    Pn = sp.csr_matrix((vals, (rows, cols)), shape=(nc + ne, nc))
    return sp.kron(Pn, sp.identity(NF), format="csr")

Ps = [prolongation(probs[l + 1].coords.shape[0], levels[l + 1][0]) for l in range(R)]   # Ps[l]: level l <- level l+1
state = sys.argv[2] if len(sys.argv) > 2 else "zero"
nv0 = probs[0].coords.shape[0]
un = np.tile(np.r_[np.ones(8), 0.0], nv0)
u = np.zeros(nv0 * NF)
if state != "zero":   # a smooth non-trivial state: one damped Newton step from zero on the COARSE mesh, prolonged
    pc = probs[-1]; nvc = pc.coords.shape[0]
    uc, _ = O.newton_solve(pc, np.zeros(nvc * NF), np.tile(np.r_[np.ones(8), 0.0], nvc), relaxation_parameter=0.9, maximum_iterations=4, error_on_nonconvergence=False)
    u = uc
    for l in range(R - 1, -1, -1):
        u = Ps[l] @ u
t0 = time.time()
b, A = O.assemble(probs[0], u, un)
A = A.tocsr()
print("fine assembly %.1fs, n = %d, nnz = %d" % (time.time() - t0, A.shape[0], A.nnz), flush=True)
bc = [np.zeros(p.ndof, dtype=bool) for p in probs]
for k, p in enumerate(probs):
    bc[k][p.bc_dofs] = True

def block_dinv(M, nv):
    M = M.tobsr(blocksize=(NF, NF))
    M.sort_indices()
    rows = np.repeat(np.arange(nv), np.diff(M.indptr))
    diag = M.data[M.indices == rows]
    assert len(diag) == nv
    return sp.bsr_matrix((np.linalg.inv(diag), np.arange(nv), np.arange(nv + 1)), shape=M.shape).tocsr()

Dinv = block_dinv(A, nv0)
# slab coarse space in slab order of the fine mesh
perm = np.asarray(slab_permutation(probs[0].coords, probs[0].cells, window=0))
pos = np.empty(nv0, dtype=np.int64); pos[perm] = np.arange(nv0)
agg = (pos * 8) // nv0
Pslab = sp.csr_matrix((np.ones(nv0 * NF), (np.arange(nv0 * NF), np.repeat(agg, NF) * NF + np.tile(np.arange(NF), nv0))), shape=(nv0 * NF, 8 * NF))
Aci = np.linalg.inv((Pslab.T @ (A @ (Dinv @ Pslab))).toarray())
base = lambda x: Dinv @ (x + Pslab @ (Aci @ (Pslab.T @ x)))

def bicgstab(A, b, Minv, rtol=1e-10, maxit=1500):
    x = np.zeros_like(b); r = b.copy(); rh = r.copy(); rho = alpha = om = 1.0; v = p = np.zeros_like(b); bn = np.linalg.norm(b)
    for k in range(maxit):
        rho_new = rh @ r
        beta = (rho_new / rho) * (alpha / om) if k else 0.0
        p = r + beta * (p - om * v) if k else r.copy()
        ph = Minv(p); v = A @ ph; alpha = rho_new / (rh @ v); s = r - alpha * v
        sh = Minv(s); t = A @ sh; om = (t @ s) / (t @ t)
        x += alpha * ph + om * sh; r = s - om * t; rho = rho_new
        if not np.isfinite(rho): return x, -1
        if np.linalg.norm(r) <= rtol * bn: return x, k + 1
    return x, maxit

t0 = time.time(); x, it = bicgstab(A, b, base); print("base: %d iterations (%.0fs)" % (it, time.time() - t0), flush=True)

# cumulative prolongations level l -> finest, masked at Dirichlet dofs on both sides
cum = [None] * (R + 1)
acc = sp.identity(nv0 * NF, format="csr")
for l in range(1, R + 1):
    acc = (acc @ Ps[l - 1]).tocsr()
    cum[l] = acc
def masked(Pm, l):
    keep_f = sp.diags((~bc[0]).astype(float)); keep_c = sp.diags((~bc[l]).astype(float))
    return (keep_f @ Pm @ keep_c).tocsr()
# rediscretised coarse Jacobians at the injected state (coarse vertices are the first vertices of the finer mesh)
ul, unl = [u], [un]
for l in range(1, R + 1):
    nvl = probs[l].coords.shape[0]
    ul.append(ul[-1].reshape(-1, NF)[:nvl].ravel().copy()); unl.append(unl[-1].reshape(-1, NF)[:nvl].ravel().copy())
Are = [A] + [O.assemble(probs[l], ul[l], unl[l])[1].tocsr() for l in range(1, R + 1)]
FAST = os.environ.get("ML_FAST", "0") == "1"    # large meshes: rediscretised additive variant only
Aga = [A]
for l in range(1, (0 if FAST else R) + 1):
    Pm = masked(Ps[l - 1], l) if False else Ps[l - 1]
    G = (Ps[l - 1].T @ Aga[-1] @ Ps[l - 1]).tocsr()
    # Dirichlet rows of the coarse level: identity
    keep = sp.diags((~bc[l]).astype(float)); G = (keep @ G @ keep + sp.diags(bc[l].astype(float))).tocsr()
    Aga.append(G)
for name, ops in ((("add-redisc", Are),) if FAST else (("add-redisc", Are), ("add-galerk", Aga))):
    Dl = [None] + [block_dinv(ops[l], probs[l].coords.shape[0]) for l in range(1, R + 1)]
    Pl = [None] + [masked(cum[l], l) for l in range(1, R + 1)]
    for theta in ((1.0,) if FAST else (1.0, 0.5, 0.25)):
        def M(x, theta=theta):
            y = base(x)
            for l in range(1, R + 1):
                y = y + theta * (Pl[l] @ (Dl[l] @ (Pl[l].T @ x)))
            return y
        t0 = time.time(); x, it = bicgstab(A, b, M); print("%s theta %.2f: %d iterations (%.0fs)" % (name, theta, it, time.time() - t0), flush=True)

# additive two-grid with an EXACT solve of the rediscretised level-1 problem: the limit of any better level-1 treatment
if os.environ.get("ML_EXACT", "0") == "1":
    lu1 = spla.splu(Are[1].tocsc())
    P1m = masked(cum[1], 1)
    for theta in (1.0, 0.5):
        def M(x, theta=theta):
            return base(x) + theta * (P1m @ lu1.solve(P1m.T @ x))
        t0 = time.time(); x, it = bicgstab(A, b, M); print("add-exact-coarse theta %.2f: %d iterations (%.0fs)" % (theta, it, time.time() - t0), flush=True)
    # level 1 treated by k sweeps of its own two-level (Jacobi + slabs) Richardson iteration (a cheap stand-in for a V-cycle there)
    nv1 = probs[1].coords.shape[0]
    D1 = block_dinv(Are[1], nv1)
    perm1 = np.asarray(slab_permutation(probs[1].coords, probs[1].cells, window=0)); pos1 = np.empty(nv1, dtype=np.int64); pos1[perm1] = np.arange(nv1)
    agg1 = (pos1 * 8) // nv1
    Ps1 = sp.csr_matrix((np.ones(nv1 * NF), (np.arange(nv1 * NF), np.repeat(agg1, NF) * NF + np.tile(np.arange(NF), nv1))), shape=(nv1 * NF, 8 * NF))
    Aci1 = np.linalg.inv((Ps1.T @ (Are[1] @ (D1 @ Ps1))).toarray())
    pre1 = lambda r: D1 @ (r + Ps1 @ (Aci1 @ (Ps1.T @ r)))
    for sweeps in (2, 3, 4):
        def coarse(r, sweeps=sweeps):
            x = pre1(r)
            for _ in range(sweeps - 1):
                x = x + pre1(r - Are[1] @ x)
            return x
        M = lambda x: base(x) + (P1m @ coarse(P1m.T @ x))
        t0 = time.time(); x, it = bicgstab(A, b, M); print("add, level 1 by %d Richardson sweeps of (Jacobi + slabs): %d iterations (%.0fs)" % (sweeps, it, time.time() - t0), flush=True)

# multiplicative V(1,1), damped Jacobi (omega), Galerkin operators; coarsest: `nc_sweeps` of slab-corrected Jacobi (base-like)
def make_vcycle(omega=0.7, ncs=2):
    Dg = [block_dinv(Aga[l], probs[l].coords.shape[0]) for l in range(R + 1)]
    nvL = probs[R].coords.shape[0]
    permL = np.asarray(slab_permutation(probs[R].coords, probs[R].cells, window=0)); posL = np.empty(nvL, dtype=np.int64); posL[permL] = np.arange(nvL)
    aggL = (posL * 8) // nvL
    PsL = sp.csr_matrix((np.ones(nvL * NF), (np.arange(nvL * NF), np.repeat(aggL, NF) * NF + np.tile(np.arange(NF), nvL))), shape=(nvL * NF, 8 * NF))
    AciL = np.linalg.inv((PsL.T @ (Aga[R] @ (Dg[R] @ PsL))).toarray())
    coarse_pre = lambda r: Dg[R] @ (r + PsL @ (AciL @ (PsL.T @ r)))
    Pm = [masked(Ps[l], l + 1) if False else Ps[l] for l in range(R)]
    def cyc(l, r):
        if l == R:
            x = np.zeros_like(r)
            for _ in range(ncs):
                x = x + coarse_pre(r - Aga[R] @ x)
            return x
        x = omega * (Dg[l] @ r)
        rc = Pm[l].T @ (r - Aga[l] @ x); rc[bc[l + 1]] = 0.0
        x = x + Pm[l] @ cyc(l + 1, rc)
        x = x + omega * (Dg[l] @ (r - Aga[l] @ x))
        return x
    return lambda r: cyc(0, r)
for omega, ncs in (() if FAST else ((0.7, 2), (0.5, 4))):
    t0 = time.time(); x, it = bicgstab(A, b, make_vcycle(omega, ncs)); print("vcycle omega %.1f coarse sweeps %d: %d iterations (%.0fs)" % (omega, ncs, it, time.time() - t0), flush=True)
