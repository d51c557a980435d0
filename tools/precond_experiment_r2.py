"""Round-2 preconditioner experiments on the CPU (NumPy/SciPy on the oracle's Jacobian of L_50_R_5): what would cut the
BiCGStab iteration count of the two-level solver?  Right-preconditioned BiCGStab to 1e-10, first Newton system of time step 0
(`zero`) or of time step 1 (`step1`).  Results of the `zero` run (iterations; the library's scheme = first line):

    8 slabs, piecewise constant (72 coarse dofs)                      65
    8 slabs x (1, x, y)            (216)                              55
    8 slabs x (1, x, y, r^2)       (288)                              56
    8 slabs x (1, x, y, r^2, xy, x^2-y^2) (432)                       52
    16 slabs x (1, x, y)           (432)                              42        32 slabs x (1, x, y) (864): 45
    ILU (SuperLU, fill 1) + 8 slabs                                   47        ILU alone: > 3000
    2-sweep Jacobi (Neumann-1, 2 SpMV per application) + 8 slabs      45  (= 90 SpMV-equivalents)
    node-block Gauss-Seidel + 8 slabs                                 61        alone: 639
    compact 7 / 14 / 28-node cluster blocks + 8 slabs                 57 / 60 / 49
    GMRES(30 / 60 / 200) matvecs with the 8-slab preconditioner       112 / 115 / 101   (BiCGStab: 2 x 65 = 130)
    GMRES(200) with 16 slabs x (1, x, y)                              73

No cheap change halves the count: a stronger smoother buys 10-30 %, a 6x larger coarse space 35 % (and needs an inverse that
no longer fits LDS), GMRES 22 % fewer matvecs for a growing orthogonalisation cost.  The solver was left as it is (DESIGN.md
section 4)."""
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import gmpnp_oracle as O
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem
from gmpnp_amd.backend import slab_permutation
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, _ = pore_problem(pp, mesh)
nv = mesh.num_vertices; NF = 9
g = np.load('/root/repo/tests/golden/pore50_steps.npz')
state = sys.argv[1] if len(sys.argv) > 1 else "zero"
if state == "zero":
    u = np.zeros(prob.ndof); un = np.tile(np.r_[np.ones(8), 0.0], nv)
else:
    un = g["states"][0]; u = g["states"][0].copy()   # first Newton iteration of step 1
b, A = O.assemble(prob, u, un, want_jacobian=True)
A = A.tocsr()
perm = np.asarray(slab_permutation(prob.coords, prob.cells, window=0))
dperm = (perm[:, None] * NF + np.arange(NF)[None, :]).ravel()
A = A[dperm][:, dperm].tocsr(); b = b[dperm]
coords = prob.coords[perm]
n = A.shape[0]

def bicgstab(A, b, Minv, rtol=1e-10, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); rh = r.copy(); rho = alpha = om = 1.0; v = p = np.zeros_like(b); bn = np.linalg.norm(b)
    for k in range(maxit):
        rho_new = rh @ r
        beta = (rho_new / rho) * (alpha / om) if k else 0.0
        p = r + beta * (p - om * v) if k else r.copy()
        ph = Minv(p); v = A @ ph; alpha = rho_new / (rh @ v); s = r - alpha * v
        sh = Minv(s); t = A @ sh; om = (t @ s) / (t @ t)
        x += alpha * ph + om * sh; r = s - om * t; rho = rho_new
        if np.linalg.norm(r) <= rtol * bn: return x, k + 1
    return x, maxit

# node-block Jacobi as sparse block-diagonal matrix
blocks = [np.linalg.inv(A[i*NF:(i+1)*NF, i*NF:(i+1)*NF].toarray()) for i in range(nv)]
Dinv = sp.block_diag(blocks, format="csr")
Bj = lambda x: Dinv @ x

def two_level_general(A, Binv_mat, Pm):
    """additive in the product form used by the library: M^-1 = Binv (I + P Aci P^T), Aci = (P^T A Binv P)^-1"""
    AsP = (A @ (Binv_mat @ Pm)).toarray() if sp.issparse(Pm) else A @ (Binv_mat @ Pm)
    Ac = (Pm.T @ AsP)
    Ac = Ac if isinstance(Ac, np.ndarray) else np.asarray(Ac)
    Aci = np.linalg.inv(Ac)
    return lambda x: Binv_mat @ (x + Pm @ (Aci @ (Pm.T @ x)))

def P_slabs(nagg, funcs):
    """funcs: list of node functions (nv,) ; coarse dof = (agg, func, field)"""
    agg = (np.arange(nv) * nagg) // nv
    nfun = len(funcs)
    rows, cols, vals = [], [], []
    for k, f in enumerate(funcs):
        for fld in range(NF):
            rows.append(np.arange(nv) * NF + fld); cols.append((agg * nfun + k) * NF + fld); vals.append(f)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, nagg * nfun * NF))

axis = np.argmax(coords.max(0) - coords.min(0)); oth = [i for i in range(3) if i != axis]
xx = coords[:, oth[0]] - coords[:, oth[0]].mean(); yy = coords[:, oth[1]] - coords[:, oth[1]].mean(); zz = coords[:, axis]
R0 = np.sqrt((xx**2+yy**2).max())
one = np.ones(nv)
print("state", state, "n", n, "|b|", np.linalg.norm(b))
for nagg in (8,):
    for name, funcs in (("const", [one]), ("const,x,y", [one, xx/R0, yy/R0]), ("const,x,y,r2", [one, xx/R0, yy/R0, (xx**2+yy**2)/R0**2]),
                        ("const,x,y,r2,xy,x2-y2", [one, xx/R0, yy/R0, (xx**2+yy**2)/R0**2, xx*yy/R0**2, (xx**2-yy**2)/R0**2])):
        Pm = P_slabs(nagg, funcs)
        M = two_level_general(A, Dinv, Pm); x, it = bicgstab(A, b, M)
        print("slabs %d, funcs %-22s coarse %4d: its %d" % (nagg, name, Pm.shape[1], it), flush=True)
for nagg in (16, 32):
    Pm = P_slabs(nagg, [one, xx/R0, yy/R0]); M = two_level_general(A, Dinv, Pm); x, it = bicgstab(A, b, M)
    print("slabs %d, const,x,y coarse %d: its %d" % (nagg, Pm.shape[1], it), flush=True)
# stronger smoothers with the const coarse space
Pm = P_slabs(8, [one])
# (a) ILU(0) of A (point), via spilu with fill_factor=1, drop_tol=0 is not exact ILU0 but close
t0=time.time(); ilu = spla.spilu(A.tocsc(), drop_tol=0.0, fill_factor=1.0, permc_spec="NATURAL", diag_pivot_thresh=0.0)
class Op:
    def __init__(s, f): s.f=f
    def __matmul__(s, x): 
        return s.f(x) if x.ndim==1 else np.column_stack([s.f(x[:, j]) for j in range(x.shape[1])])
Ilu = Op(ilu.solve)
x, it = bicgstab(A, b, ilu.solve); print("ILU(fill 1) only: its", it, "nnz", ilu.L.nnz+ilu.U.nnz, flush=True)
def two_level_op(A, Bop, Pm):
    Pd = Pm.toarray()
    AsP = A @ (Bop @ Pd); Aci = np.linalg.inv(Pd.T @ AsP)
    return lambda x: Bop @ (x + Pd @ (Aci @ (Pd.T @ x)))
M = two_level_op(A, Ilu, Pm); x, it = bicgstab(A, b, M); print("ILU(fill 1) + 8 slabs: its", it, flush=True)
# (b) two sweeps damped Jacobi: B2 = (2I - Dinv A) Dinv  (Neumann degree 1)
B2 = Op(lambda x: (lambda y: 2*y - Dinv @ (A @ y))(Dinv @ x))
M = two_level_op(A, B2, Pm); x, it = bicgstab(A, b, M); print("Neumann-1 Jacobi (2 SpMV per apply) + 8 slabs: its", it, flush=True)
# (c) block Gauss-Seidel (forward) in slab order: (D+L)^-1
Lb = sp.tril(A, format="csr")  # point lower incl diag ~ GS
# block lower: use node-block structure
rows_node = np.repeat(np.arange(n)//NF, np.diff(A.indptr)); cols_node = A.indices//NF
mask = cols_node <= rows_node
BL = sp.csr_matrix((A.data[mask], (np.repeat(np.arange(n), np.diff(A.indptr))[mask], A.indices[mask])), shape=A.shape).tocsc()
lu_bl = spla.splu(BL, permc_spec="NATURAL", diag_pivot_thresh=0.0)
GS = Op(lu_bl.solve)
x, it = bicgstab(A, b, lu_bl.solve); print("block GS only: its", it, flush=True)
M = two_level_op(A, GS, Pm); x, it = bicgstab(A, b, M); print("block GS + 8 slabs: its", it, flush=True)
print("---- GMRES matvec counts (right preconditioned), const coarse 8 slabs")
Pm = P_slabs(8, [one]); M = two_level_general(A, Dinv, Pm)
for restart in (30, 60, 200):
    cnt = [0]
    def mv(y): cnt[0] += 1; return A @ M(y)
    Aop = spla.LinearOperator(A.shape, matvec=mv)
    y, info = spla.gmres(Aop, b, rtol=1e-10, atol=0.0, restart=restart, maxiter=20)
    x = M(y); print("GMRES(%d): matvecs %d info %d true rel res %.2e" % (restart, cnt[0], info, np.linalg.norm(A@x-b)/np.linalg.norm(b)), flush=True)
Pm = P_slabs(16, [one, xx/R0, yy/R0]); M = two_level_general(A, Dinv, Pm)
cnt=[0]
Aop = spla.LinearOperator(A.shape, matvec=mv)
y, info = spla.gmres(Aop, b, rtol=1e-10, atol=0.0, restart=200, maxiter=20); print("GMRES(200) with 16 slabs x (1,x,y): matvecs", cnt[0])
