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
u = np.zeros(prob.ndof); un = np.tile(np.r_[np.ones(8), 0.0], nv)
b, A = O.assemble(prob, u, un, want_jacobian=True)
A, b = O.apply_identity_rows(prob, A, b, u) if hasattr(O, 'apply_identity_rows') and False else (A, b)
A = A.tocsr()
print("n", A.shape, "nnz", A.nnz, "|b|", np.linalg.norm(b))
perm = slab_permutation(prob.coords, prob.cells, window=0)   # new -> old vertex
perm = np.asarray(perm)
# dof permutation
dperm = (perm[:, None] * NF + np.arange(NF)[None, :]).ravel()
A = A[dperm][:, dperm].tocsr(); b = b[dperm]
coords = prob.coords[perm]

def bicgstab(A, b, Minv, rtol=1e-10, maxit=2000):
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

def block_jacobi(A, groups):
    """groups: list of dof index arrays -> function applying blockdiag(A_gg)^-1"""
    invs = [np.linalg.inv(A[g][:, g].toarray()) for g in groups]
    def apply(x):
        y = np.empty_like(x)
        for g, Bi in zip(groups, invs): y[g] = Bi @ x[g]
        return y
    return apply

def two_level(A, Binv, agg_of_node, nagg):
    n = A.shape[0]
    rows = np.arange(n); cols = agg_of_node[rows // NF] * NF + rows % NF
    P = sp.csr_matrix((np.ones(n), (rows, cols)), shape=(n, nagg * NF))
    # As = A Binv ; Ac = P^T As P
    AsP = np.column_stack([A @ Binv(P[:, j].toarray().ravel()) for j in range(nagg * NF)])
    Ac = P.T @ AsP
    Aci = np.linalg.inv(Ac)
    def Minv(x):
        return Binv(x + P @ (Aci @ (P.T @ x)))
    return Minv

nagg = 15
agg = (np.arange(nv) * nagg) // nv
node_groups = [np.arange(i * NF, (i + 1) * NF) for i in range(nv)]
t0 = time.time(); Bj = block_jacobi(A, node_groups)
M1 = two_level(A, Bj, agg, nagg); x, it = bicgstab(A, b, M1); print("node-block Jacobi + 15 slabs: its", it, "t", time.time() - t0)
x, it = bicgstab(A, b, Bj); print("node-block Jacobi only: its", it)

# spatial clusters of ~7 nodes: greedy within slab order using the graph
import scipy.sparse.csgraph as csg
G = sp.csr_matrix((np.ones(A.nnz), A.indices // NF, A.indptr))  # rows dof -> node cols (dup)
Gn = sp.csr_matrix(A != 0)
nodeG = sp.csr_matrix((np.ones(len(Gn.indices)), (np.repeat(np.arange(A.shape[0]), np.diff(Gn.indptr)) // NF, Gn.indices // NF)), shape=(nv, nv)).tocsr()
def clusters(size):
    assigned = -np.ones(nv, int); groups = []
    for seed in range(nv):
        if assigned[seed] >= 0: continue
        cur = [seed]; assigned[seed] = len(groups); frontier = [seed]
        while len(cur) < size and frontier:
            nxt = []
            for f in frontier:
                for j in nodeG.indices[nodeG.indptr[f]:nodeG.indptr[f + 1]]:
                    if assigned[j] < 0 and len(cur) < size and agg[j] == agg[seed]:
                        assigned[j] = len(groups); cur.append(j); nxt.append(j)
            frontier = nxt
        groups.append(np.array(cur))
    return groups
for size in (4, 7, 14, 28):
    t0 = time.time(); gs = clusters(size)
    dg = [np.concatenate([np.arange(i * NF, (i + 1) * NF) for i in g]) for g in gs]
    Bc = block_jacobi(A, dg)
    Mc = two_level(A, Bc, agg, nagg); x, it = bicgstab(A, b, Mc)
    print("cluster(%d) block Jacobi + 15 slabs: %d clusters (mean %.1f), its %d, t %.1f" % (size, len(gs), nv / len(gs), it, time.time() - t0))
    x, it = bicgstab(A, b, Bc); print("   cluster only: its", it)
print("---- aggregate count / shape")
for na in (8, 15, 30, 60, 120):
    ag = (np.arange(nv) * na) // nv
    M = two_level(A, Bj, ag, na); x, it = bicgstab(A, b, M); print("slabs %d: its %d" % (na, it))
# axial x radial shells
ax = coords[:, np.argmax(coords.max(0) - coords.min(0))]
axis = np.argmax(coords.max(0) - coords.min(0))
oth = [i for i in range(3) if i != axis]
rad = np.hypot(coords[:, oth[0]] - coords[:, oth[0]].mean(), coords[:, oth[1]] - coords[:, oth[1]].mean())
for (nax, nr) in ((15, 2), (15, 3), (15, 4), (8, 2), (5, 3)):
    qa = np.minimum((np.argsort(np.argsort(ax)) * nax) // nv, nax - 1)
    ag = np.zeros(nv, int)
    for a_ in range(nax):
        m = np.where(qa == a_)[0]
        rr = np.argsort(np.argsort(rad[m]))
        ag[m] = a_ * nr + np.minimum((rr * nr) // len(m), nr - 1)
    M = two_level(A, Bj, ag, nax * nr); x, it = bicgstab(A, b, M); print("axial %d x radial %d (=%d aggs): its %d" % (nax, nr, nax * nr, it))
print("---- fewer aggregates / smoothed prolongation")
def P_of(ag, na):
    n = A.shape[0]; rows = np.arange(n); cols = ag[rows // NF] * NF + rows % NF
    return sp.csr_matrix((np.ones(n), (rows, cols)), shape=(n, na * NF))
def apply_cols(op, M):
    M = M.toarray() if sp.issparse(M) else M
    return np.column_stack([op(M[:, j]) for j in range(M.shape[1])])
for na in (1, 2, 4):
    ag = (np.arange(nv) * na) // nv
    M = two_level(A, Bj, ag, na); x, it = bicgstab(A, b, M); print("slabs %d: its %d" % (na, it))
for na in (4, 15):
    ag = (np.arange(nv) * na) // nv
    P = P_of(ag, na)
    for name, T in (("T = Dinv P (current)", apply_cols(Bj, P)),
                    ("T = (I - 2/3 Dinv A) P", P.toarray() - (2/3) * apply_cols(Bj, A @ P.toarray())),
                    ("T = (I - 2/3 Dinv A) Dinv P", (lambda T0: T0 - (2/3) * apply_cols(Bj, A @ T0))(apply_cols(Bj, P)))):
        for rname, R in (("R = P^T", P.T.toarray()), ("R = T^T", T.T)):
            Ac = R @ (A @ T); Aci = np.linalg.inv(Ac)
            Minv = lambda x, T=T, R=R, Aci=Aci: Bj(x) + T @ (Aci @ (R @ x))
            x, it = bicgstab(A, b, Minv); print("slabs %d, %s, %s: its %d" % (na, name, rname, it))
    # multiplicative: coarse first then Jacobi on the updated residual
    T = apply_cols(Bj, P); R = P.T.toarray(); Aci = np.linalg.inv(R @ (A @ T))
    def Mmult(x):
        x1 = T @ (Aci @ (R @ x)); return x1 + Bj(x - A @ x1)
    x, it = bicgstab(A, b, Mmult); print("slabs %d multiplicative (coarse, then Jacobi): its %d (2x SpMV each)" % (na, it))
