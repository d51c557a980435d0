"""Cluster-block smoothers for tools/precond_experiment_r2.py (compact greedy clusters inside the slabs)."""
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
exec(open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'precond_experiment_r2.py')).read().split('print("state"')[0])   # setup: A, b, coords, Dinv, bicgstab, P_slabs, two_level_general ...
print("n", n)
# node graph
Gn = sp.csr_matrix((np.ones(A.nnz), A.indices // NF, np.r_[0, np.cumsum(np.diff(A.indptr))]), shape=(n, nv))
rows_node = np.repeat(np.arange(n) // NF, np.diff(A.indptr))
nodeG = sp.csr_matrix((np.ones(A.nnz), (rows_node, A.indices // NF)), shape=(nv, nv)).tocsr()
nodeG.sum_duplicates()
nagg = 8
agg = (np.arange(nv) * nagg) // nv
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
def block_inv_matrix(groups):
    blocks, idx = [], []
    for g in groups:
        d = np.concatenate([np.arange(i * NF, (i + 1) * NF) for i in g])
        blocks.append(np.linalg.inv(A[d][:, d].toarray())); idx.append(d)
    perm = np.concatenate(idx)
    B = sp.block_diag(blocks, format="csr")
    Pm = sp.csr_matrix((np.ones(n), (perm, np.arange(n))), shape=(n, n))
    return (Pm @ B @ Pm.T).tocsr()
one = np.ones(nv)
Pc = P_slabs(nagg, [one])
for size in (1, 7, 14, 28):
    gs = clusters(size) if size > 1 else [np.array([i]) for i in range(nv)]
    Binv = block_inv_matrix(gs)
    # right-preconditioned two-level (as in the library, Binv instead of node Dinv)
    M = two_level_general(A, Binv, Pc); x, it = bicgstab(A, b, M)
    # left block + right coarse:  Ahat = Binv A ; solve Ahat (I + P Aci P^T) y = Binv b, Aci = (P^T Ahat P)^-1
    Ah = (Binv @ A).tocsr(); bh = Binv @ b
    Pd = Pc.toarray(); Aci = np.linalg.inv(Pd.T @ (Ah @ Pd))
    Mr = lambda v: v + Pd @ (Aci @ (Pd.T @ v))
    y, it2 = bicgstab(Ah, bh, Mr)
    xx2 = Mr(y)
    print("cluster size %2d (%4d clusters): right two-level its %d | left-block + right-coarse its %d (true rel res %.1e)" % (
        size, len(gs), it, it2, np.linalg.norm(A @ xx2 - b) / np.linalg.norm(b)), flush=True)
