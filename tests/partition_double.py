"""NumPy TEST DOUBLE of the partitioned Newton / BiCGStab iteration (test infrastructure; the product's loops are in
libgmpnp.so: csrc/gmpnp_group.h).  It exists so that what CAN be checked without a GPU is checked at world size 2 on gloo:

* ``gmpnp_amd.dist.partition_plan`` — the ownership ranges, the local vertex order and the send / receive tables the library gets;
* ``gmpnp_amd.dist.host_transport_callbacks`` — the all-reduce and neighbour exchange the library's host-staged transport calls
  back into, driven here with the library's own buffer layout (node-major ghost rows: one contiguous message per neighbour,
  offsets = table pointers x doubles per node, as csrc/gmpnp_group.h::group_transfer computes them);
* the lock-step structure of the algorithm: owned rows only in every dot product, ghost rows refreshed from their owners after
  every preconditioner application, identical scalars (hence identical branches) on every rank.

The local operations come from the CPU oracle (assemble / A x / subdomain LU as preconditioner); the library's preconditioner
(node-block Jacobi + global slab coarse space) is NOT restated here — its parity is the GPU tests' business."""
import numpy as np


class PlanComm:
    """Collectives of one rank's plan (the dict partition_plan returns) over the product's host-transport callbacks."""

    def __init__(self, part, n_local, nf, callbacks):
        self.part, self.nf, self.n_local = part, nf, n_local
        self.allreduce_cb, self.exchange_cb = callbacks
        self.nb = [int(q) for q in part["neighbour_rank"]]
        self.send_ptr, self.recv_ptr = part["send_ptr"].astype(np.int64), part["recv_ptr"].astype(np.int64)
        self.send_v, self.recv_v = part["send_vertices"].astype(np.int64), part["recv_vertices"].astype(np.int64)

    def allreduce_sum(self, values):
        buf = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        assert self.allreduce_cb(buf) == 0
        return buf

    def exchange(self, x):
        """ghost rows of the local nodal vector x (AoS [node][field]) from their owners, in place"""
        per = self.nf
        x2 = x.reshape(self.n_local, per)
        sbuf = np.ascontiguousarray(x2[self.send_v].ravel()) if len(self.send_v) else np.zeros(1)   # k_halo_pack with one vector
        rbuf = np.zeros(max(1, len(self.recv_v) * per))
        s_off = [int(self.send_ptr[j] * per) for j in range(len(self.nb))]
        s_cnt = [int((self.send_ptr[j + 1] - self.send_ptr[j]) * per) for j in range(len(self.nb))]
        r_off = [int(self.recv_ptr[j] * per) for j in range(len(self.nb))]
        r_cnt = [int((self.recv_ptr[j + 1] - self.recv_ptr[j]) * per) for j in range(len(self.nb))]
        assert self.exchange_cb(self.nb, s_off, s_cnt, sbuf, r_off, r_cnt, rbuf) == 0
        if len(self.recv_v):
            x2[self.recv_v] = rbuf[: len(self.recv_v) * per].reshape(-1, per)                        # k_halo_unpack
        return x


class OracleLocalOps:
    """assemble / A x / subdomain-LU preconditioner of one rank's local problem (ghost rows are identity rows)"""

    def __init__(self, problem):
        import scipy.sparse.linalg as spla
        import gmpnp_oracle as O
        self.O, self.spla, self.problem = O, spla, problem
        self.A = self.lu = None

    def assemble(self, u, un, want_jacobian):
        F, A = self.O.assemble(self.problem, u, un, want_jacobian=want_jacobian)
        if want_jacobian:
            self.A, self.lu = A, self.spla.splu(A.tocsc())
        return F

    def spmv(self, x):
        return self.A @ x

    def precond(self, r):
        return self.lu.solve(r)


def bicgstab(ops, comm, n_own_dofs, b, rtol=1e-10, atol=0.0, maxit=10000):
    """Right-preconditioned BiCGStab on the owned dofs; returns (x_local incl. ghosts, iterations, converged)."""
    own = slice(0, n_own_dofs)

    def dots(pairs):
        return comm.allreduce_sum([float(np.dot(a[own], c[own])) for a, c in pairs])

    def apply(p):
        pin = p.copy()
        pin[own.stop:] = 0.0
        z = ops.precond(pin)
        z[own.stop:] = 0.0
        comm.exchange(z)
        y = ops.spmv(z)
        y[own.stop:] = 0.0
        return y

    r = b.copy()
    r[own.stop:] = 0.0
    rhat = r.copy()
    bnorm = np.sqrt(dots([(r, r)])[0])
    tol = max(rtol * bnorm, atol)
    y = np.zeros_like(b)
    if not bnorm > 0.0:
        return np.zeros_like(b), 0, True
    rho = bnorm * bnorm
    p = r.copy()
    it = 0
    tiny = np.finfo(np.float64).tiny

    def restart():
        nonlocal r, rhat, p, rho
        r = b - apply(y)
        r[own.stop:] = 0.0
        rhat = r.copy()
        p = r.copy()
        rho = dots([(r, r)])[0]
        return np.sqrt(rho) <= tol

    while True:
        if it >= maxit:
            return None, it, False
        v = apply(p)
        rv = dots([(rhat, v)])[0]
        if abs(rv) <= 1e-14 * abs(rho) or abs(rho) <= tiny:
            it += 1
            if restart():
                break
            continue
        alpha = rho / rv
        s = r - alpha * v
        if np.sqrt(dots([(s, s)])[0]) <= tol:
            y += alpha * p
            it += 1
            break
        t = apply(s)
        ts, tt, rs, rt = dots([(t, s), (t, t), (rhat, s), (rhat, t)])
        omega = ts / tt
        y += alpha * p + omega * s
        r = s - omega * t
        it += 1
        if np.sqrt(dots([(r, r)])[0]) <= tol:
            break
        rho_new = rs - omega * rt
        if abs(omega) <= 1e-14 or abs(rho_new) <= 1e-14 * abs(rho) * abs(omega):
            if restart():
                break
            continue
        beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        rho = rho_new
    yin = y.copy()
    yin[own.stop:] = 0.0
    x = ops.precond(yin)
    x[own.stop:] = 0.0
    comm.exchange(x)
    return x, it, True


def newton_solve(ops, comm, n_own_dofs, u, un, maximum_iterations=50, relative_tolerance=1e-4, absolute_tolerance=1e-4,
                 relaxation_parameter=1.0, krylov_rtol=1e-10, krylov_maxit=10000):
    """[3P] dolfin::NewtonSolver semantics on a partitioned state; u, un are local (the plan's vertex order)."""
    own = slice(0, n_own_dofs)
    u = comm.exchange(u.copy())
    un = comm.exchange(un.copy())

    def residual(want_j):
        F = ops.assemble(u, un, want_j)
        F[own.stop:] = 0.0
        return F, float(np.sqrt(comm.allreduce_sum([float(np.dot(F[own], F[own]))])[0]))

    stats = {"iterations": 0, "residuals": [], "krylov_per_iteration": [], "converged": False}
    b, r = residual(False)
    r0 = r
    stats["residuals"].append(r)
    conv = lambda res: bool(res / r0 < relative_tolerance or res < absolute_tolerance)  # noqa: E731
    done = conv(r)
    while not done and stats["iterations"] < maximum_iterations:
        b, _ = residual(True)
        dx, kits, ok = bicgstab(ops, comm, n_own_dofs, b, rtol=krylov_rtol, maxit=krylov_maxit)
        assert ok, "partitioned BiCGStab did not converge (%d iterations)" % kits
        stats["krylov_per_iteration"].append(kits)
        u = u - relaxation_parameter * dx
        stats["iterations"] += 1
        b, r = residual(False)
        stats["residuals"].append(r)
        done = conv(r)
    stats["converged"] = done
    return u, stats
