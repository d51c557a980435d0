"""CPU ORACLE for the GMPNP Newton hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module.
The product path (gmpnp_amd + libgmpnp.so) never does and fails loudly without its HIP library.

What it restates (NumPy/SciPy, fp64), with the reference lines each function follows:

* element residual / exact Jacobian of the mixed P1 forms    3D/MPNP_CO2ER_pore.py:505-769,
                                                              1D/MPNP_CO2ER_EDL.py:383-595
* boundary terms ds(2)/ds(3) (3D intended form)   3D:474-499,560,588,616,644,671,698,724,750
  and the OHP point fluxes of 1D                              1D:371-375,553,738
* assembly + DirichletBC.apply semantics                      3D:460-467,789 ; [3P] DOLFIN Assembler/DirichletBC
* damped Newton with DOLFIN's "residual" stopping rule        3D:789-798, 1D:357-364 ; [3P] dolfin NewtonSolver
* sparse direct LU per Newton iteration                       3D:792 ('mumps'), 1D default LU (UMFPACK)
  -> scipy.sparse.linalg.splu (SuperLU): same algorithm class.

PARITY STATUS: the arithmetic of the reference lives in FEniCS 2019.1.0 / PETSc 3.12.3 / MUMPS 5.2.1 /
UMFPACK (environment.yml:21-27,78,86,110), none of which exists in /root/reference or is installed
or installable here, and the reference has no tests.  What pins this oracle:
* 1D MPNP path — PINNED on the only hot-path outputs the reference stores (1D/Stern_CO2ER.py:66-68, field_OHP and
  eps_rel_OHP at five voltages), through its sibling: oracle/edl1d_oracle.c (an independent C restatement with literal
  Gauss-point integrands) reproduces ALL FIVE vectors to <= 4e-11 over the reference's 20,000-solve staged schedule
  (tests/golden/stern_oracle.json), and this oracle agrees with it on F, J (1e-12) and Newton iterates (1e-9)
  (tests/test_edl1d_oracle.py::test_agrees_with_the_numpy_oracle); the GPU product, which reproduces this oracle's goldens
  step for step, meets the same five vectors (tests/test_gpu_parity.py::test_staged_schedule_reproduces_the_recorded_digits).
  Those vectors also decide the one thing the forms leave open: the Jacobian uses the residual's Gauss rule
  (gmpnp_amd/model.py::default_quadrature).
* 3D path — **parity unpinned** by reference-held numbers (there are none).  What stands in for them: four closed-form cases of
  the 3D forms (tests/closed_forms.py: zero-flux steric-Boltzmann equilibrium for the transport terms, the Debye-Hueckel Bessel
  profile for the Poisson coupling, the literal rate equations for reactions + time term, the exact discrete wall / exit flux
  balance with the literal coefficients), solved by this oracle (tests/test_oracle_pins.py) and by the GPU product
  (tests/test_gpu_parity.py).  tests/test_literal_forms.py evaluates the
  published 3D/1D integrands and parameter formulas literally (no Model tables) against element_residual_jacobian /
  facet_terms; finite-difference Jacobians, closed-form element integrals vs brute-force quadrature, the steric-Boltzmann
  equilibrium, the wall-area check of 3D/mesh_tests.py:80-85 and the L4 scalars are in tests/test_oracle*.py.  The
  quadrature points of the rational steric term on tetrahedra follow FIAT's default degree-3 scheme
  restated from memory of the published FIAT sources (gmpnp_amd/model.py::Quadrature): that ingredient is unpinned
  (bounded at 1e-6 of the field range by tests/test_quadrature.py).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from gmpnp_amd.model import Model, Quadrature, default_quadrature  # noqa: E402  (data classes only)
from gmpnp_amd.problem import Problem, merge_dirichlet  # noqa: E402,F401  (data classes only)


# ---------------------------------------------------------------------------------------------
# Cell integrals
# ---------------------------------------------------------------------------------------------
def _geometry(X):
    """Volumes and constant P1 gradients. X: (nc, d+1, d) -> vol (nc,), grad (nc, d+1, d)."""
    nc, nn, d = X.shape
    if d == 1:
        h = X[:, 1, 0] - X[:, 0, 0]
        grad = np.zeros((nc, 2, 1))
        grad[:, 0, 0], grad[:, 1, 0] = -1.0 / h, 1.0 / h
        return np.abs(h), grad
    T = X[:, 1:, :] - X[:, :1, :]  # rows = edge vectors
    Tinv = np.linalg.inv(T)  # columns = grad phi_1..3
    grad = np.empty((nc, 4, 3))
    grad[:, 1:, :] = np.transpose(Tinv, (0, 2, 1))
    grad[:, 0, :] = -grad[:, 1:, :].sum(axis=1)
    return np.abs(np.linalg.det(T)) / 6.0, grad


def _mass_tables(d):
    """Reference-cell integrals of products of P1 basis functions divided by |K|:
    int phi_a = 1/(d+1); M_ab = (1+delta_ab)/((d+1)(d+2)); T_abc = d! m!/(d+3)!."""
    nn = d + 1
    from math import factorial
    M = (np.ones((nn, nn)) + np.eye(nn)) / ((d + 1) * (d + 2))
    T3 = np.empty((nn, nn, nn))
    for a in range(nn):
        for b in range(nn):
            for c in range(nn):
                mult = np.bincount([a, b, c], minlength=nn)
                m = np.prod([factorial(int(k)) for k in mult])
                T3[a, b, c] = factorial(d) * m / factorial(d + 3)
    return 1.0 / nn, M, T3


def element_residual_jacobian(model: Model, quad: Quadrature, X, U, Un, want_jacobian=True):
    """Element vectors Fe (nc,nn,nf) and matrices Je (nc,nn,nf,nn,nf) of the ``dx`` integrals.

    Species residual  reference 3D:534-750 / 1D:457-593 (PNP variant 1D:430-453 when not model.steric)
    Poisson residual  reference 3D:752-767 / 1D:412-427
    Jacobian = exact Gateaux derivative (what ``solve(F == 0, ...)`` builds) of each
    term; polynomial terms are integrated in closed form (exact, as the reference's degree-3/4
    quadrature is for them); the steric quotient uses ``quad`` (lam_f for F, lam_j for J)."""
    ns, nf = model.n_species, model.n_fields
    nc, nn, d = X.shape
    vol, g = _geometry(X)
    i1, M, T3 = _mass_tables(d)
    u, p = U[:, :, :ns], U[:, :, ns]  # (nc,nn,ns), (nc,nn)
    un = Un[:, :, :ns]
    Kab = np.einsum("ead,ebd->eab", g, g) * vol[:, None, None]  # stiffness
    Mab = M[None] * vol[:, None, None]
    gradp = np.einsum("ea,ead->ed", p, g)  # (nc,d)
    gp_a = np.einsum("ed,ead->ea", gradp, g)  # grad p . grad phi_a
    ubar = u.mean(axis=1)  # (nc,ns)
    a = np.asarray(model.a)
    z = np.asarray(model.z)

    Fe = np.zeros((nc, nn, nf))
    # time + diffusion + migration
    Fe[:, :, :ns] += model.inv_dt * np.einsum("eab,ebi->eai", Mab, u - un)
    Fe[:, :, :ns] += np.einsum("eab,ebi->eai", Kab, u)
    Fe[:, :, :ns] += (vol[:, None, None] * gp_a[:, :, None]) * (z[None, None, :] * ubar[:, None, :])
    # reactions: int (-R_i) phi_a
    Fe[:, :, :ns] += i1 * vol[:, None, None] * model.rc0[None, None, :]
    Fe[:, :, :ns] += np.einsum("eab,ebj,ij->eai", Mab, u, model.rc1)
    for t, (bj, bk) in enumerate(model.bil):
        mono = np.einsum("abc,eb,ec->ea", T3, u[:, :, bj], u[:, :, bk]) * vol[:, None]
        Fe[:, :, :ns] += mono[:, :, None] * model.rc2[None, None, :, t]
    # steric: (int u_i beta) G . grad phi_a
    if model.steric:
        G = np.einsum("j,ebj,ebd->ed", a, u, g)  # sum_j a_j grad u_j
        Gg_a = np.einsum("ed,ead->ea", G, g)

        def quad_moments(lam, w):
            uq = np.einsum("qb,ebi->eqi", lam, u)  # (nc,nq,ns)
            beta = 1.0 / (1.0 - uq @ a)  # (nc,nq)
            wv = w[None, :] * vol[:, None]
            I = np.einsum("eq,eqi->ei", wv * beta, uq)  # int u_i beta
            B = np.einsum("eq,qb->eb", wv * beta, lam)  # int beta phi_b
            C = np.einsum("eq,eqi,qb->eib", wv * beta * beta, uq, lam)  # int u_i beta^2 phi_b
            return I, B, C

        I_f, _, _ = quad_moments(quad.lam_f, quad.w_f)
        Fe[:, :, :ns] += Gg_a[:, :, None] * I_f[:, None, :]
    # Poisson
    epsbar = model.eps0 + ubar @ model.epsc  # eps at the cell mean == mean of the affine eps
    Fe[:, :, ns] += -epsbar[:, None] * np.einsum("eab,eb->ea", Kab, p)
    qzb = model.q * z * np.asarray(model.bulk)
    Fe[:, :, ns] += np.einsum("eab,ebj,j->ea", Mab, u, qzb)
    if not want_jacobian:
        return Fe, None

    Je = np.zeros((nc, nn, nf, nn, nf))
    for i in range(ns):
        Je[:, :, i, :, i] += model.inv_dt * Mab + Kab + z[i] * i1 * vol[:, None, None] * gp_a[:, :, None]
        Je[:, :, i, :, ns] += z[i] * ubar[:, i, None, None] * Kab  # d/dp
        for j in range(ns):
            if model.rc1[i, j] != 0.0:
                Je[:, :, i, :, j] += model.rc1[i, j] * Mab
        for t, (bj, bk) in enumerate(model.bil):
            c = model.rc2[i, t]
            if c == 0.0:
                continue
            Je[:, :, i, :, bj] += c * vol[:, None, None] * np.einsum("abc,ec->eab", T3, u[:, :, bk])
            Je[:, :, i, :, bk] += c * vol[:, None, None] * np.einsum("abc,ec->eab", T3, u[:, :, bj])
    if model.steric:
        I_j, B_j, C_j = quad_moments(quad.lam_j, quad.w_j)
        gg = np.einsum("ead,ebd->eab", g, g)
        for i in range(ns):
            Je[:, :, i, :, i] += Gg_a[:, :, None] * B_j[:, None, :]
            for j in range(ns):
                Je[:, :, i, :, j] += a[j] * (Gg_a[:, :, None] * C_j[:, i, None, :] + I_j[:, i, None, None] * gg)
    Kp = np.einsum("eab,eb->ea", Kab, p)
    for j in range(ns):
        # d eps(ubar)/d u_jb = epsc_j/(d+1) for every b
        Je[:, :, ns, :, j] += -model.epsc[j] * i1 * Kp[:, :, None] + qzb[j] * Mab
    Je[:, :, ns, :, ns] += -epsbar[:, None, None] * Kab
    return Fe, Je


def supg_terms(model: Model, X, U, Un, rho_e, w_index, want_jacobian=True):
    """SUPG stabilisation of the PNP model, reference 1D/MPNP_CO2ER_EDL.py:687-714:

        F_stab = - sum_i  rho_i z_i [ (u_i - u_i^n)/(dt L_D) + z_i grad(w_i).grad(p) + R_i ] grad(p).grad(v_i) dx

    with P1 nodal rho_i (zero for unstabilised species), w_i = u_i except w_OH = u_H (the reference's typo),
    and R_i the production rate (the model tables hold -R_i).  Everything is polynomial of degree <= 3 on a P1 element:
    closed form, exact like the reference's degree-3 rule.  Returns the ADDITIONS (Fe, Je) to the element vectors."""
    ns, nf = model.n_species, model.n_fields
    nc, nn, d = X.shape
    vol, g = _geometry(X)
    i1, M, T3 = _mass_tables(d)
    u, p, un = U[:, :, :ns], U[:, :, ns], Un[:, :, :ns]
    z = np.asarray(model.z)
    Mab = M[None] * vol[:, None, None]
    gradp = np.einsum("ea,ead->ed", p, g)
    gp_a = np.einsum("ed,ead->ea", gradp, g)                 # grad p . grad phi_a
    gg = np.einsum("ead,ebd->eab", g, g)
    w = u[:, :, np.asarray(w_index)]                          # (nc,nn,ns)
    gradw = np.einsum("eai,ead->eid", w, g)                   # (nc,ns,d)
    wp = np.einsum("eid,ed->ei", gradw, gradp)                # grad w_i . grad p
    gw_b = np.einsum("eid,ebd->eib", gradw, g)                # grad w_i . grad phi_b
    rbar = rho_e.mean(axis=1)                                 # (nc,ns)
    rM = np.einsum("eai,eab->eib", rho_e, Mab)                # sum_a rho_ia M_ab
    # S_i = int rho_i [ ... ]
    S = model.inv_dt * np.einsum("eib,ebi->ei", rM, u - un)
    S += z[None, :] * wp * vol[:, None] * rbar
    S -= model.rc0[None, :] * vol[:, None] * rbar
    S -= np.einsum("ij,eib,ebj->ei", model.rc1, rM, u)
    rT = np.einsum("eai,abc->eibc", rho_e, T3) * vol[:, None, None, None]   # sum_a rho_ia T_abc |K|
    for t, (bj, bk) in enumerate(model.bil):
        S -= model.rc2[None, :, t] * np.einsum("eibc,eb,ec->ei", rT, u[:, :, bj], u[:, :, bk])
    Fe = np.zeros((nc, nn, nf))
    Fe[:, :, :ns] = -z[None, None, :] * gp_a[:, :, None] * S[:, None, :]
    if not want_jacobian:
        return Fe, None
    Je = np.zeros((nc, nn, nf, nn, nf))
    for i in range(ns):
        if z[i] == 0.0:
            continue
        dS_du = np.zeros((nc, nn, ns))                        # [b, j]
        dS_du[:, :, i] += model.inv_dt * rM[:, i, :]
        dS_du[:, :, w_index[i]] += z[i] * gp_a * (vol * rbar[:, i])[:, None]
        dS_du -= rM[:, i, :, None] * model.rc1[i][None, None, :]
        for t, (bj, bk) in enumerate(model.bil):
            c = model.rc2[i, t]
            if c != 0.0:
                dS_du[:, :, bj] -= c * np.einsum("ebc,ec->eb", rT[:, i], u[:, :, bk])
                dS_du[:, :, bk] -= c * np.einsum("ecb,ec->eb", rT[:, i], u[:, :, bj])
        dS_dp = z[i] * gw_b[:, i, :] * (vol * rbar[:, i])[:, None]   # [b]
        Je[:, :, i, :, :ns] += -z[i] * gp_a[:, :, None, None] * dS_du[:, None, :, :]
        Je[:, :, i, :, ns] += -z[i] * (gg * S[:, i, None, None] + gp_a[:, :, None] * dS_dp[:, None, :])
    return Fe, Je


# ---------------------------------------------------------------------------------------------
# Exterior-facet / point integrals
# ---------------------------------------------------------------------------------------------
def facet_terms(prob: Problem, u2d, want_jacobian=True):
    """Boundary contributions as COO triplets.  u2d: (nv,nf).

    3D wall ds(2): J_X |f|/3 per facet vertex (3D:474-481; X in OH, CO2, CO, H2)
    3D exit ds(3): kappa_X int (u_X-1) phi_a  (3D:484-499; all 8 species); Jacobian kappa_X |f|(1+delta_ab)/12
    1D OHP point:  F[X, vertex] += J_X        (1D:371-375,553,738; SURVEY Q6: both ends are integrated by
                   the unrestricted ds but every dof at x=1 is Dirichlet, so only the OHP vertex matters)"""
    m, nf, ns = prob.model, prob.nf, prob.model.n_species
    Fb = np.zeros(prob.ndof)
    rows, cols, vals = [], [], []
    X = prob.coords

    def areas(fv):
        x = X[fv]
        return 0.5 * np.linalg.norm(np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), axis=1)

    if len(prob.wall_facets):
        fv = prob.wall_facets
        ar = areas(fv)
        for i in np.nonzero(m.wall_flux)[0]:
            np.add.at(Fb, (fv * nf + i).ravel(), np.repeat(m.wall_flux[i] * ar / 3.0, 3))
    if len(prob.exit_facets):
        fv = prob.exit_facets
        ar = areas(fv)
        Mf = (np.ones((3, 3)) + np.eye(3)) / 12.0
        for i in np.nonzero(m.exit_kappa)[0]:
            ui = u2d[fv, i]  # (nfac,3)
            contrib = m.exit_kappa[i] * ar[:, None] * (ui @ Mf.T - 1.0 / 3.0)
            np.add.at(Fb, (fv * nf + i).ravel(), contrib.ravel())
            if want_jacobian:
                r = np.repeat(fv * nf + i, 3, axis=1).ravel()
                c = np.tile(fv * nf + i, (1, 3)).ravel()
                v = (m.exit_kappa[i] * ar[:, None, None] * Mf[None]).ravel()
                rows.append(r), cols.append(c), vals.append(v)
    for v in prob.point_vertices:
        Fb[int(v) * nf: int(v) * nf + ns] += m.point_flux
    if rows:
        return Fb, (np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
    return Fb, None


# ---------------------------------------------------------------------------------------------
# Global assembly ([3P] DOLFIN Assembler + DirichletBC.apply)
# ---------------------------------------------------------------------------------------------
def _pattern(prob: Problem, dofs, Jb):
    """COO -> CSR scatter map of the assembled Jacobian, built once per Problem and cached."""
    cache = getattr(prob, "_oracle_pattern", None)
    if cache is not None:
        return cache
    n = dofs.shape[1]
    rows = np.repeat(dofs, n, axis=1).ravel().astype(np.int64)
    cols = np.tile(dofs, (1, n)).ravel().astype(np.int64)
    if Jb is not None:
        rows = np.concatenate([rows, Jb[0]])
        cols = np.concatenate([cols, Jb[1]])
    key, pos = np.unique(rows * prob.ndof + cols, return_inverse=True)
    r, c = key // prob.ndof, key % prob.ndof
    indptr = np.zeros(prob.ndof + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=prob.ndof), out=indptr[1:])
    cache = {"pos": pos, "nnz": len(key), "indices": c.astype(np.int32), "indptr": indptr.astype(np.int32)}
    prob._oracle_pattern = cache
    return cache


def assemble(prob: Problem, u, un, want_jacobian=True, apply_bc=True):
    """b = assemble(F) (+ ``b[dof] = x[dof]-g`` rows) and A = assemble(J) (+ identity rows)."""
    nf, nv = prob.nf, prob.coords.shape[0]
    u2d, un2d = u.reshape(nv, nf), un.reshape(nv, nf)
    cells = prob.cells
    Fe, Je = element_residual_jacobian(prob.model, prob.quad, prob.coords[cells], u2d[cells], un2d[cells],
                                       want_jacobian)
    if getattr(prob, "supg_rho", None) is not None:
        Fs, Js = supg_terms(prob.model, prob.coords[cells], u2d[cells], un2d[cells], prob.supg_rho[cells], prob.supg_w,
                            want_jacobian)
        Fe = Fe + Fs
        if want_jacobian:
            Je = Je + Js
    nn = cells.shape[1]
    dofs = (cells[:, :, None] * nf + np.arange(nf)[None, None, :]).reshape(len(cells), nn * nf)
    F = np.bincount(dofs.ravel(), weights=Fe.reshape(len(cells), -1).ravel(), minlength=prob.ndof)
    Fb, Jb = facet_terms(prob, u2d, want_jacobian)
    F += Fb
    A = None
    if want_jacobian:
        pat = _pattern(prob, dofs, Jb)
        vals = Je.reshape(len(cells), -1).ravel()
        if Jb is not None:
            vals = np.concatenate([vals, Jb[2]])
        data = np.bincount(pat["pos"], weights=vals, minlength=pat["nnz"])
        A = sp.csr_matrix((data, pat["indices"], pat["indptr"]), shape=(prob.ndof, prob.ndof))
    if apply_bc and len(prob.bc_dofs):
        F[prob.bc_dofs] = u[prob.bc_dofs] - prob.bc_vals
        if A is not None:
            A = apply_identity_rows(A, prob.bc_dofs)
    return F, A


def apply_identity_rows(A: sp.csr_matrix, dofs):
    """DirichletBC.apply(A): zero the rows, 1 on the diagonal, columns kept."""
    A = A.tocsr(copy=True)
    mask = np.zeros(A.shape[0], dtype=bool)
    mask[dofs] = True
    rowid = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    kill = mask[rowid]
    A.data[kill & (A.indices != rowid)] = 0.0
    A.data[kill & (A.indices == rowid)] = 1.0
    return A


# ---------------------------------------------------------------------------------------------
# Newton ([3P] dolfin::NewtonSolver, criterion "residual")
# ---------------------------------------------------------------------------------------------
@dataclass
class NewtonStats:
    iterations: int = 0
    converged: bool = False
    residuals: list = field(default_factory=list)  # ||b||_2 at iteration 0..its
    t_assemble: float = 0.0
    t_linear: float = 0.0


def newton_solve(prob: Problem, u, un, maximum_iterations=50, relative_tolerance=1e-4, absolute_tolerance=1e-4,
                 relaxation_parameter=1.0, error_on_nonconvergence=True, linear_solve=None):
    """x <- x - omega * A^{-1} b until ||b||/||b0|| < rtol or ||b|| < atol, tested BEFORE the first
    iteration and after every update (reference 3D:789-798, 1D:357-364)."""
    import time
    u = np.array(u, dtype=np.float64, copy=True)
    st = NewtonStats()
    t0 = time.perf_counter()
    b, _ = assemble(prob, u, un, want_jacobian=False)
    st.t_assemble += time.perf_counter() - t0
    r = float(np.linalg.norm(b))
    r0 = r
    st.residuals.append(r)

    def conv(res):
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.float64(res) / np.float64(r0)
        return bool(rel < relative_tolerance or res < absolute_tolerance)

    done = conv(r)
    while not done and st.iterations < maximum_iterations:
        t0 = time.perf_counter()
        b, A = assemble(prob, u, un, want_jacobian=True)
        st.t_assemble += time.perf_counter() - t0
        t0 = time.perf_counter()
        dx = linear_solve(A, b) if linear_solve is not None else spla.splu(A.tocsc()).solve(b)
        st.t_linear += time.perf_counter() - t0
        u -= relaxation_parameter * dx
        st.iterations += 1
        t0 = time.perf_counter()
        b, _ = assemble(prob, u, un, want_jacobian=False)
        st.t_assemble += time.perf_counter() - t0
        r = float(np.linalg.norm(b))
        st.residuals.append(r)
        done = conv(r)
    st.converged = done
    if not done and error_on_nonconvergence:
        raise RuntimeError("Newton solver did not converge because maximum number of iterations reached")
    return u, st


# ---------------------------------------------------------------------------------------------
# Consistent-mass L2 projection of a cell-wise constant gradient (reference 1D:802-805, 3D:884-909)
# ---------------------------------------------------------------------------------------------
def project_gradient(coords, cells, f, sign=1.0):
    """``project(sign*grad(f), W)`` on P1^d: solve M g_k = int (sign d_k f) phi for every component k.
    Returns (nv,d)."""
    nv, d = coords.shape
    X = coords[cells]
    vol, g = _geometry(X)
    i1, M, _ = _mass_tables(d)
    nn = d + 1
    gradf = np.einsum("ea,ead->ed", f[cells], g)  # (nc,d)
    rows = np.repeat(cells, nn, axis=1).ravel()
    cols = np.tile(cells, (1, nn)).ravel()
    vals = (M[None] * vol[:, None, None]).ravel()
    Mg = sp.coo_matrix((vals, (rows, cols)), shape=(nv, nv)).tocsc()
    lu = spla.splu(Mg)
    out = np.empty((nv, d))
    for k in range(d):
        rhs = np.bincount(cells.ravel(), weights=np.repeat(sign * gradf[:, k] * vol * i1, nn), minlength=nv)
        out[:, k] = lu.solve(rhs)
    return out


def project_cellwise(coords, cells, values):
    """``project(f, Y).compute_vertex_values()`` of a cell-wise constant f onto P1: consistent-mass L2 projection
    (reference 1D:599 ``project(CellDiameter(mesh))``, 1D:651-653 ``project(sqrt(inner(grad(u_np), grad(u_np))))``)."""
    nv, d = coords.shape
    vol, _ = _geometry(coords[cells])
    i1, M, _ = _mass_tables(d)
    nn = d + 1
    rows = np.repeat(cells, nn, axis=1).ravel()
    cols = np.tile(cells, (1, nn)).ravel()
    Mg = sp.coo_matrix(((M[None] * vol[:, None, None]).ravel(), (rows, cols)), shape=(nv, nv)).tocsc()
    rhs = np.bincount(cells.ravel(), weights=np.repeat(values * vol * i1, nn), minlength=nv)
    return spla.splu(Mg).solve(rhs)


def supg_rho(coords, cells, z, p_prev, h_vertex=None, fact=1.0, tol=1e-14):
    """Nodal SUPG parameters of reference 1D:597-670: Pe_i = fact h |grad p| |z_i| / 2 at the vertices (h and |grad p|
    projected onto P1); rho_i = fact h / (2 |z_i| |grad p|) where Pe_i > 1, else fact^2 h^2 / 4; 0 for z_i = 0.
    Returns (rho (nv, ns), h_vertex)."""
    X = coords[cells]
    if h_vertex is None:
        diam = np.abs(X[:, 1, 0] - X[:, 0, 0]) if coords.shape[1] == 1 else None
        if diam is None:
            raise NotImplementedError("CellDiameter in 3D")
        h_vertex = project_cellwise(coords, cells, diam)
    _, g = _geometry(X)
    gradp = np.einsum("ea,ead->ed", p_prev[cells], g)
    norm = project_cellwise(coords, cells, np.sqrt((gradp ** 2).sum(axis=1)))
    z = np.asarray(z, dtype=float)
    rho = np.zeros((coords.shape[0], len(z)))
    rho_small = fact ** 2 * h_vertex ** 2 / 4
    for i, zi in enumerate(z):
        if zi == 0:
            continue
        Pe = fact * h_vertex * norm * abs(zi) / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            rho_large = fact * h_vertex / (2 * abs(zi) * norm)
        rho[:, i] = np.where(Pe > 1.0 + tol, rho_large, rho_small)
    return rho, h_vertex


# ---------------------------------------------------------------------------------------------
# Time loops (reference 3D:783-858 and 1D:633-796), oracle-side restatement for golden vectors
# ---------------------------------------------------------------------------------------------
def pore_time_loop(pp, prob, bnd, n_steps, newton_kwargs=None, verbose=False, cation_from_electroneutrality=False):
    """Backward-Euler loop of the 3D pore driver: Newton from the previous u (zeros at step 0, SURVEY §3.3
    item 6), median -> Sechenov -> new CO2 Dirichlet value (3D:817-838), u_n.assign(u) (3D:856).
    Returns dict(states (n_steps, ndof), newton_its, residuals, co2_bc)."""
    from gmpnp_amd.problem import pore_dirichlet
    kw = dict(maximum_iterations=50, relative_tolerance=1e-4, absolute_tolerance=1e-4, relaxation_parameter=0.9)
    kw.update(newton_kwargs or {})
    nv, nf = prob.coords.shape[0], prob.nf
    u = np.zeros(prob.ndof)
    un = np.tile(np.r_[np.ones(nf - 1), 0.0], nv)
    out = {"states": [], "newton_its": [], "residuals": [], "co2_bc": []}
    for n in range(n_steps):
        u, st = newton_solve(prob, u, un, **kw)
        u2 = u.reshape(nv, nf)
        med_cat = np.median(u2[:, 7])
        if cation_from_electroneutrality:  # reaction-diffusion variant, reference 3D/rxn_diff_CO2ER_pore.py:564-568
            b = pp.bulk_conc
            med_cat = (np.median(u2[:, 2]) * b["HCO3"] + 2 * np.median(u2[:, 3]) * b["CO32"] + np.median(u2[:, 1]) * b["OH"]
                       - np.median(u2[:, 0]) * b["H"]) / b[pp.cation]
        co2 = pp.sechenov_co2_scaled(np.median(u2[:, 1]), np.median(u2[:, 2]), np.median(u2[:, 3]), med_cat)
        prob.bc_dofs, prob.bc_vals = pore_dirichlet(pp, bnd, co2)
        un = u.copy()
        out["states"].append(u.copy()); out["newton_its"].append(st.iterations)
        out["residuals"].append(list(st.residuals)); out["co2_bc"].append(co2)
        if verbose:
            print("step", n, "its", st.iterations, "res", st.residuals[-1], "co2", co2, flush=True)
    out["states"] = np.array(out["states"])
    return out


def edl_time_loop(ep, prob, n_steps, newton_kwargs=None, verbose=False, stabilization=False):
    """Dry-run loop of the 1D EDL driver (1D:633-796) without stabilisation: Newton (omega = 1), optional
    proton-flux controller (1D:766-793), u_n.assign(u)."""
    import copy
    kw = dict(maximum_iterations=50, relative_tolerance=1e-4, absolute_tolerance=1e-4, relaxation_parameter=1.0)
    kw.update(newton_kwargs or {})
    nv, nf = prob.coords.shape[0], prob.nf
    u = np.zeros(prob.ndof)
    un = np.tile(np.r_[np.ones(nf - 1), 0.0], nv)
    frac = ep.current_H_frac
    out = {"states": [], "newton_its": [], "residuals": [], "current_H_frac": []}
    prob.model = copy.deepcopy(prob.model)
    iH, iOH = ep.species.index("H"), ep.species.index("OH")
    h_vertex = None
    for n in range(n_steps):
        if stabilization:  # PNP + SUPG (1D:650-722): rho from the PREVIOUS step's potential, OH takes grad(u_H) (Q7)
            assert not prob.model.steric, "the reference stabilises the PNP model only"
            prob.supg_rho, h_vertex = supg_rho(prob.coords, prob.cells, prob.model.z, un.reshape(nv, nf)[:, nf - 1], h_vertex)
            w = np.arange(nf - 1); w[iOH] = iH
            prob.supg_w = w
        u, st = newton_solve(prob, u, un, **kw)
        f = u.reshape(nv, nf)[0, iH]
        if ep.H_OHP is not None:
            H = ep.H_OHP
            if f < 0:
                frac = frac / 1.1
            elif f < (H - 0.05):
                frac = frac / 1.05
            elif f < (H - 0.025):
                frac = frac / 1.01
            elif (f > H and f <= (H + 0.4) and frac <= 1.0):
                frac = frac * 1.04
            elif f > (H + 0.4) and frac <= 1.0:
                frac = frac * 1.15
            JH, JOH = ep.ohp_fluxes(frac)
            prob.model.point_flux[iH], prob.model.point_flux[iOH] = JH, JOH
        un = u.copy()
        out["states"].append(u.copy()); out["newton_its"].append(st.iterations)
        out["residuals"].append(list(st.residuals)); out["current_H_frac"].append(frac)
        if verbose:
            print("step", n, "its", st.iterations, "res", st.residuals[-1], flush=True)
    out["states"] = np.array(out["states"])
    return out
