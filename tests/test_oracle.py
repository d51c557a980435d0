"""Pins of the CPU oracle (oracle/gmpnp_oracle.py): its Jacobian is the derivative of its residual, its closed-form
element integrals agree with brute-force quadrature, DOLFIN's BC / Newton semantics, and the committed golden
vectors (tests/golden, tools/make_golden.py)."""
import copy
import os

import numpy as np
import pytest
import scipy.sparse as sp

import gmpnp_oracle as O
from conftest import GOLDEN, random_state
from gmpnp_amd.model import Quadrature, default_quadrature
from gmpnp_amd.problem import Problem, merge_dirichlet


def _same_rule(dim):
    from gmpnp_amd.model import ufl_estimate_quadrature
    q = ufl_estimate_quadrature(dim)
    return Quadrature(q.lam_j, q.w_j, q.lam_j, q.w_j)


@pytest.mark.parametrize("which", ["pore", "edl", "edl_pnp"])
def test_jacobian_is_derivative_of_residual(which, pore10, edl1):
    """Central finite differences of the element residual vs the analytic element Jacobian, with the SAME rule for
    F and J (here the degree-4 one; the default shares the degree-3 one)."""
    rng = np.random.default_rng(1)
    if which == "pore":
        model, dim, ns = pore10[0].model, 3, 8
        X = rng.uniform(0, 0.05, (4, 4, 3))
    else:
        model, dim, ns = copy.deepcopy(edl1[0].model), 1, 6
        model.steric = which == "edl"
        X = np.sort(rng.uniform(0, 1e-2, (4, 2, 1)), axis=1)
    nn, nf = dim + 1, ns + 1
    U = np.concatenate([rng.uniform(.5, 1.5, (4, nn, ns)), rng.uniform(-1, 0, (4, nn, 1))], 2)
    Un = np.concatenate([rng.uniform(.5, 1.5, (4, nn, ns)), rng.uniform(-1, 0, (4, nn, 1))], 2)
    q = _same_rule(dim)
    Fe, Je = O.element_residual_jacobian(model, q, X, U, Un)
    h = 1e-6
    for b in range(nn):
        for j in range(nf):
            Up, Um = U.copy(), U.copy()
            Up[:, b, j] += h
            Um[:, b, j] -= h
            fd = (O.element_residual_jacobian(model, q, X, Up, Un, False)[0]
                  - O.element_residual_jacobian(model, q, X, Um, Un, False)[0]) / (2 * h)
            ref = Je[:, :, :, b, j]
            assert np.abs(fd - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max()), (which, b, j)


def test_closed_form_integrals_match_brute_force(pore10):
    """Polynomial terms are integrated in closed form; compare the whole residual with a degree-4 evaluation of the
    polynomial integrands by the 14-point rule (exact for them), steric term switched off."""
    rng = np.random.default_rng(2)
    model = copy.deepcopy(pore10[0].model)
    model.steric = False
    X = rng.uniform(0, 0.05, (3, 4, 3))
    U = np.concatenate([rng.uniform(.5, 1.5, (3, 4, 8)), rng.uniform(-1, 0, (3, 4, 1))], 2)
    Un = np.concatenate([rng.uniform(.5, 1.5, (3, 4, 8)), rng.uniform(-1, 0, (3, 4, 1))], 2)
    Fe, _ = O.element_residual_jacobian(model, default_quadrature(3), X, U, Un, False)
    vol, g = O._geometry(X)
    q = default_quadrature(3)
    ref = np.zeros_like(Fe)
    for lam, w in zip(q.lam_j, q.w_j):
        u = np.einsum("b,ebf->ef", lam, U)
        un = np.einsum("b,ebf->ef", lam, Un)
        gu = np.einsum("ebf,ebd->efd", U, g)
        for a in range(4):
            phi, gphi = lam[a], g[:, a, :]
            for i in range(8):
                minusR = model.rc0[i] + u[:, :8] @ model.rc1[i] + sum(
                    model.rc2[i, t] * u[:, bj] * u[:, bk] for t, (bj, bk) in enumerate(model.bil))
                val = (u[:, i] - un[:, i]) * model.inv_dt * phi + np.einsum("ed,ed->e", gu[:, i], gphi) \
                    + model.z[i] * u[:, i] * np.einsum("ed,ed->e", gu[:, 8], gphi) + minusR * phi
                ref[:, a, i] += w * vol * val
            eps = model.eps0 + u[:, :8] @ model.epsc
            rho = u[:, :8] @ (model.z * model.bulk)
            ref[:, a, 8] += w * vol * (-eps * np.einsum("ed,ed->e", gu[:, 8], gphi) + model.q * rho * phi)
    assert np.abs(Fe - ref).max() <= 1e-12 * np.abs(ref).max()


def test_steric_rules_differ_only_slightly(pore10):
    """F uses the degree-3 rule and J the degree-4 rule for the rational term; on smooth states they agree closely."""
    pp, mesh, prob, _ = pore10
    u, un = random_state(mesh.num_vertices, 8)
    F1, _ = O.assemble(prob, u, un, want_jacobian=False, apply_bc=False)
    p2 = copy.copy(prob)
    p2.quad = _same_rule(3)
    F2, _ = O.assemble(p2, u, un, want_jacobian=False, apply_bc=False)
    assert 0 < np.linalg.norm(F1 - F2) / np.linalg.norm(F1) < 1e-4


def test_assembly_bc_semantics(pore10):
    pp, mesh, prob, bnd = pore10
    u, un = random_state(mesh.num_vertices, 8)
    F, A = O.assemble(prob, u, un)
    F0, A0 = O.assemble(prob, u, un, apply_bc=False)
    d = prob.bc_dofs
    assert np.allclose(F[d], u[d] - prob.bc_vals)  # b = x - g
    rows = A[d]
    assert np.allclose(rows.diagonal(k=0) if False else A.diagonal()[d], 1.0)
    assert abs(rows).sum() == pytest.approx(len(d))  # identity rows, nothing else
    free = np.setdiff1d(np.arange(prob.ndof), d)
    assert np.array_equal(F[free], F0[free]) and abs(A[free] - A0[free]).max() == 0.0  # columns are kept
    assert A.nnz == A0.nnz  # pattern kept
    # later DirichletBC wins on shared dofs: rim vertices carry p = V (wall, bc3) not 0 (bc1/bc2)
    rim = np.intersect1d(bnd.dirichlet_vertices[1], bnd.dirichlet_vertices[2])
    assert len(rim) > 0
    vals = dict(zip(prob.bc_dofs, prob.bc_vals))
    assert all(vals[v * 9 + 8] == pp.voltage_scaled for v in rim)


def test_merge_dirichlet_order():
    dofs, vals = merge_dirichlet([([0, 1], 2, 5.0), ([1, 2], 2, 7.0), ([1], 0, 1.0)], 3)
    assert dict(zip(dofs, vals)) == {2: 5.0, 5: 7.0, 8: 7.0, 3: 1.0}


def test_newton_stopping_rule():
    """DOLFIN's residual criterion: tested before the first iteration; 0 iterations when already converged;
    omega < 1 approaches Dirichlet values geometrically."""
    from gmpnp_amd.model import Model
    ns = 6
    m = Model(dim=1, species=list("abcdef"), z=np.zeros(ns), bulk=np.ones(ns), a=np.zeros(ns), inv_dt=1.0, q=0.0,
              eps0=1.0, epsc=np.zeros(ns), rc0=np.zeros(ns), rc1=np.zeros((ns, ns)), bil=[], rc2=np.zeros((ns, 0)),
              steric=False)
    x = np.linspace(0, 1, 6)[:, None]
    cells = np.stack([np.arange(5), np.arange(1, 6)], 1).astype(np.int32)
    dofs, vals = merge_dirichlet([([0], f, 2.0) for f in range(7)], 7)
    prob = Problem(coords=x, cells=cells, model=m, bc_dofs=dofs, bc_vals=vals)
    un = np.ones(prob.ndof)
    u, st = O.newton_solve(prob, np.zeros(prob.ndof), un, relaxation_parameter=1.0)
    assert st.iterations == 1 and st.converged and st.residuals[1] < 1e-10  # linear problem: one full step
    u2, st2 = O.newton_solve(prob, u, un)
    assert st2.iterations == 0 and st2.converged and np.array_equal(u2, u)
    u3, st3 = O.newton_solve(prob, np.zeros(prob.ndof), un, relaxation_parameter=0.9, relative_tolerance=1e-4,
                             absolute_tolerance=1e-4)
    r = np.array(st3.residuals)
    assert np.allclose(r[1:] / r[:-1], 0.1, rtol=1e-6)  # error x0.1 per damped iteration
    assert st3.iterations == 5 if r[0] * 1e-4 > 1e-4 else st3.iterations >= 4
    with pytest.raises(RuntimeError):
        O.newton_solve(prob, np.zeros(prob.ndof), un, relaxation_parameter=0.5, maximum_iterations=3)


def test_golden_elements(pore10, edl1):
    g = np.load(os.path.join(GOLDEN, "elements.npz"))
    from gmpnp_amd.params import edl_parameters, pore_parameters
    pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
    Fe, Je = O.element_residual_jacobian(pp.model, default_quadrature(3), g["X"], g["U"], g["Un"])
    assert np.allclose(Fe, g["Fe"], rtol=1e-12, atol=0) and np.allclose(Je, g["Je"], rtol=1e-12, atol=1e-300)
    ep = edl_parameters(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    Fe1, Je1 = O.element_residual_jacobian(ep.model, default_quadrature(1), g["X1"], g["U1"], g["Un1"])
    assert np.allclose(Fe1, g["Fe1"], rtol=1e-12, atol=0) and np.allclose(Je1, g["Je1"], rtol=1e-12, atol=1e-300)


def test_golden_edl1_first_steps(edl1):
    """Re-run the first two dry-run steps of the 1 um / Cs / V=-5 case (block-tridiagonal LU: fast) against the fixture."""
    ep, mesh, prob = edl1
    g = np.load(os.path.join(GOLDEN, "edl1_steps.npz"))
    out = O.edl_time_loop(ep, copy.copy(prob), 2)
    assert out["newton_its"] == list(g["newton_its"][:2])
    assert np.abs(out["states"] - g["states"][:2]).max() <= 1e-9 * np.abs(g["states"][:2]).max()


def test_project_gradient_of_linear_field(pore10):
    _, mesh, _, _ = pore10
    f = 2.0 * mesh.coords[:, 0] - 3.0 * mesh.coords[:, 2]
    gproj = O.project_gradient(mesh.coords, mesh.cells, f, sign=-1.0)
    assert np.allclose(gproj, np.array([-2.0, 0.0, 3.0])[None, :], atol=1e-9)


@pytest.mark.parametrize("case", ["edl1_pnp", "edl1_li", "edl5_na", "edl10_hohp", "edl1_pnp_supg"])
def test_oracle_reproduces_extra_edl_goldens(case):
    """The committed flag-surface goldens are what the oracle gives today (guards against silent oracle edits)."""
    import os
    from conftest import GOLDEN
    from golden_cases import EXTRA_EDL
    from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
    from gmpnp_amd.params import edl_parameters, utilities_dir
    from gmpnp_amd.problem import edl_problem
    kw, nsteps = EXTRA_EDL[case]
    ep = edl_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), ep.mesh_name))
    out = O.edl_time_loop(ep, edl_problem(ep, mesh), nsteps, stabilization=(kw.get("stabilization") == "Y"))
    g = np.load(os.path.join(GOLDEN, case + "_steps.npz"))
    assert out["newton_its"] == list(g["newton_its"])
    assert np.allclose(out["states"], g["states"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("dim", [1, 3])
def test_supg_terms_jacobian_is_the_derivative(dim):
    """SUPG additions of the PNP stabilisation (reference 1D:687-714): exact Jacobian = central differences of the residual
    addition; in 1D the closed-form residual = 6-point Gauss quadrature of the published integrand (with its grad(u_H) in
    the OH term)."""
    from gmpnp_amd.params import edl_parameters, pore_parameters
    rng = np.random.default_rng(3)
    if dim == 1:
        m = edl_parameters(L_n=1e-6, model="PNP", voltage_multiplier=-2.5).model
    else:
        m = copy.deepcopy(pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9).model)
        m.steric = False
    ns, nc, nn = m.n_species, 4, dim + 1
    X = rng.uniform(0, 0.05, (nc, nn, dim)) if dim == 3 else np.sort(rng.uniform(0, 1e-2, (nc, nn, 1)), axis=1)
    U = np.concatenate([rng.uniform(.5, 1.5, (nc, nn, ns)), rng.uniform(-1, 0, (nc, nn, 1))], 2)
    Un = np.concatenate([rng.uniform(.5, 1.5, (nc, nn, ns)), rng.uniform(-1, 0, (nc, nn, 1))], 2)
    rho = rng.uniform(1e-6, 1e-4, (nc, nn, ns)) * (np.asarray(m.z) != 0)[None, None, :]
    w = np.arange(ns)
    w[1] = 0
    F0, J = O.supg_terms(m, X, U, Un, rho, w)
    eps, Jfd = 1e-6, np.zeros_like(J)
    for b in range(nn):
        for j in range(ns + 1):
            Up, Um = U.copy(), U.copy()
            Up[:, b, j] += eps
            Um[:, b, j] -= eps
            Jfd[:, :, :, b, j] = (O.supg_terms(m, X, Up, Un, rho, w, False)[0] - O.supg_terms(m, X, Um, Un, rho, w, False)[0]) / (2 * eps)
    assert np.abs(J - Jfd).max() / np.abs(J).max() < 1e-8
    assert not J[:, :, ns].any() and not F0[:, :, ns].any()  # the Poisson row is not stabilised
    if dim == 1:
        xs, ws = np.polynomial.legendre.leggauss(6)
        lam, ws = np.stack([(1 - xs) / 2, (1 + xs) / 2], 1), ws / 2
        vol, g = O._geometry(X)
        z = np.asarray(m.z)
        gradp = np.einsum("ea,ea->e", U[:, :, ns], g[:, :, 0])
        gradw = np.einsum("eai,ea->ei", U[:, :, :ns][:, :, w], g[:, :, 0])
        Fq = np.zeros_like(F0)
        for q in range(len(ws)):
            uq, unq, rq = (np.einsum("b,ebi->ei", lam[q], A) for A in (U[:, :, :ns], Un[:, :, :ns], rho))
            R = -(m.rc0[None] + uq @ m.rc1.T + sum(m.rc2[None, :, t] * (uq[:, bj] * uq[:, bk])[:, None] for t, (bj, bk) in enumerate(m.bil)))
            br = m.inv_dt * (uq - unq) + z[None] * gradw * gradp[:, None] + R
            for a in range(2):
                Fq[:, a, :ns] += -(ws[q] * vol)[:, None] * rq * z[None] * br * (gradp * g[:, a, 0])[:, None]
        assert np.abs(F0 - Fq).max() / np.abs(F0).max() < 1e-13


def test_supg_parameters_product_side_equals_oracle(edl1):
    """The Peclet switch of gmpnp_amd.solver.supg_parameters (host glue of the driver; its two projections run on the
    device in the product, here they are handed the oracle's projection) against the oracle's restatement of 1D:597-670,
    including nodes on both branches.  The device projection itself is checked against the oracle's in the GPU tests."""
    from gmpnp_amd.solver import supg_parameters
    ep, mesh, prob = edl1
    nv = mesh.num_vertices
    p = -5.0 * np.exp(-mesh.coords[:, 0] * 1.0e4) - 40.0 * mesh.coords[:, 0] ** 2  # steep near x = 0 (Pe > 1), gentle outside

    def project_cellwise(values):
        return O.project_cellwise(mesh.coords, mesh.cells, values)

    r_prod, h_prod = supg_parameters(mesh.coords, mesh.cells, prob.model.z, p, project_cellwise)
    r_orac, h_orac = O.supg_rho(mesh.coords, mesh.cells, prob.model.z, p)
    assert np.allclose(r_prod, r_orac, rtol=1e-13, atol=0) and np.allclose(h_prod, h_orac, rtol=1e-13, atol=0)
    small = (h_orac ** 2 / 4)[:, None] * (np.asarray(prob.model.z) != 0)[None, :]
    on_large = (r_orac != small).any(axis=1)
    assert on_large.any() and (~on_large).any() and not r_orac[:, 4].any()  # both branches; CO2 (z = 0) untouched
    assert np.allclose(project_cellwise(np.full(len(mesh.cells), 3.0)), 3.0)


def test_config0_first_newton_solve_diverges_in_the_oracle():
    """BASELINE configs[0] (1 um mesh, Cs, voltage_multiplier = -10, README example shape): the UNDAMPED Newton of the
    reference's 1D script (relaxation 1.0, 1D:357-364) started from u = 0 does not converge within 50 iterations on this
    mesh in the oracle; DOLFIN would raise the same RuntimeError.  (V = -5 on the same mesh and V = -10 on the 50 um
    mesh converge: goldens edl1 / edl50.)  Whether FEniCS itself survives that step cannot be checked here."""
    import warnings
    from conftest import _edl
    ep, mesh, prob = _edl(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    nv = mesh.num_vertices
    u0, un = np.zeros(prob.ndof), np.tile(np.r_[np.ones(6), 0.0], nv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, st = O.newton_solve(prob, u0, un, relaxation_parameter=1.0, error_on_nonconvergence=False)
        with pytest.raises(RuntimeError, match="did not converge"):
            O.newton_solve(prob, u0, un, relaxation_parameter=1.0)
    r = np.array(st.residuals)
    assert not st.converged and st.iterations == 50
    assert not np.all(np.isfinite(r)) or r[-1] > 1e-2 * r[0]   # nowhere near the 1e-4 tolerances
