"""The quadrature tables restated from FIAT (model.py::Quadrature) integrate every monomial up to their degree
exactly: degree 3 for the residual rule — which the Jacobian shares (model.default_quadrature: decided by the reference's
recorded outputs, tests/test_edl1d_oracle.py) — and degree 4 for the rule the UFL-estimate reading would give J."""
import itertools
import math

import numpy as np
import pytest

from gmpnp_amd.model import MAX_QUAD, default_quadrature, to_cquadrature, ufl_estimate_quadrature


def simplex_monomial(exps):
    """int over the unit simplex of prod lam_i^e_i (barycentric), normalised by the simplex volume."""
    d = len(exps) - 1
    return math.factorial(d) * math.prod(math.factorial(e) for e in exps) / math.factorial(d + sum(exps))


@pytest.mark.parametrize("dim", [1, 3])
def test_exactness(dim):
    q, qu = default_quadrature(dim), ufl_estimate_quadrature(dim)
    assert np.array_equal(q.lam_f, q.lam_j) and np.array_equal(q.w_f, q.w_j) and np.array_equal(q.lam_f, qu.lam_f)
    for lam, w, deg in ((q.lam_f, q.w_f, 3), (qu.lam_j, qu.w_j, 4)):
        assert abs(w.sum() - 1.0) < 1e-13 and np.allclose(lam.sum(1), 1.0, atol=1e-14)
        for exps in itertools.product(range(deg + 1), repeat=dim + 1):
            if sum(exps) > deg:
                continue
            val = (w * np.prod(lam ** np.array(exps), axis=1)).sum()
            assert abs(val - simplex_monomial(exps)) < 1e-13, (dim, deg, exps)
    # the rules are not exact one degree higher (so they are the intended ones, not over-integrating)
    e = (4,) + (0,) * dim
    assert abs((q.w_f * q.lam_f[:, 0] ** 4).sum() - simplex_monomial(e)) > 1e-6


def test_sizes_and_c_image():
    q3, q1 = default_quadrature(3), default_quadrature(1)
    assert (len(q3.w_f), len(q3.w_j), len(q1.w_f), len(q1.w_j)) == (5, 5, 2, 2)
    assert q3.w_f.min() < 0  # the 5-point degree-3 rule has a negative centroid weight
    q3, q1 = ufl_estimate_quadrature(3), ufl_estimate_quadrature(1)
    assert (len(q3.w_f), len(q3.w_j), len(q1.w_f), len(q1.w_j)) == (5, 14, 2, 3)
    c = to_cquadrature(q3)
    assert c.nq_f == 5 and c.nq_j == 14 and c.nq_j <= MAX_QUAD
    assert c.lam_j[6][1] == pytest.approx(1 - 3 * 0.1005267652252045)


def test_the_unpinned_tetrahedron_rules_move_a_solution_by_less_than_1e_6():
    """The one ingredient no reference-held number pins: at WHICH points of a tetrahedron the rational steric term u_i / (1 - S) is
    sampled (FIAT's default degree-3 / degree-4 schemes, restated from memory in gmpnp_amd/model.py).  Every polynomial term is
    exact under any rule of that degree, so what the choice can change is bounded by the quadrature error of the steric term:
    the first time step of the full 3D pore problem (oracle, generated cylinder) with the default rules against a 125-point
    degree-7 rule in both F and J — same Newton count, converged states within 1e-6 of each field's range (4e-7 on the finer
    cylinder of tools/quadrature_sensitivity.py).  Whatever degree-3 rule FFC really picks, it cannot be further away."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    import gmpnp_oracle as O
    from gmpnp_amd.mesh import mark_pore_boundaries
    from gmpnp_amd.meshgen import cylinder_mesh
    from gmpnp_amd.model import Quadrature, default_quadrature
    from gmpnp_amd.params import pore_parameters
    from gmpnp_amd.problem import Problem, pore_dirichlet
    x, w = np.polynomial.legendre.leggauss(5)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    pts, wts = [], []
    for a, wa in zip(x, w):
        for b, wb in zip(x, w):
            for c, wc in zip(x, w):
                p = (a, b * (1 - a), c * (1 - a) * (1 - b))                       # conical (Duffy) product rule, exact to degree 7
                pts.append((1 - sum(p),) + p); wts.append(wa * wb * wc * (1 - a) ** 2 * (1 - b) * 6.0)
    l7, w7 = np.array(pts), np.array(wts)
    assert abs(w7.sum() - 1.0) < 1e-12
    pp = pore_parameters(concentration_elec=0.5, L=10e-9, R=5e-9)
    rings, layers = 3, 6
    mesh = cylinder_mesh(pp.aspect_pore, rings, layers)
    bnd = mark_pore_boundaries(mesh, pp.aspect_pore, 1.5 * pp.aspect_pore ** 2 * (1.0 - np.cos(np.pi / (6 * rings)) ** 2))
    dofs, vals = pore_dirichlet(pp, bnd)
    nv, sols = mesh.num_vertices, []
    for qd in (default_quadrature(3), Quadrature(l7, w7, l7, w7)):
        prob = Problem(coords=mesh.coords, cells=mesh.cells, model=pp.model, quad=qd, wall_facets=bnd.ds_facets[2], exit_facets=bnd.ds_facets[3],
                       bc_dofs=dofs, bc_vals=vals)
        u, st = O.newton_solve(prob, np.zeros(prob.ndof), np.tile(np.r_[np.ones(8), 0.0], nv), relaxation_parameter=0.9,
                               relative_tolerance=1e-12, absolute_tolerance=1e-12, maximum_iterations=60)
        sols.append((u.reshape(nv, 9), st.iterations))
    assert sols[0][1] == sols[1][1]
    rng = sols[1][0].max(0) - sols[1][0].min(0)
    assert (np.abs(sols[0][0] - sols[1][0]).max(0) / rng).max() < 2e-6
