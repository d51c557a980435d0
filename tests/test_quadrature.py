"""The quadrature tables restated from FIAT (model.py::Quadrature) integrate every monomial up to their degree
exactly: degree 3 for the residual rule, degree 4 for the Jacobian rule."""
import itertools
import math

import numpy as np
import pytest

from gmpnp_amd.model import MAX_QUAD, default_quadrature, to_cquadrature


def simplex_monomial(exps):
    """int over the unit simplex of prod lam_i^e_i (barycentric), normalised by the simplex volume."""
    d = len(exps) - 1
    return math.factorial(d) * math.prod(math.factorial(e) for e in exps) / math.factorial(d + sum(exps))


@pytest.mark.parametrize("dim", [1, 3])
def test_exactness(dim):
    q = default_quadrature(dim)
    for lam, w, deg in ((q.lam_f, q.w_f, 3), (q.lam_j, q.w_j, 4)):
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
    assert (len(q3.w_f), len(q3.w_j), len(q1.w_f), len(q1.w_j)) == (5, 14, 2, 3)
    assert q3.w_f.min() < 0  # the 5-point degree-3 rule has a negative centroid weight
    c = to_cquadrature(q3)
    assert c.nq_f == 5 and c.nq_j == 14 and c.nq_j <= MAX_QUAD
    assert c.lam_j[6][1] == pytest.approx(1 - 3 * 0.1005267652252045)
