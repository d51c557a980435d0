"""Cylinder mesh generator: the seven meshes the reference names but does not ship
(.MISSING_LARGE_BLOBS) generated here must behave like the shipped ones under the reference's own checks."""
import math

import numpy as np
import pytest

from gmpnp_amd.mesh import mark_pore_boundaries, read_dolfin_xml, write_dolfin_xml
from gmpnp_amd.meshgen import MISSING, cylinder_mesh, default_density, mesh_filename, pore_mesh, wall_tolerance


def _volumes(mesh):
    X = mesh.coords[mesh.cells]
    return np.linalg.det(X[:, 1:, :] - X[:, :1, :]) / 6.0


def _area(mesh, fv):
    x = mesh.coords[fv]
    return 0.5 * np.linalg.norm(np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), axis=1).sum()


def test_the_flat_disc_case_is_sized_by_the_wall_tolerance():
    """L_10_R_50 (aspect 5): the reference's wall test |r^2 - 25| < 5e-3 needs a rim of > 600 vertices before a lateral facet's
    centroid passes it: 111 rings, 0.86 M vertices.  Generated on demand (python -m gmpnp_amd.meshgen --L 10e-9 --R 50e-9), not shipped."""
    nr, nl = default_density(10e-9, 50e-9)
    assert 25.0 * (4.0 / 9.0) * (1.0 - math.cos(2.0 * math.pi / (6 * nr))) <= 0.5 * wall_tolerance(10e-9, 50e-9)
    assert wall_tolerance(10e-9, 50e-9) == 5e-3 and wall_tolerance(100e-9, 50e-9) == 1e-3 and wall_tolerance(10e-9, 5e-9) == 5e-3


@pytest.mark.parametrize("Lnm,Rnm", [m for m in MISSING if m != (10, 50)])
def test_generated_pore_meshes_pass_the_reference_checks(Lnm, Rnm):
    L, R = Lnm * 1e-9, Rnm * 1e-9
    aspect = R / L
    mesh = pore_mesh(L, R)
    nr, nl = default_density(L, R)
    vol = _volumes(mesh)
    assert (vol > 0).all()                                            # positively oriented, like the shipped files
    n_rim = 6 * nr
    poly = 0.5 * n_rim * aspect ** 2 * math.sin(2 * math.pi / n_rim)  # area of the inscribed polygon
    assert abs(vol.sum() - poly) < 1e-12 * max(1.0, poly)             # the tetrahedra tile the prism exactly
    assert abs(vol.sum() / (math.pi * aspect ** 2) - 1.0) < 0.02
    fv, ext, _ = mesh.facets()
    # conforming: every facet belongs to one (boundary) or two (interior) cells, and the boundary is closed
    assert ext.sum() == 2 * (len(mesh.cells) // (3 * nl)) + 2 * n_rim * nl
    bnd = mark_pore_boundaries(mesh, aspect, wall_tolerance(L, R))
    (e1, i1), (e2, i2), (e3, i3) = bnd.counts[1], bnd.counts[2], bnd.counts[3]
    assert i1 == i2 == i3 == 0                                        # no interior facet is swallowed by the wall test
    assert e1 == e3 == len(mesh.cells) // (3 * nl) and e2 == 2 * n_rim * nl and e1 + e2 + e3 == ext.sum()
    # 3D/mesh_tests.py:80-85: assembled wall area against 2 pi R / L
    assert abs(_area(mesh, bnd.ds_facets[2]) / (2 * math.pi * aspect) - 1.0) < 0.01
    assert abs(_area(mesh, bnd.ds_facets[1]) / (math.pi * aspect ** 2) - 1.0) < 0.02
    # quasi-uniform: edge lengths within a factor of 4
    c = mesh.cells
    e = np.concatenate([mesh.coords[c[:, a]] - mesh.coords[c[:, b]] for a in range(4) for b in range(a + 1, 4)])
    ln = np.linalg.norm(e, axis=1)
    assert ln.max() / ln.min() < 4.0


def test_generated_mesh_round_trips_through_dolfin_xml(tmp_path):
    m = cylinder_mesh(0.5, 3, 4)
    path = str(tmp_path / mesh_filename(100e-9, 50e-9)) + ".gz"
    write_dolfin_xml(m, path)
    back = read_dolfin_xml(path)
    assert np.array_equal(back.coords, m.coords) and np.array_equal(back.cells, m.cells)
    assert mesh_filename(100e-9, 50e-9) == "L_100_R_50.xml" and mesh_filename(50e-9, 7.5e-9) == "L_50_R_7.xml"   # int() truncation kept (Q4)
