"""Mesh ingestion and DOLFIN marking semantics (reference 3D/mesh_tests.py:76-85)."""
import os

import numpy as np
import pytest

from gmpnp_amd.mesh import (Mesh, mark_pore_boundaries, pore_wall_tolerance, read_dolfin_xml, resolve_mesh_path,
                            write_dolfin_xml)
from gmpnp_amd.params import utilities_dir

UTIL = utilities_dir()

# (file, vertices, cells, aspect R/L, tol, exterior S1/S3/S2 facets, interior wall facets, p-Dirichlet vertices)
TABLE = [
    ("L_50_R_1.xml", 1367, 4462, 1 / 50, 1e-3, (0, 0, 2056), 7895, 1367),
    ("L_50_R_2.xml", 1866, 7297, 2 / 50, 1e-3, (95, 95, 2066), 5072, 1476),
    ("L_50_R_2.5.xml", 3530, 16352, 2.5 / 50, 1e-3, (112, 112, 2656), 3771, 1762),
    ("L_50_R_4.xml", 3238, 15057, 4 / 50, 1e-3, (112, 112, 2432), 0, 1330),
    ("L_50_R_5.xml", 3679, 17297, 5 / 50, 1e-3, (112, 112, 2688), 0, 1458),
    ("L_50_R_7.5.xml", 4762, 21895, 7.5 / 50, 1e-3, (240, 240, 3648), 0, 2066),
    ("L_50_R_10.xml", 5411, 24984, 10 / 50, 1e-3, (400, 400, 3840), 0, 2322),
    ("L_80_R_5.xml", 3216, 14920, 5 / 80, 1e-3, (112, 112, 2400), 57, 1320),
    ("L_25_R_5.xml", 6223, 29902, 5 / 25, 1e-3, (400, 400, 3840), 0, 2322),
    ("L_10_R_5.xml", 1767, 7696, 5 / 10, 5e-3, (300, 308, 1152), 0, 882),
]


@pytest.mark.parametrize("name,nv,nc,aspect,tol,ext,interior_wall,npdir", TABLE)
def test_marking_matches_dolfin_semantics(name, nv, nc, aspect, tol, ext, interior_wall, npdir):
    mesh = read_dolfin_xml(resolve_mesh_path(UTIL, name))
    assert (mesh.num_vertices, mesh.num_cells) == (nv, nc)
    b = mark_pore_boundaries(mesh, aspect, tol)
    assert (b.counts[1][0], b.counts[3][0], b.counts[2][0]) == ext
    assert b.counts[2][1] == interior_wall
    pdir = np.unique(np.concatenate([b.dirichlet_vertices[k] for k in (1, 2, 3)]))
    assert len(pdir) == npdir


@pytest.mark.parametrize("name,aspect", [("L_50_R_5.xml", 0.1), ("L_50_R_4.xml", 0.08), ("L_50_R_10.xml", 0.2),
                                          ("L_25_R_5.xml", 0.2)])
def test_wall_area_check(name, aspect):
    """The reference's manual check: assemble(1*ds(2)) next to 2*pi*R/L (3D/mesh_tests.py:80-85)."""
    mesh = read_dolfin_xml(resolve_mesh_path(UTIL, name))
    b = mark_pore_boundaries(mesh, aspect, 1e-3)
    area = mesh.facet_areas(b.ds_facets[2]).sum()
    assert abs(area - 2 * np.pi * aspect) / (2 * np.pi * aspect) < 5e-3  # inscribed polygon: slightly smaller
    assert area < 2 * np.pi * aspect
    vol = mesh.cell_volumes().sum()
    assert abs(vol - np.pi * aspect ** 2) / (np.pi * aspect ** 2) < 1.5e-2


def test_wall_tolerance_branch():
    assert pore_wall_tolerance(10.0e-9, 5.0e-9) == 5e-3
    assert pore_wall_tolerance(10.0e-9, 50.0e-9) == 5e-3
    assert pore_wall_tolerance(50.0e-9, 5.0e-9) == 1e-3


@pytest.mark.parametrize("name,nv", [("1D_variable_1um_mesh_1090.xml.gz", 1091), ("1D_variable_5um_mesh_1490.xml.gz", 1491),
                                      ("1D_variable_10um_mesh_1990.xml.gz", 1991), ("1D_variable_50um_mesh_5990.xml.gz", 5991),
                                      ("1D_variable_200um_mesh_4998.xml.gz", 4999)])
def test_interval_meshes(name, nv):
    mesh = read_dolfin_xml(resolve_mesh_path(UTIL, name))
    assert mesh.dim == 1 and mesh.num_vertices == nv and mesh.num_cells == nv - 1
    x = mesh.coords[:, 0]
    assert x.min() == 0.0 and abs(x.max() - 1.0) < 1e-12
    assert np.all(np.diff(x) > 0)
    assert abs(mesh.cell_volumes().sum() - 1.0) < 1e-12
    _, ext, _ = mesh.facets()
    assert list(np.nonzero(ext)[0]) == [0, nv - 1]


def test_duplicate_mesh_files_are_identical():
    a = read_dolfin_xml(resolve_mesh_path(UTIL, "L_50_R_2.5.xml"))
    b = read_dolfin_xml(resolve_mesh_path(UTIL, "L_100_R_5.xml"))
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.cells, b.cells)


def test_roundtrip_and_missing_file(tmp_path):
    mesh = read_dolfin_xml(resolve_mesh_path(UTIL, "L_10_R_5.xml"))
    for fn in ("m.xml", "m.xml.gz"):
        p = str(tmp_path / fn)
        write_dolfin_xml(mesh, p)
        back = read_dolfin_xml(p)
        assert np.array_equal(back.coords, mesh.coords) and np.array_equal(back.cells, mesh.cells)
    with pytest.raises(RuntimeError):
        read_dolfin_xml(str(tmp_path / "L_50_R_7.xml"))  # the truncated name does not exist
    assert resolve_mesh_path(UTIL, "L_50_R_5.xml").endswith("L_50_R_5.xml.gz")


def test_positive_orientation_not_required():
    m = Mesh(dim=3, coords=np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]), cells=np.array([[0, 2, 1, 3]], dtype=np.int32))
    assert abs(m.cell_volumes()[0] - 1 / 6) < 1e-15
    fv, ext, _ = m.facets()
    assert len(fv) == 4 and ext.all()


def test_uniform_refinement_conserves_geometry_and_markers():
    from gmpnp_amd.mesh import refine_pore, refine_uniform
    mesh = read_dolfin_xml(resolve_mesh_path(UTIL, "L_10_R_5.xml"))
    bnd = mark_pore_boundaries(mesh, 0.5, 5e-3)
    fine, fb = refine_pore(mesh, bnd)
    assert fine.num_cells == 8 * mesh.num_cells
    fv, ext, _ = mesh.facets()
    nedges = len(np.unique(np.sort(mesh.cells[:, [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]]].reshape(-1, 2), axis=1), axis=0))
    assert fine.num_vertices == mesh.num_vertices + nedges
    assert abs(fine.cell_volumes().sum() - mesh.cell_volumes().sum()) < 1e-13
    assert fine.cell_volumes().min() > 0.1 * mesh.cell_volumes().min() / 8
    for k in (1, 2, 3):
        assert fb.counts[k][0] == 4 * bnd.counts[k][0]
        assert abs(fine.facet_areas(fb.ds_facets[k]).sum() - mesh.facet_areas(bnd.ds_facets[k]).sum()) < 1e-12
    ffv, fext, _ = fine.facets()
    assert fext.sum() == 4 * ext.sum()  # the boundary is refined consistently: no hanging nodes
    assert (fb.markers[fext] != 9999).all() and (fb.markers[~fext] == 9999).all()
    m1 = read_dolfin_xml(resolve_mesh_path(UTIL, "1D_variable_1um_mesh_1090.xml.gz"))
    f1, _ = refine_uniform(m1)
    assert f1.num_cells == 2 * m1.num_cells and abs(f1.cell_volumes().sum() - 1.0) < 1e-13


def test_refinement_records_the_nested_space_table():
    """refine_uniform leaves, on the mesh it returns, the two parent vertices every vertex interpolates from: what the multilevel
    term of the preconditioner is built from (gmpnp_attach_coarse_level); pore_hierarchy lists the levels finest first."""
    import numpy as np
    from gmpnp_amd.mesh import read_dolfin_xml, refine_uniform, resolve_mesh_path
    from gmpnp_amd.params import pore_parameters, utilities_dir
    from gmpnp_amd.problem import pore_hierarchy
    pp = pore_parameters(concentration_elec=0.5, L=10e-9, R=5e-9)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
    fine, _ = refine_uniform(mesh)
    par = fine.parents
    nv = mesh.num_vertices
    assert par.shape == (fine.num_vertices, 2) and np.array_equal(par[:nv, 0], np.arange(nv)) and np.array_equal(par[:nv, 0], par[:nv, 1])
    assert (par[nv:, 0] != par[nv:, 1]).all()
    assert np.abs(0.5 * (mesh.coords[par[:, 0]] + mesh.coords[par[:, 1]]) - fine.coords).max() == 0.0
    lv = pore_hierarchy(pp, mesh, 2)
    assert [p.coords.shape[0] for p, _, _ in lv] == [89305, 12109, 1767] and lv[2][2] is None
    for k in (0, 1):   # P1 interpolation of a linear function is exact, and Dirichlet vertices of a coarse level stay Dirichlet on the finer one
        pf, pc, par = lv[k][0], lv[k + 1][0], lv[k][2]
        f = pc.coords @ np.array([0.3, -1.1, 2.0]) + 0.7
        assert np.abs(0.5 * (f[par[:, 0]] + f[par[:, 1]]) - (pf.coords @ np.array([0.3, -1.1, 2.0]) + 0.7)).max() < 1e-13
        copies = np.nonzero(par[:, 0] == par[:, 1])[0]
        bc_f, bc_c = set(pf.bc_dofs.tolist()), set(pc.bc_dofs.tolist())
        assert all(((int(v) * 9 + fld) in bc_f) for v in copies[:200] for fld in range(9) if (int(par[v, 0]) * 9 + fld) in bc_c)
