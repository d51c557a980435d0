"""Mesh ingestion and boundary marking for the GMPNP hot path.

Replaces the pieces of DOLFIN the reference drivers touch before the solve:

* ``Mesh("....xml")``                      reference 1D/MPNP_CO2ER_EDL.py:231-234, 3D/MPNP_CO2ER_pore.py:329-332
* ``SubDomain.mark`` / ``MeshFunction``    reference 3D/MPNP_CO2ER_pore.py:335-356, 368-379
* facet sets behind ``DirichletBC(V.sub(k), g, boundary_markers, id)`` and ``ds(id)``
                                            reference 3D/MPNP_CO2ER_pore.py:382, 460-465

Only the DOLFIN-XML dialect of the reference's ``utilities/*.xml(.gz)`` files is read:
``<vertex index x [y z]>`` and ``<interval|tetrahedron index v0 ...>``.  Vertex order of the file is
kept: every user-visible array of the drivers is in file vertex order
(``compute_vertex_values()``, reference 3D:805-813).
"""
from __future__ import annotations

import gzip
import os
import re
from dataclasses import dataclass, field

import numpy as np

_UNMARKED = 9999  # reference 3D:369 ``boundary_markers.set_all(9999)``


@dataclass
class Mesh:
    """P1 simplex mesh in file order. ``dim`` is 1 (intervals) or 3 (tetrahedra)."""

    dim: int
    coords: np.ndarray  # (nv, dim) float64
    cells: np.ndarray  # (nc, dim+1) int32
    _facets: dict = field(default_factory=dict, repr=False)
    # set by refine_uniform on the mesh it returns: (nv, 2) the two vertices of the PARENT mesh each vertex interpolates from
    # (equal: the vertex is a copy of that parent vertex) — the nested-space table of the multilevel preconditioner
    parents: np.ndarray = field(default=None, repr=False)

    @property
    def num_vertices(self) -> int:
        return int(self.coords.shape[0])

    @property
    def num_cells(self) -> int:
        return int(self.cells.shape[0])

    # ---- geometry -------------------------------------------------------------------------
    def cell_volumes(self) -> np.ndarray:
        """Signed-free measure |K| of every cell."""
        x = self.coords[self.cells]  # (nc, d+1, d)
        if self.dim == 1:
            return np.abs(x[:, 1, 0] - x[:, 0, 0])
        e = x[:, 1:, :] - x[:, :1, :]
        return np.abs(np.linalg.det(e)) / 6.0

    # ---- topology -------------------------------------------------------------------------
    def facets(self):
        """All facets of a 3D mesh: ``(fv, exterior, cell_of_facet)``.

        ``fv`` (nf,3) sorted vertex triples, ``exterior`` (nf,) bool (facet has one incident cell),
        ``cell_of_facet`` (nf,) one incident cell.  1D meshes: facets are the vertices.
        """
        if self._facets:
            return self._facets["fv"], self._facets["ext"], self._facets["cell"]
        if self.dim == 1:
            fv = np.arange(self.num_vertices, dtype=np.int32)[:, None]
            cnt = np.bincount(self.cells.ravel(), minlength=self.num_vertices)
            ext = cnt == 1
            cell = np.zeros(self.num_vertices, dtype=np.int32)
            cell[self.cells[:, 0]] = np.arange(self.num_cells)
            cell[self.cells[:, 1]] = np.arange(self.num_cells)
        else:
            c = self.cells
            loc = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
            allf = np.sort(c[:, loc].reshape(-1, 3), axis=1)
            owner = np.repeat(np.arange(self.num_cells, dtype=np.int32), 4)
            fv, first, cnt = np.unique(allf, axis=0, return_index=True, return_counts=True)
            ext = cnt == 1
            cell = owner[first]
            fv = fv.astype(np.int32)
        self._facets = {"fv": fv, "ext": ext, "cell": cell}
        return fv, ext, cell

    def facet_areas(self, fv: np.ndarray) -> np.ndarray:
        if self.dim == 1:
            return np.ones(fv.shape[0])
        x = self.coords[fv]
        return 0.5 * np.linalg.norm(np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), axis=1)


# ---------------------------------------------------------------------------------------------
# DOLFIN-XML reader / writer
# ---------------------------------------------------------------------------------------------
_VERT_RE = re.compile(rb'<vertex\s+index="(\d+)"\s+x="([^"]+)"(?:\s+y="([^"]+)")?(?:\s+z="([^"]+)")?')
_TET_RE = re.compile(rb'<tetrahedron\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"\s+v3="(\d+)"')
_INT_RE = re.compile(rb'<interval\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"')
_HEAD_RE = re.compile(rb'<mesh\s+celltype="(\w+)"\s+dim="(\d+)"')


def read_dolfin_xml(path: str) -> Mesh:
    """Read a DOLFIN-XML mesh (plain or gzip).  Raises ``RuntimeError`` like DOLFIN does when the
    file is missing (reference quirk Q8: a missing mesh aborts the run)."""
    if not os.path.exists(path):
        raise RuntimeError("Unable to open file: %s" % path)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as fh:
        raw = fh.read()
    head = _HEAD_RE.search(raw)
    if head is None:
        raise RuntimeError("Not a DOLFIN-XML mesh: %s" % path)
    celltype, dim = head.group(1).decode(), int(head.group(2))
    verts = _VERT_RE.findall(raw)
    nv = len(verts)
    coords = np.empty((nv, dim), dtype=np.float64)
    for m in verts:
        i = int(m[0])
        for k in range(dim):
            coords[i, k] = float(m[1 + k])
    if celltype == "tetrahedron":
        found = _TET_RE.findall(raw)
        cells = np.empty((len(found), 4), dtype=np.int32)
    elif celltype == "interval":
        found = _INT_RE.findall(raw)
        cells = np.empty((len(found), 2), dtype=np.int32)
    else:
        raise RuntimeError("Unsupported cell type %r in %s" % (celltype, path))
    for m in found:
        cells[int(m[0])] = [int(v) for v in m[1:]]
    return Mesh(dim=dim, coords=coords, cells=cells)


def write_dolfin_xml(mesh: Mesh, path: str) -> None:
    """Write ``mesh`` in the same DOLFIN-XML dialect (``repr`` floats: lossless round trip)."""
    names = "xyz"
    cellname = "interval" if mesh.dim == 1 else "tetrahedron"
    out = ['<?xml version="1.0"?>', '<dolfin xmlns:dolfin="http://fenicsproject.org">',
           '  <mesh celltype="%s" dim="%d">' % (cellname, mesh.dim),
           '    <vertices size="%d">' % mesh.num_vertices]
    for i, x in enumerate(mesh.coords):
        attrs = " ".join('%s="%r"' % (names[k], float(x[k])) for k in range(mesh.dim))
        out.append('      <vertex index="%d" %s />' % (i, attrs))
    out.append("    </vertices>")
    out.append('    <cells size="%d">' % mesh.num_cells)
    for i, c in enumerate(mesh.cells):
        attrs = " ".join('v%d="%d"' % (k, int(v)) for k, v in enumerate(c))
        out.append('      <%s index="%d" %s />' % (cellname, i, attrs))
    out += ["    </cells>", "  </mesh>", "</dolfin>", ""]
    data = "\n".join(out).encode()
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as fh:
        fh.write(data)


def resolve_mesh_path(utilities: str, name: str) -> str:
    """The drivers ask for ``<name>`` exactly as the reference spells it (``L_50_R_5.xml``,
    ``1D_variable_1um_mesh_1090.xml.gz``); the repo ships 3D meshes gzip-compressed, so a
    ``.xml`` request falls through to ``.xml.gz`` when only that exists."""
    p = os.path.join(utilities, name)
    if os.path.exists(p):
        return p
    if not name.endswith(".gz") and os.path.exists(p + ".gz"):
        return p + ".gz"
    return p  # read_dolfin_xml raises


# ---------------------------------------------------------------------------------------------
# Boundary marking with DOLFIN ``SubDomain.mark`` semantics
# ---------------------------------------------------------------------------------------------
def _near(x, x0, eps):
    return np.abs(x - x0) < eps


@dataclass
class PoreBoundaries:
    """Facet markers of the 3D pore (S1 entry=1, S3 exit=3, S2 wall=2) and what the solver needs
    from them.  ``dirichlet_vertices[id]``: vertices of ALL facets carrying ``id`` (interior ones
    included); ``ds_facets[id]``: exterior facets carrying ``id`` (the ``ds(id)`` measure)."""

    markers: np.ndarray  # (nf,) per facet, 9999 = unmarked
    fv: np.ndarray
    exterior: np.ndarray
    dirichlet_vertices: dict
    ds_facets: dict  # id -> (n,3) vertex triples
    counts: dict


def pore_wall_tolerance(L: float, R: float) -> float:
    """reference 3D:350-356 — the absolute tolerance on r**2 is 5e-3 for the L=10 nm, R in {5,50} nm
    meshes and 1e-3 otherwise (float equality on the CLI values, as in the reference)."""
    if (R == 5.0e-9 or R == 50.0e-9) and L == 10.0e-9:
        return 5.0e-3
    return 1.0e-3


def mark_pore_boundaries(mesh: Mesh, aspect_pore: float, wall_tol: float) -> PoreBoundaries:
    """Reproduce reference 3D:335-379: ``inside`` ignores ``on_boundary`` so every facet (interior
    too) whose three vertices AND midpoint pass the test is marked; order entry(1) -> exit(3) ->
    wall(2), later marks overwrite earlier ones."""
    assert mesh.dim == 3
    fv, ext, _ = mesh.facets()
    X = mesh.coords

    def inside_entry(p):
        return _near(p[..., 2], 0.0, 1.0e-12)

    def inside_exit(p):
        return _near(p[..., 2], 1.0, 1.0e-12)

    def inside_wall(p):
        return _near(p[..., 0] ** 2 + p[..., 1] ** 2, aspect_pore ** 2, wall_tol)

    pts = X[fv]  # (nf,3,3)
    mid = pts.mean(axis=1)
    markers = np.full(fv.shape[0], _UNMARKED, dtype=np.int64)
    for test, value in ((inside_entry, 1), (inside_exit, 3), (inside_wall, 2)):
        ok = test(pts).all(axis=1) & test(mid)
        markers[ok] = value
    dirichlet_vertices, ds_facets, counts = {}, {}, {}
    for value in (1, 2, 3):
        sel = markers == value
        dirichlet_vertices[value] = np.unique(fv[sel]).astype(np.int32)
        ds_facets[value] = fv[sel & ext]
        counts[value] = (int((sel & ext).sum()), int((sel & ~ext).sum()))
    return PoreBoundaries(markers=markers, fv=fv, exterior=ext, dirichlet_vertices=dirichlet_vertices,
                          ds_facets=ds_facets, counts=counts)


# ---------------------------------------------------------------------------------------------
# Locality-preserving vertex ordering (file order has no locality)
# ---------------------------------------------------------------------------------------------
def node_graph(mesh: Mesh):
    """CSR node-to-node adjacency (self included), columns sorted."""
    nv, c = mesh.num_vertices, mesh.cells
    k = c.shape[1]
    rows = np.repeat(c, k, axis=1).ravel()
    cols = np.tile(c, (1, k)).ravel()
    key = np.unique(rows.astype(np.int64) * nv + cols.astype(np.int64))
    rows, cols = (key // nv).astype(np.int32), (key % nv).astype(np.int32)
    rowptr = np.zeros(nv + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=nv), out=rowptr[1:])
    return rowptr, cols


def rcm_permutation(mesh: Mesh) -> np.ndarray:
    """Reverse Cuthill-McKee ordering: ``perm[new] = old``."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    rowptr, cols = node_graph(mesh)
    g = csr_matrix((np.ones(cols.size, dtype=np.int8), cols, rowptr), shape=(mesh.num_vertices,) * 2)
    return np.asarray(reverse_cuthill_mckee(g, symmetric_mode=True), dtype=np.int32)


# ---------------------------------------------------------------------------------------------
# Uniform (red) refinement — SURVEY §8f item 1: the reference's mesh generator was never published and its larger
# meshes are missing; refinement also gives problem sizes at which the kernels leave the cache-resident regime.
# ---------------------------------------------------------------------------------------------
def refine_uniform(mesh: Mesh):
    """Split every tetrahedron into 8 (4 corner cells + the inner octahedron cut along the m02-m13 diagonal) or every
    interval into 2.  Returns (fine mesh, edges (ne,2), edge_of) where the midpoint of edge k is vertex nv + k and
    ``edge_of[(a, b)]`` lookups are provided through the returned ``midpoint(a, b)`` function."""
    nv = mesh.num_vertices
    c = mesh.cells
    if mesh.dim == 1:
        mid = 0.5 * (mesh.coords[c[:, 0]] + mesh.coords[c[:, 1]])
        coords = np.concatenate([mesh.coords, mid])
        m = nv + np.arange(len(c), dtype=np.int32)
        cells = np.concatenate([np.stack([c[:, 0], m], 1), np.stack([m, c[:, 1]], 1)]).astype(np.int32)
        par = np.concatenate([np.stack([np.arange(nv), np.arange(nv)], 1), c[:, :2]]).astype(np.int32)
        return Mesh(dim=1, coords=coords, cells=cells, parents=par), None
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
    e = np.sort(c[:, pairs].reshape(-1, 2), axis=1).astype(np.int64)
    key = e[:, 0] * nv + e[:, 1]
    ukey, inv = np.unique(key, return_inverse=True)
    edges = np.stack([ukey // nv, ukey % nv], 1).astype(np.int32)
    coords = np.concatenate([mesh.coords, 0.5 * (mesh.coords[edges[:, 0]] + mesh.coords[edges[:, 1]])])
    m = (nv + inv.reshape(-1, 6)).astype(np.int32)  # m01 m02 m03 m12 m13 m23
    v0, v1, v2, v3 = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    m01, m02, m03, m12, m13, m23 = (m[:, k] for k in range(6))
    kids = [(v0, m01, m02, m03), (v1, m01, m12, m13), (v2, m02, m12, m23), (v3, m03, m13, m23),
            (m01, m02, m03, m13), (m01, m02, m12, m13), (m02, m03, m13, m23), (m02, m12, m13, m23)]
    cells = np.concatenate([np.stack(k, 1) for k in kids]).astype(np.int32)

    def midpoint(a, b):
        a, b = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
        return (nv + np.searchsorted(ukey, a * nv + b)).astype(np.int32)

    par = np.concatenate([np.stack([np.arange(nv), np.arange(nv)], 1), edges]).astype(np.int32)
    return Mesh(dim=3, coords=coords, cells=cells, parents=par), midpoint


def refine_pore(mesh: Mesh, bnd: PoreBoundaries):
    """Refine a marked pore mesh once; child facets inherit their parent's marker (the geometric marking rule of the
    reference, an absolute tolerance on r^2, would swallow interior facets on finer meshes: SURVEY Q5)."""
    fine, midpoint = refine_uniform(mesh)
    ds_facets, dirichlet_vertices, counts = {}, {}, {}
    for value in (1, 2, 3):
        sel = bnd.markers == value
        out = []
        for fv in (bnd.fv[sel & bnd.exterior], bnd.fv[sel & ~bnd.exterior]):
            if len(fv) == 0:
                out.append(np.zeros((0, 3), dtype=np.int32))
                continue
            a, b, c = fv[:, 0], fv[:, 1], fv[:, 2]
            mab, mac, mbc = midpoint(a, b), midpoint(a, c), midpoint(b, c)
            out.append(np.concatenate([np.stack(k, 1) for k in ((a, mab, mac), (b, mab, mbc), (c, mac, mbc), (mab, mbc, mac))]).astype(np.int32))
        ds_facets[value] = out[0]
        dirichlet_vertices[value] = np.unique(np.concatenate([out[0].ravel(), out[1].ravel()])).astype(np.int32)
        counts[value] = (len(out[0]), len(out[1]))
    # marker table of the fine mesh (exterior facets of the fine mesh that descend from marked parents)
    ffv, fext, _ = fine.facets()
    markers = np.full(len(ffv), _UNMARKED, dtype=np.int64)
    nvf = fine.num_vertices
    fkey = (ffv[:, 0].astype(np.int64) * nvf + ffv[:, 1]) * nvf + ffv[:, 2]
    for value in (1, 3, 2):
        kids = np.sort(ds_facets[value], axis=1).astype(np.int64)
        k = (kids[:, 0] * nvf + kids[:, 1]) * nvf + kids[:, 2]
        pos = np.searchsorted(fkey, k)
        ok = (pos < len(fkey)) & (fkey[np.minimum(pos, len(fkey) - 1)] == k)
        markers[pos[ok]] = value
    return fine, PoreBoundaries(markers=markers, fv=ffv, exterior=fext, dirichlet_vertices=dirichlet_vertices,
                                ds_facets=ds_facets, counts=counts)
