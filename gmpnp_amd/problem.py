"""Discrete problem description handed to a backend: mesh arrays, model tables,
boundary facet sets and the merged Dirichlet table, all in FILE vertex order with
``dof = vertex * n_fields + field`` (the node-interleaved layout of the reference's MixedElement,
3D/MPNP_CO2ER_pore.py:404-408, 1D/MPNP_CO2ER_EDL.py:300-304).

``pore_problem`` / ``edl_problem`` restate how the reference turns markers into boundary conditions
(3D:335-382,460-467; 1D:237-254,350-355).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mesh import Mesh, mark_pore_boundaries, pore_wall_tolerance
from .model import Model, Quadrature, default_quadrature


@dataclass
class Problem:
    coords: np.ndarray  # (nv,d)
    cells: np.ndarray  # (nc,d+1)
    model: Model
    quad: Quadrature = None
    wall_facets: np.ndarray = None  # (nw,3) ds(2) facets (3D)
    exit_facets: np.ndarray = None  # (ne,3) ds(3) facets (3D)
    point_vertices: np.ndarray = None  # 1D: vertices carrying the point fluxes (the OHP vertex)
    bc_dofs: np.ndarray = None  # Dirichlet dofs (unique, sorted; the last DirichletBC in the list wins)
    bc_vals: np.ndarray = None
    # SUPG stabilisation of the PNP model (reference 1D:597-722): nodal rho_i (nv, ns), zero where a species is not
    # stabilised, and the species whose gradient enters species i's strong residual (identity except the reference's
    # OH term, which takes grad(u_H): SURVEY Q7).  None = no stabilisation.
    supg_rho: np.ndarray = None
    supg_w: np.ndarray = None

    def __post_init__(self):
        d = self.coords.shape[1]
        if self.quad is None:
            self.quad = default_quadrature(d)
        e3 = np.zeros((0, 3), dtype=np.int32)
        if self.wall_facets is None:
            self.wall_facets = e3
        if self.exit_facets is None:
            self.exit_facets = e3
        if self.point_vertices is None:
            self.point_vertices = np.zeros(0, dtype=np.int32)
        if self.bc_dofs is None:
            self.bc_dofs, self.bc_vals = np.zeros(0, dtype=np.int64), np.zeros(0)

    @property
    def nf(self) -> int:
        return self.model.n_fields

    @property
    def ndof(self) -> int:
        return self.coords.shape[0] * self.nf


def merge_dirichlet(bcs, nf):
    """``bcs``: list of (vertices, field, value) in the reference's list order (3D:467, 1D:355).
    DOLFIN applies them in order, so on a shared dof the LAST one wins."""
    dof_parts, val_parts = [], []
    for verts, fld, val in bcs:
        d = np.asarray(verts, dtype=np.int64).ravel() * nf + int(fld)
        dof_parts.append(d)
        val_parts.append(np.full(d.shape, float(val)))
    if not dof_parts:
        return np.zeros(0, dtype=np.int64), np.zeros(0)
    dofs, vals = np.concatenate(dof_parts), np.concatenate(val_parts)
    # keep the last occurrence of each dof
    rev_dofs, rev_vals = dofs[::-1], vals[::-1]
    uniq, first = np.unique(rev_dofs, return_index=True)
    return uniq, rev_vals[first]


def pore_dirichlet(pp, bnd, co2_value=None):
    """bcs = [bc1..bc6] of reference 3D:460-467 (bc4 rebuilt every step, 3D:835-838)."""
    ns = len(pp.species)
    co2 = pp.eq_conc_CO2_scaled if co2_value is None else co2_value
    s1, s2, s3 = bnd.dirichlet_vertices[1], bnd.dirichlet_vertices[2], bnd.dirichlet_vertices[3]
    # Only the CO2 value changes between time steps: the merged dof set and the positions of the bc4 entries in it are
    # kept on the boundary record (keyed by everything else that enters), so the per-step rebuild is one fill.
    key = (ns, float(pp.voltage_scaled), float(pp.eq_conc_CO_scaled), float(pp.eq_conc_H2_scaled))
    cache = getattr(bnd, "_dirichlet_cache", None)
    if cache is None or cache[0] != key:
        bcs = [(s1, ns, 0.0), (s3, ns, 0.0), (s2, ns, pp.voltage_scaled),
               (s1, 4, co2), (s1, 5, pp.eq_conc_CO_scaled), (s1, 6, pp.eq_conc_H2_scaled)]
        dofs, vals = merge_dirichlet(bcs, ns + 1)
        # field 4 appears in bc4 only, so "later bc wins" never touches these entries
        idx = np.searchsorted(dofs, np.asarray(s1, dtype=np.int64).ravel() * (ns + 1) + 4)
        cache = (key, dofs, vals, idx)
        try:
            bnd._dirichlet_cache = cache
        except AttributeError:  # a record type without room for it: no caching
            return dofs, vals
    _, dofs, vals, idx = cache
    vals = vals.copy()
    vals[idx] = co2
    return dofs, vals


def pore_problem(pp, mesh: Mesh, quad: Quadrature = None, refine: int = 0):
    """Problem + boundary record for the 3D pore (reference 3D:329-382,460-467).  ``refine`` > 0 applies that many
    uniform refinements with inherited markers (gmpnp_amd.mesh.refine_pore); returns (problem, boundaries) of the
    mesh actually used (``problem.coords/cells``)."""
    bnd = mark_pore_boundaries(mesh, pp.aspect_pore, pore_wall_tolerance(pp.L, pp.R))
    for _ in range(refine):
        from .mesh import refine_pore
        mesh, bnd = refine_pore(mesh, bnd)
    dofs, vals = pore_dirichlet(pp, bnd)
    prob = Problem(coords=mesh.coords, cells=mesh.cells, model=pp.model, quad=quad,
                   wall_facets=bnd.ds_facets[2], exit_facets=bnd.ds_facets[3], bc_dofs=dofs, bc_vals=vals)
    return prob, bnd


def pore_hierarchy(pp, mesh: Mesh, refine: int, quad: Quadrature = None):
    """The nested problems of a uniformly refined pore mesh, FINEST FIRST: [(problem, boundaries, parents), ...] with
    ``parents`` (nv, 2) = the vertices of the next-coarser mesh each vertex interpolates from (None on the coarsest, the
    reference mesh itself).  What the multilevel term of the preconditioner is built from (DeviceSolver.attach_coarse_level)."""
    bnd = mark_pore_boundaries(mesh, pp.aspect_pore, pore_wall_tolerance(pp.L, pp.R))
    levels = [(mesh, bnd)]
    for _ in range(refine):
        from .mesh import refine_pore
        levels.append(refine_pore(*levels[-1]))
    out = []
    for m, b in levels[::-1]:
        dofs, vals = pore_dirichlet(pp, b)
        prob = Problem(coords=m.coords, cells=m.cells, model=pp.model, quad=quad, wall_facets=b.ds_facets[2], exit_facets=b.ds_facets[3],
                       bc_dofs=dofs, bc_vals=vals)
        out.append((prob, b, getattr(m, "parents", None)))
    return out


def edl_problem(ep, mesh: Mesh, quad: Quadrature = None):
    """Problem for the 1D EDL (reference 1D:237-254,350-355): all 7 fields pinned to
    (1,..,1,0) at x=1; p = voltage_multiplier at x=0; point fluxes at the x=0 vertex."""
    tol = 1.0e-14
    x = mesh.coords[:, 0]
    _, ext, _ = mesh.facets()
    right = np.nonzero(ext & (np.abs(x - 1.0) < tol))[0]
    left = np.nonzero(ext & (np.abs(x - 0.0) < tol))[0]
    ns = len(ep.species)
    bcs = [(right, f, 1.0) for f in range(ns)] + [(right, ns, 0.0), (left, ns, ep.voltage_scaled)]
    dofs, vals = merge_dirichlet(bcs, ns + 1)
    return Problem(coords=mesh.coords, cells=mesh.cells, model=ep.model, quad=quad,
                   point_vertices=left.astype(np.int32), bc_dofs=dofs, bc_vals=vals)
