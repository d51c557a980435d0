"""Host-side mirror of the reference's operator API for the hot path: ``solve(F == 0, u, bcs, solver_parameters)``
(reference 3D/MPNP_CO2ER_pore.py:789-799, 1D/MPNP_CO2ER_EDL.py:737-742) and the per-step glue around it
(``u_n.assign(u)``, ``compute_vertex_values``, Dirichlet rebuilds).  All arithmetic happens in libgmpnp.so."""
from __future__ import annotations

import numpy as np

from . import backend
from .problem import Problem


class GMPNPSystem:
    """The objects a reference script holds between ``FunctionSpace`` and the time loop: mesh, forms (model tables),
    ``u``/``u_n`` (device resident) and the ``bcs`` list."""

    def __init__(self, problem: Problem, **device_kwargs):
        self.problem = problem
        self.dev = backend.DeviceSolver(problem, **device_kwargs)
        self.nv = problem.coords.shape[0]
        self.nf = problem.nf
        self.newton_iterations = 0
        self.krylov_iterations = 0
        self.last_stats = None

    # u = Function(V) is zero-initialised; u_n = interpolate(u_0, V)  (3D:425-432, 1D:320-326)
    def initialise(self, u0_values):
        u_n = np.tile(np.asarray(u0_values, dtype=np.float64), self.nv)
        self.dev.set_state(np.zeros(self.problem.ndof), u_n)

    def set_bcs(self, dofs, vals):
        self.problem.bc_dofs, self.problem.bc_vals = dofs, vals
        self.dev.set_dirichlet(dofs, vals)

    def set_model(self, model):
        self.problem.model = model
        self.dev.set_model(model)

    def solve(self, solver_parameters=None):
        """``solve(F == 0, u, bcs, solver_parameters=...)``.  RuntimeError on non-convergence, as DOLFIN."""
        opts = backend.newton_options(solver_parameters, dim=self.problem.coords.shape[1])
        try:
            st = self.dev.newton_solve(opts)
        except backend.GmpnpError as e:
            if e.code == backend.ERR_NOT_CONVERGED:
                raise RuntimeError("Newton solver did not converge because maximum number of iterations reached") from e
            raise
        self.newton_iterations += st["iterations"]
        self.krylov_iterations += st["krylov_iterations"]
        self.last_stats = st
        return st

    def vertex_values(self):
        """(nv, nf) array = compute_vertex_values() of every sub-function, file vertex order."""
        return self.dev.get_state().reshape(self.nv, self.nf)

    def assign_previous(self):
        self.dev.assign_previous()

    # post-processing of the reference's drivers: project(+-grad(u_X), W) (3D:884-909, 1D:802-805) and the cell-wise
    # projections of the SUPG parameters (1D:599,651-653), on the device
    def project_gradient(self, f, sign=1.0):
        return self.dev.project_gradient(f, sign=sign)

    def project_cellwise(self, values):
        return self.dev.project_cellwise(values)

    def close(self):
        self.dev.close()


class PartitionedSystem:
    """The same operator surface on ONE problem cut into `nparts` mesh partitions (BASELINE configs[3]):
    the Newton and BiCGStab loops run inside libgmpnp.so across the ranks (gmpnp_group_newton_solve).  `rank` = None keeps
    every rank in this process (one GPU, rehearsal); `rank` = r is the one-process-per-GPU form over RCCL."""

    def __init__(self, problem: Problem, nparts: int, rank: int = None, **device_kwargs):
        from .dist import PartitionedSolver
        self.problem = problem
        self.ps = PartitionedSolver(problem, nparts, rank=rank, **device_kwargs)
        self.dev = self.ps.devs[0]            # this rank's LOCAL partition handle (local vertex numbering)
        self._device_kwargs = {k: v for k, v in device_kwargs.items() if k in ("device_id",)}
        self._post = None                     # unpartitioned handle on the global mesh, for post-processing only
        self.nv = problem.coords.shape[0]
        self.nf = problem.nf
        self.newton_iterations = 0
        self.krylov_iterations = 0
        self.last_stats = None

    def initialise(self, u0_values):
        u_n = np.tile(np.asarray(u0_values, dtype=np.float64), self.nv)
        self.ps.set_state(np.zeros(self.problem.ndof), u_n)

    def set_bcs(self, dofs, vals):
        self.problem.bc_dofs, self.problem.bc_vals = dofs, vals
        self.ps.set_dirichlet(dofs, vals)

    def solve(self, solver_parameters=None):
        opts = backend.newton_options(solver_parameters, dim=3)
        try:
            st = self.ps.newton_solve(opts)
        except backend.GmpnpError as e:
            if e.code == backend.ERR_NOT_CONVERGED:
                raise RuntimeError("Newton solver did not converge because maximum number of iterations reached") from e
            raise
        self.newton_iterations += st["iterations"]
        self.krylov_iterations += st["krylov_iterations"]
        self.last_stats = st
        return st

    def vertex_values(self):
        return self.ps.get_state().reshape(self.nv, self.nf)

    def assign_previous(self):
        self.ps.assign_previous()

    def _post_handle(self):
        """The projections of the drivers' output stage take GLOBAL vertex arrays: they run on an unpartitioned handle of
        the global mesh, created on first use (every rank that writes outputs holds one; it never solves)."""
        if self._post is None:
            self._post = backend.DeviceSolver(self.problem, shared_device=1, **self._device_kwargs)
        return self._post

    def project_gradient(self, f, sign=1.0):
        return self._post_handle().project_gradient(f, sign=sign)

    def project_cellwise(self, values):
        return self._post_handle().project_cellwise(values)

    def close(self):
        if self._post is not None:
            self._post.close()
            self._post = None
        self.ps.close()


def column_medians(vals, cols):
    """``[np.median(vals[:, c]) for c in cols]`` (reference 3D:817-824 takes the medians of four vertex arrays every
    time step) with one selection pass over a contiguous copy: 45 us instead of 230 us for 4 x 3,679 values.  Same
    values as ``np.median`` (middle element, or the mean of the two middle ones); NaNs fall back to it."""
    a = np.ascontiguousarray(np.asarray(vals)[:, list(cols)].T)
    n = a.shape[1]
    if n == 0 or np.isnan(a).any():
        return np.array([np.median(r) for r in a])
    h = n // 2
    if n % 2:
        return np.partition(a, h, axis=1)[:, h]
    p = np.partition(a, (h - 1, h), axis=1)
    return np.array([np.mean(p[i, h - 1:h + 1]) for i in range(p.shape[0])])


def supg_parameters(coords, cells, z, p_prev, project_cellwise, h_vertex=None, fact=1.0, tol=1.0e-14):
    """Nodal SUPG parameters of the PNP stabilisation, reference 1D:597-670 (1D meshes): Pe_i = fact h |grad p| |z_i| / 2
    at the vertices (h = projected cell diameter, |grad p| = projected gradient norm of the PREVIOUS step's potential);
    rho_i = fact h / (2 |z_i| |grad p|) where Pe_i > 1 + tol, else fact^2 h^2 / 4; 0 for uncharged species.
    ``project_cellwise(values)`` is the consistent-mass P1 projection of a cell-wise constant field — in the drivers the
    device's (``DeviceSolver.project_cellwise`` = gmpnp_project_cellwise).  Returns (rho (nv, ns), h_vertex)."""
    assert coords.shape[1] == 1, "the reference stabilises the 1D script only"
    X = coords[cells]
    length = X[:, 1, 0] - X[:, 0, 0]
    if h_vertex is None:
        h_vertex = project_cellwise(np.abs(length))
    gradp = (p_prev[cells[:, 1]] - p_prev[cells[:, 0]]) / length
    norm = project_cellwise(np.abs(gradp))
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
