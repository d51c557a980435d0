import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def random_state(nv, ns, seed=0):
    """SURVEY §8d micro-benchmark state: u_i ~ U(0.5,1.5), p ~ U(-1,0)."""
    rng = np.random.default_rng(seed)
    u = np.concatenate([rng.uniform(.5, 1.5, (nv, ns)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
    un = np.concatenate([rng.uniform(.5, 1.5, (nv, ns)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
    return u, un


def _pore(L, R, **kw):
    from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
    from gmpnp_amd.params import pore_parameters, utilities_dir
    from gmpnp_amd.problem import pore_problem
    pp = pore_parameters(concentration_elec=0.5, L=L, R=R, **kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
    prob, bnd = pore_problem(pp, mesh)
    return pp, mesh, prob, bnd


def _edl(**kw):
    from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
    from gmpnp_amd.params import edl_parameters, utilities_dir
    from gmpnp_amd.problem import edl_problem
    ep = edl_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), ep.mesh_name))
    return ep, mesh, edl_problem(ep, mesh)


@pytest.fixture(scope="session")
def pore10():
    return _pore(10e-9, 5e-9)


@pytest.fixture(scope="session")
def pore50():
    return _pore(50e-9, 5e-9)


@pytest.fixture(scope="session")
def edl1():
    return _edl(L_n=1e-6, cation="Cs", voltage_multiplier=-5.0)


@pytest.fixture(scope="session")
def edl50():
    return _edl(cation="Cs", voltage_multiplier=-10.0)


@pytest.fixture(scope="session")
def gpu_lib():
    """Build (no-op when up to date) and load the HIP backend; GPU tests fail loudly without it."""
    import __graft_entry__ as ge
    ge.build()
    from gmpnp_amd import backend
    return backend


def box_pore_problem(nx=4, nz=12, half_width=0.1):
    """Small synthetic 'pore': the box [-w,w]^2 x [0,1] cut into 6 tetrahedra per cell, with the 3D pore model,
    ds(2) = lateral faces, ds(3) = the z=1 face and the reference's Dirichlet pattern (3D:460-467).  Used where the
    reference meshes are too large for a CPU test (sparse LU of a 3D Jacobian takes ~10 s on them)."""
    from gmpnp_amd.mesh import Mesh
    from gmpnp_amd.params import pore_parameters
    from gmpnp_amd.problem import Problem, merge_dirichlet
    pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
    xs = np.linspace(-half_width, half_width, nx + 1)
    zs = np.linspace(0.0, 1.0, nz + 1)
    X, Y, Z = np.meshgrid(xs, xs, zs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    idx = np.arange(coords.shape[0]).reshape(nx + 1, nx + 1, nz + 1)
    cells = []
    for i in range(nx):
        for j in range(nx):
            for k in range(nz):
                v = [idx[i + a, j + b, k + c] for a in (0, 1) for b in (0, 1) for c in (0, 1)]  # v[4a+2b+c]
                for t in ((0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7)):
                    cells.append([v[q] for q in t])
    mesh = Mesh(dim=3, coords=coords, cells=np.array(cells, dtype=np.int32))
    fv, ext, _ = mesh.facets()
    P = coords[fv]
    on = lambda test: ext & test(P).all(axis=1)  # noqa: E731
    lateral = on(lambda p: (np.abs(np.abs(p[..., 0]) - half_width) < 1e-12)) | on(lambda p: (np.abs(np.abs(p[..., 1]) - half_width) < 1e-12))
    entry, exit_ = on(lambda p: np.abs(p[..., 2]) < 1e-12), on(lambda p: np.abs(p[..., 2] - 1.0) < 1e-12)
    s1, s2, s3 = np.unique(fv[entry]), np.unique(fv[lateral]), np.unique(fv[exit_])
    dofs, vals = merge_dirichlet([(s1, 8, 0.0), (s3, 8, 0.0), (s2, 8, pp.voltage_scaled), (s1, 4, pp.eq_conc_CO2_scaled),
                                  (s1, 5, pp.eq_conc_CO_scaled), (s1, 6, pp.eq_conc_H2_scaled)], 9)
    prob = Problem(coords=coords, cells=mesh.cells, model=pp.model, wall_facets=fv[lateral], exit_facets=fv[exit_],
                   bc_dofs=dofs, bc_vals=vals)
    return pp, mesh, prob


@pytest.fixture(scope="session")
def boxpore():
    return box_pore_problem()
