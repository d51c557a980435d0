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
