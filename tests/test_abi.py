"""The C-ABI library loads on a CPU-only box, exports every symbol include/gmpnp.h declares, its structs have the
size the ctypes images assume, and it refuses to create a solver without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gmpnp.h")


@pytest.fixture(scope="module")
def backend():
    import __graft_entry__ as ge
    ge.build()
    from gmpnp_amd import backend as b
    return b


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gmpnp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(backend):
    lib = backend.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(backend.EXPORTS) == names
    assert lib.gmpnp_version().decode().startswith("gmpnp-mi355x")


def test_struct_layouts_match_the_header(backend, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gmpnp.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(gmpnp_model_t),sizeof(gmpnp_quadrature_t),sizeof(gmpnp_mesh_t),sizeof(gmpnp_newton_options_t),'
                   'sizeof(gmpnp_newton_stats_t),sizeof(gmpnp_linear_stats_t),sizeof(gmpnp_options_t),'
                   'offsetof(gmpnp_model_t,rc2),offsetof(gmpnp_newton_stats_t,ms_assemble));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    from gmpnp_amd.model import CModel, CQuadrature
    b = backend
    assert sizes[:7] == [ctypes.sizeof(CModel), ctypes.sizeof(CQuadrature), ctypes.sizeof(b.CMesh),
                         ctypes.sizeof(b.CNewtonOptions), ctypes.sizeof(b.CNewtonStats), ctypes.sizeof(b.CLinearStats),
                         ctypes.sizeof(b.COptions)]
    assert sizes[7] == CModel.rc2.offset and sizes[8] == b.CNewtonStats.ms_assemble.offset


def test_no_cpu_fallback(backend, pore10):
    """Without a GPU the product path must fail loudly; with one this test is a no-op.  (No PyTorch here: asking PyTorch whether a
    GPU is there would initialise ITS HIP runtime on the card in a process whose libgmpnp.so is bound to the system's — README.)"""
    try:
        dev = backend.DeviceSolver(pore10[2])
    except backend.GmpnpError as e:
        assert e.code == backend.ERR_HIP and "no CPU fallback" in str(e)
        return
    dev.close()
    pytest.skip("GPU present")


def test_missing_library_is_an_error(backend, tmp_path):
    with pytest.raises(RuntimeError):
        backend.load_library(str(tmp_path / "libgmpnp.so"))


def test_newton_options_translation(backend):
    o = backend.newton_options({"nonlinear_solver": "newton", "newton_solver": {
        "linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": 1e-4, "absolute_tolerance": 1e-4,
        "relaxation_parameter": 0.9}})
    assert (o.maximum_iterations, o.relaxation_parameter, o.linear_solver) == (50, 0.9, backend.LINEAR_TWOLEVEL)
    assert o.krylov_relative_tolerance == 1e-10
    d = backend.newton_options({"newton_solver": {"maximum_iterations": 50, "relative_tolerance": 1e-4,
                                                  "absolute_tolerance": 1e-4}})
    assert d.relaxation_parameter == 1.0 and d.linear_solver == backend.LINEAR_TWOLEVEL  # [3P] defaults (1D:357-364)
    d1 = backend.newton_options({"newton_solver": {"maximum_iterations": 50}}, dim=1)
    assert d1.linear_solver == backend.LINEAR_BLOCK_TRIDIAGONAL  # default LU of the 1D script -> direct solver
    j = backend.newton_options({"newton_solver": {"linear_solver": "bicgstab", "preconditioner": "jacobi",
                                                  "krylov_solver": {"relative_tolerance": 1e-8}}})
    assert j.linear_solver == backend.LINEAR_JACOBI and j.krylov_relative_tolerance == 1e-8
    with pytest.warns(UserWarning, match="two-level"):   # 'ilu' is served by the two-level preconditioner, and says so
        i = backend.newton_options({"newton_solver": {"linear_solver": "bicgstab", "preconditioner": "ilu"}})
    assert i.linear_solver == backend.LINEAR_TWOLEVEL and i.krylov_relative_tolerance == 1e-6
    for pc in ("none", "no_such_preconditioner"):
        with pytest.raises(RuntimeError):
            backend.newton_options({"newton_solver": {"linear_solver": "bicgstab", "preconditioner": pc}})
    with pytest.raises(RuntimeError):
        backend.newton_options({"newton_solver": {"linear_solver": "cholmod"}})
    with pytest.raises(RuntimeError):
        backend.newton_options({"nonlinear_solver": "snes"})


def test_slab_permutation(backend, pore10):
    _, mesh, _, _ = pore10
    perm = backend.slab_permutation(mesh.coords, mesh.cells, window=224)
    assert sorted(perm) == list(range(mesh.num_vertices))
    z = mesh.coords[perm, 2]
    w = 224
    means = [z[i:i + w].mean() for i in range(0, len(z), w)]
    assert all(b > a for a, b in zip(means, means[1:]))  # slabs advance along the pore axis
