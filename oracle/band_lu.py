"""ctypes front end of oracle/band_lu_omp.c — CPU BASELINE LEG, TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline and tests/).

A threaded direct solve of one Newton system of the 3D path: the node-block band of the Jacobian in the library's slab order,
factored right-looking with OpenMP over the window behind each pivot.  Stand-in for the reference's MUMPS (3D/MPNP_CO2ER_pore.py:792),
which is threaded and not in this image."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "band_lu_omp.c")
LIB = os.path.join(HERE, "libband_lu_omp.so")
NF = 9


def build(force=False):
    """gcc -fopenmp; AVX2+FMA only (the library travels from the build container to the GPU box, so no -march=native)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-mavx2", "-mfma", SRC, "-lm", "-o", LIB], check=True)


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # waiting threads sleep: under a CFS quota spinning ones eat the period
        lib = ctypes.CDLL(LIB)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.band_lu_solve.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int]
        lib.band_lu_solve.restype = ctypes.c_int
        lib.band_lu_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def block_band(A, pos, nf=NF):
    """CSR/COO matrix in file dof order -> (n, b, blk): node-block band in the order pos (file vertex -> band position)."""
    assert nf == NF
    C = A.tocoo()
    n = pos.shape[0]
    bi, bj = pos[C.row // nf], pos[C.col // nf]
    b = int(np.abs(bi - bj).max())
    W = 2 * b + 1
    blk = np.zeros(n * W * nf * nf)
    idx = ((bi * W + (bj - bi + b)) * nf + C.row % nf) * nf + C.col % nf
    np.add.at(blk, idx, C.data)
    return n, b, blk


def solve(A, rhs, pos, threads=0, nf=NF):
    """x with A x = rhs (file dof order in and out).  Returns (x, seconds of the factor+solve call, half-bandwidth in blocks)."""
    import time
    lib = load()
    n, b, blk = block_band(A, pos, nf)
    dof_new = (pos[:, None] * nf + np.arange(nf)[None, :]).ravel()
    y = np.empty(n * nf)
    y[dof_new] = rhs
    dp = ctypes.POINTER(ctypes.c_double)
    t0 = time.perf_counter()
    rc = lib.band_lu_solve(n, b, blk.ctypes.data_as(dp), y.ctypes.data_as(dp), int(threads))
    dt = time.perf_counter() - t0
    if rc:
        raise RuntimeError("band_lu_solve: singular diagonal block %d" % (rc - 1))
    return y[dof_new], dt, b
