"""The parts of bench.py that do not need a GPU: how the line is put together around the partitioned phase (also by the
watchdog), and the all-cores banded CPU leg."""
import argparse
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _base(world):
    out = {"metric": "newton_iterations_per_sec", "value": 100.0, "ms_per_step": 10.0, "scaling": "weak", "n_gpus": world,
           "config": {"newton_iterations": 298.0, "krylov_iterations": 17000.0, "parallelism": "x"}, "roofline": {"frac": 0.39}}
    if world > 1:
        out["replicas"] = {"value": 100.0}
    return out


def test_partitioned_result_becomes_the_headline_only_when_it_has_a_value():
    a = argparse.Namespace(steps=50)
    good = {"value": 40.0, "ms_per_step": 25.0, "seconds": 1.25, "newton_iterations": 298.0, "krylov_iterations": 17100.0, "transport": "peer",
            "transport_checks": {"peer": {"selftest": "pass", "run": "pass"}}, "state_vs_single_gpu": 3e-12}
    out = bench.attach_partitioned(_base(4), a, 4, good)
    assert out["value"] == 40.0 and out["scaling"] == "strong" and out["replicas"]["value"] == 100.0
    assert out["partitioned"]["transport"] == "peer" and out["partitioned"]["transport_checks"]["peer"]["run"] == "pass"
    assert "4 z-slab mesh partitions" in out["config"]["parallelism"] and out["config"]["krylov_iterations"] == 17100.0
    # a failed / hung phase leaves the replica figure as the value and carries the reason (what the watchdog prints before exit 3)
    bad = {"error": "partitioned phase did not finish within 240 s", "watchdog": "hung in rccl: timed run; ...", "transport_checks": {"peer": {"selftest": "selftest: largest deviation 1.0e+06", "run": None}}}
    out = bench.attach_partitioned(_base(4), a, 4, bad)
    assert out["value"] == 100.0 and out["scaling"] == "weak" and out["partitioned"]["watchdog"].startswith("hung in rccl")
    # the base dict is not modified (the watchdog and the main path share it)
    base = _base(2)
    bench.attach_partitioned(base, a, 2, good)
    assert base["value"] == 100.0 and "partitioned" not in base
    # N = 1 rehearsal
    out = bench.attach_partitioned(_base(1), a, 1, good)
    assert out["value"] == 100.0 and out["partitioned_rehearsal"]["transport"] == "peer"


def test_banded_all_cores_leg_solves_the_same_system_as_superlu(boxpore):
    import gmpnp_oracle as O
    _, mesh, prob = boxpore
    nv = mesh.num_vertices
    u0 = np.zeros(prob.ndof)
    un = np.tile(np.r_[np.ones(8), 0.0], nv)
    leg = bench.banded_leg(O, prob, u0, un)
    assert leg["solution_vs_superlu"] < 1e-8 and leg["threads"] >= 1 and leg["lu_seconds"] > 0
    kl, ku = leg["half_bandwidth_scalars"]
    assert 9 <= kl < prob.ndof // 4 and 9 <= ku < prob.ndof // 4   # slab order: a band, not the whole matrix
    # the threaded node-block band LU (oracle/band_lu_omp.c) solves the same system
    omp = leg["block_band_openmp"]
    assert "error" not in omp, omp
    assert omp["solution_vs_superlu"] < 1e-8 and omp["lu_seconds"] > 0
    assert (omp["half_bandwidth_blocks"] + 1) * 9 > kl >= omp["half_bandwidth_blocks"] * 9 - 8


def test_block_band_lu_is_thread_count_independent_and_reports_singular_blocks():
    import band_lu
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, b, nf = 40, 6, 9
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in range(max(0, i - b), min(n, i + b + 1)):
            if i == j or rng.random() < 0.5:
                blk = rng.standard_normal((nf, nf)) + (12.0 * np.eye(nf) if i == j else 0.0)
                r, c = np.meshgrid(np.arange(nf), np.arange(nf), indexing="ij")
                rows.append((i * nf + r).ravel()); cols.append((j * nf + c).ravel()); vals.append(blk.ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * nf, n * nf))
    rhs = rng.standard_normal(n * nf)
    pos = rng.permutation(n)          # any vertex order: the band just gets wider
    x1, _, hb = band_lu.solve(A, rhs, pos, threads=1)
    x4, _, _ = band_lu.solve(A, rhs, pos, threads=4)
    assert np.array_equal(x1, x4)     # every block has one owner and one summation order, whatever the thread count
    assert np.linalg.norm(A @ x1 - rhs) < 1e-10 * np.linalg.norm(rhs) and hb <= n - 1
    Z = A.tolil(); Z[0:nf, :] = 0.0
    try:
        band_lu.solve(Z.tocsr(), rhs, np.arange(n))
        raise AssertionError("singular diagonal block went unnoticed")
    except RuntimeError as e:
        assert "singular diagonal block 0" in str(e)


def test_usable_cpus_is_within_the_machine():
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
