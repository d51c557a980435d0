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
