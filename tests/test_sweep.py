"""Parameter sweep driver (BASELINE configs[4]): job dealing and voltage continuation on the CPU, a two-job sweep on the GPU."""
import json

import pytest

from gmpnp_amd import sweep


def test_jobs_are_dealt_round_robin():
    all_jobs = sweep.jobs()
    assert len(all_jobs) == 35 and all_jobs[0] == (1, -1.0) and all_jobs[-1] == (10, -10.0)
    for world in (1, 2, 4, 8):
        dealt = [sweep.my_jobs(all_jobs, r, world) for r in range(world)]
        assert sorted(j for d in dealt for j in d) == sorted(all_jobs)          # every job exactly once
        assert max(len(d) for d in dealt) - min(len(d) for d in dealt) <= 1    # balanced
    assert sweep.my_jobs(all_jobs, 3, 8) == [all_jobs[k] for k in range(3, 35, 8)]


def test_ramp_value():
    assert sweep.ramp_value(-5.0, 0, 0) == -5.0                      # no ramp: the reference's behaviour
    assert sweep.ramp_value(-1.0, 7, 10) == -1.0                     # targets inside the start value are not ramped
    vals = [sweep.ramp_value(-5.0, n, 4) for n in range(7)]
    assert vals == [-1.0, -2.0, -3.0, -4.0, -5.0, -5.0, -5.0]


@pytest.mark.gpu
def test_two_job_sweep(gpu_lib, capsys):
    assert sweep.main(["--num_steps", "2", "--radii", "5", "--voltages", "-1", "-2.5", "--as_published"]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["jobs"] == 2 and out["ok"] == 2 and out["world_size"] == 1
    r0, r1 = out["results"]
    assert r0["n_vertices"] == 3679 and r0["steps_done"] == 2 and r0["newton_iterations"] >= 5
    assert r1["wall_potential"] == -2.5
    # Newton from the zero state diverges at -10 (in the CPU oracle too): reported per job, the sweep itself goes on
    assert sweep.main(["--num_steps", "1", "--radii", "5", "--voltages", "-10", "--as_published"]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["ok"] == 0 and out["results"][0]["status"] == "newton_failed"
