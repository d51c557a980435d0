"""Parameter sweep over pore radii and voltages, one independent pore problem per GPU at a time (BASELINE configs[4]:
"L_50_R_{1,2,2.5,4,5,7.5,10} x voltage ramp, 8 x MI355X one partition per GPU").

The reference runs such sweeps as a shell loop over ``python MPNP_CO2ER_pore.py --R=... --voltage_multiplier=...``
(3D/MPNP_CO2ER_pore.py:1088-1253 is the per-run CLI); every run is independent, so the multi-GPU mapping has no
data-path collective: rank r takes jobs r, r+world, r+2*world, ... and the per-job summaries are gathered on rank 0
at the end (one ``all_gather_object``).

  python -m gmpnp_amd.sweep --num_steps 20                                   # 1 GPU, all 35 jobs in turn
  python -m torch.distributed.run --nproc-per-node 8 -m gmpnp_amd.sweep      # one rank per GPU (RCCL only for the gather)
"""
from __future__ import annotations

import argparse
import json
import os
import time

RADII_NM = (1, 2, 2.5, 4, 5, 7.5, 10)                  # the seven L_50_R_* meshes of the reference's utilities/
VOLTAGES = (-1.0, -2.5, -5.0, -7.5, -10.0)             # the ramp recorded in 1D/Stern_CO2ER.py:66-68


def jobs(radii=RADII_NM, voltages=VOLTAGES):
    return [(r, v) for r in radii for v in voltages]


def my_jobs(all_jobs, rank, world):
    """Round-robin deal; the radii vary fastest across ranks so that every GPU sees small and large meshes."""
    return [j for k, j in enumerate(all_jobs) if k % world == rank]


def ramp_value(target, n, ramp_steps, start=-1.0):
    """Wall potential in force during time step n of a ramped run: linear from ``start`` to ``target`` over the
    first ``ramp_steps`` steps, ``target`` afterwards (ramp_steps = 0: the reference's behaviour, target from step 0)."""
    if ramp_steps <= 0 or abs(target) <= abs(start):
        return float(target)
    return float(start + (target - start) * min(n, ramp_steps) / ramp_steps)


def run_job(radius_nm, voltage, num_steps, concentration_elec=0.5, device_id=0, write=False, as_published=False, ramp_steps=0,
            one_stream=False):
    """One pore run of ``num_steps`` time steps; returns a small summary dict (never raises for a diverged Newton).

    ``ramp_steps`` > 0 is a continuation the reference does not have: the wall potential Dirichlet value (bc3 of
    3D:460-467) moves from -1 to the target over that many time steps, every step starting from the previous state;
    Newton from the zero state diverges for |voltage_multiplier| >= 5 (as published) or > 1 (with the wall fluxes),
    in the CPU oracle exactly as on the GPU."""
    from .pore3d import PoreRun
    from .problem import pore_dirichlet
    t0 = time.perf_counter()
    out = {"R_nm": radius_nm, "voltage_multiplier": voltage, "steps_requested": num_steps, "ramp_steps": ramp_steps}
    run = None
    try:
        run = PoreRun(num_steps=num_steps, concentration_elec=concentration_elec, L=50e-9, R=radius_nm * 1e-9,
                      voltage_multiplier=ramp_value(voltage, 0, ramp_steps), as_published=as_published,
                      # several runs in flight on one GPU: every handle keeps to ONE stream (coarse rebuild and warm-start test in
                      # the main stream): with side streams K handles are 2K streams on the process's four hardware queues
                      device_kwargs=dict({"device_id": device_id}, **({"coarse_refresh": 3, "warm_in_stream": 1} if one_stream else {})))
        out.update(n_vertices=run.mesh.num_vertices, n_dofs=run.problem.ndof)
        for n in range(num_steps):
            run.step(verbose=False)
            v_next = ramp_value(voltage, n + 1, ramp_steps)
            if v_next != run.pp.voltage_scaled:  # the Dirichlet set of the NEXT step (3D:835-838 rebuilds bc4 the same way)
                run.pp.voltage_scaled = v_next
                run.sys.set_bcs(*pore_dirichlet(run.pp, run.bnd, run.co2_bc))
        out.update(wall_potential=float(run.pp.voltage_scaled))
        out.update(status="ok")
        if write:
            out["directory"] = run.write_outputs()
    except RuntimeError as e:  # DOLFIN's behaviour: a non-converged Newton aborts THAT run (3D:789-799)
        out.update(status="newton_failed", error=str(e)[:200])
    if run is not None:
        out.update(steps_done=run.n, newton_iterations=int(sum(run.newton_its)), krylov_iterations=int(run.sys.krylov_iterations),
                   CO2_min=None if run.CO2_min is None else float(run.CO2_min))
        run.sys.close()
    out["seconds"] = time.perf_counter() - t0
    return out


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("--num_steps", type=int, default=20)
    p.add_argument("--concentration_elec", type=float, default=0.5)
    p.add_argument("--radii", type=float, nargs="*", default=list(RADII_NM))
    p.add_argument("--voltages", type=float, nargs="*", default=list(VOLTAGES))
    p.add_argument("--as_published", action="store_true",
                   help="the weak form exactly as published (no ds(2)/ds(3) flux terms, SURVEY Q1); the intended form with "
                        "the wall fluxes only converges from the zero initial guess at voltage_multiplier = -1")
    p.add_argument("--ramp_steps", type=int, default=0,
                   help="move the wall potential from -1 to the target over this many time steps (continuation; 0 = the "
                        "reference's behaviour, the target applies from step 0)")
    p.add_argument("--jobs_per_gpu", type=int, default=1, help="independent runs kept in flight on each GPU (separate streams; 3 gives 1.56x the throughput of 1)")
    p.add_argument("--write", action="store_true", help="write the reference's output files of every run under $GMPNP_OUT")
    p.add_argument("--backend", default=None, help="torch.distributed backend for the final gather (default: nccl)")
    a = p.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = a.backend or "nccl"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    radii = [int(r) if float(r).is_integer() else r for r in a.radii]
    mine = my_jobs(jobs(radii, a.voltages), rank, world)
    t0 = time.perf_counter()
    def one(job):
        return run_job(job[0], job[1], a.num_steps, a.concentration_elec, device_id=local, write=a.write,
                       as_published=a.as_published, ramp_steps=a.ramp_steps, one_stream=a.jobs_per_gpu > 1)

    if a.jobs_per_gpu > 1:
        # A 3.7k-vertex problem is launch-latency bound: independent problems on separate HIP streams
        # overlap on one GPU.  Measured (tools/concurrent_runs_probe.py, profiles/r03/concurrent_runs*.json): 2 / 3 / 4 runs from
        # threads of one process = 1.40 / 1.56 / 1.38 x the throughput of one (3 is the sweet spot: a launch needs 537 of the
        # GPU's 768 workgroup slots, so launches of different problems overlap by their tails only).  One host thread per job;
        # ctypes releases the GIL inside the library.
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=a.jobs_per_gpu) as pool:
            res = list(pool.map(one, mine))
    else:
        res = [one(j) for j in mine]
    dt = time.perf_counter() - t0
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "seconds": dt, "results": res})
        dist.barrier()
        dist.destroy_process_group()
    else:
        gathered = [{"rank": 0, "seconds": dt, "results": res}]
    if rank == 0:
        allres = [r for g in gathered for r in g["results"]]
        wall = max(g["seconds"] for g in gathered)
        its = sum(r.get("newton_iterations", 0) for r in allres)
        print(json.dumps({"jobs": len(allres), "ok": sum(r["status"] == "ok" for r in allres), "world_size": world,
                          "wall_seconds": wall, "newton_iterations": its, "newton_iterations_per_sec": its / wall if wall else None,
                          "results": allres}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
