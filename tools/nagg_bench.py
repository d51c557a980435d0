import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd.pore3d import PoreRun
for na in (15, 12, 10, 8, 6):
    run = PoreRun(num_steps=30, concentration_elec=0.5, L=50e-9, R=5e-9, device_kwargs={"n_aggregates": na})
    run.step(verbose=False)
    t0 = time.perf_counter()
    for _ in range(29): run.step(verbose=False)
    dt = time.perf_counter() - t0
    its = sum(run.newton_its[1:])
    print("nagg %2d (%d): newton %d krylov %d  %.1f its/s  coarse_a %.2f coarse_b %.2f bicg_a %.2f bicg_b %.2f" % (na, run.sys.dev.n_aggregates, its, run.sys.krylov_iterations, its / dt,
          run.sys.dev.time_kernel(6, 100), run.sys.dev.time_kernel(7, 100), run.sys.dev.time_kernel(4, 100), run.sys.dev.time_kernel(5, 100)), flush=True)
    run.sys.close()
