"""Long 3D run (default L_50_R_5, 0.5 M): Newton / Krylov counts along the way, wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd.pore3d import PoreRun
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
run = PoreRun(num_steps=n, concentration_elec=0.5, L=50e-9, R=5e-9)
t0 = time.perf_counter(); k0 = 0
for i in range(n):
    run.step(verbose=False)
    if (i + 1) % (n // 10) == 0:
        k = run.sys.krylov_iterations
        print("step %4d  newton its of the last %d steps: %s  krylov %d  CO2_min %.4f  t %.2fs" % (i + 1, n // 10, sum(run.newton_its[-(n // 10):]), k - k0, run.CO2_min, time.perf_counter() - t0), flush=True)
        k0 = k
its = sum(run.newton_its); dt = time.perf_counter() - t0
print("total: %d steps, %d Newton its, %d Krylov its, %.2f s -> %.1f its/s, finite %s" % (n, its, run.sys.krylov_iterations, dt, its / dt, bool(np.isfinite(run.history[-1]).all())))
run.sys.close()
