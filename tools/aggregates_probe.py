"""Slab aggregates of the two-level preconditioner on L_50_R_5: BiCGStab iterations and wall time of the 50-step bench window per count.
python tools/aggregates_probe.py 6 8 10 12 15"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.pore3d import PoreRun
for k in [int(a) for a in sys.argv[1:]] or [8]:
    run = PoreRun(num_steps=52, concentration_elec=0.5, L=50e-9, R=5e-9, device_kwargs={"n_aggregates": k})
    for _ in range(2): run.step(verbose=False)
    its0, kr0 = sum(run.newton_its), run.sys.krylov_iterations
    t0 = time.perf_counter()
    for _ in range(50): run.step(verbose=False)
    dt = time.perf_counter() - t0
    its, kr = sum(run.newton_its) - its0, run.sys.krylov_iterations - kr0
    print("aggregates %2d: %d Newton, %d BiCGStab (%.1f per solve), %.3f s, %.1f Newton its/s, %.2f us per BiCGStab iteration all in" % (k, its, kr, kr / its, dt, its / dt, 1e6 * dt / kr), flush=True)
    run.sys.close()
