"""Where a Newton iteration of the bench workload goes: device time of assembly / preconditioner set-up / Krylov
(hipEvents inside gmpnp_newton_solve) against the wall clock of the solve and of the whole time step (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.pore3d import PoreRun

run = PoreRun(num_steps=52, concentration_elec=0.5, L=50e-9, R=5e-9, device_kwargs={"phase_timing": 1})
for _ in range(2):
    run.step(verbose=False)
acc = dict(ms_assemble=0.0, ms_setup=0.0, ms_krylov=0.0, ms_total=0.0, iterations=0, krylov_iterations=0)
t0 = time.perf_counter()
for _ in range(50):
    st = run.step(verbose=False)
    for k in acc:
        acc[k] += st[k]
wall = (time.perf_counter() - t0) * 1e3
n = acc["iterations"]
print("50 steps: wall %.1f ms, inside gmpnp_newton_solve %.1f ms, Newton its %d, Krylov its %d" % (wall, acc["ms_total"], n, acc["krylov_iterations"]))
print("per Newton iteration: wall %.3f ms | solve %.3f | assembly (device) %.3f | set-up (device) %.3f | Krylov (device span) %.3f | solve - those = %.3f | outside the solve %.3f"
      % (wall / n, acc["ms_total"] / n, acc["ms_assemble"] / n, acc["ms_setup"] / n, acc["ms_krylov"] / n,
         (acc["ms_total"] - acc["ms_assemble"] - acc["ms_setup"] - acc["ms_krylov"]) / n, (wall - acc["ms_total"]) / n))
print("Krylov span per BiCGStab iteration: %.2f us" % (1e3 * acc["ms_krylov"] / acc["krylov_iterations"]))
run.sys.close()
