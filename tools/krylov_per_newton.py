"""BiCGStab iterations per Newton iteration over the bench window (first / second / third / later solves of a time step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.pore3d import PoreRun
run = PoreRun(num_steps=50, concentration_elec=0.5, L=50e-9, R=5e-9)
tot=0
for n in range(50):
    st = run.step(verbose=False)
    k = list(st.get('krylov_per_iteration', []))[:st['iterations']]
    tot += sum(k)
    if n < 12 or n % 10 == 0: print(n, st['iterations'], k)
print('total', tot)
run.sys.close()
