"""How well does the previous time step's total update predict the next one? (decides whether Newton iteration 0 of a
step is worth warm-starting from it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd.pore3d import PoreRun
run = PoreRun(num_steps=14, concentration_elec=0.5, L=50e-9, R=5e-9)
run.run(verbose=False)
H = np.array(run.history)
D = H[1:] - H[:-1]
for n in range(1, len(D) - 1):
    a, b = D[n].ravel(), D[n + 1].ravel()
    rho = (a @ b) / (a @ a)
    print("step %2d: |D| %.3e  rho %.4f  |D_next - D|/|D_next| %.3f  |D_next - rho D|/|D_next| %.3f" % (n + 1, np.linalg.norm(b), rho, np.linalg.norm(b - a) / np.linalg.norm(b), np.linalg.norm(b - rho * a) / np.linalg.norm(b)))
run.sys.close()
