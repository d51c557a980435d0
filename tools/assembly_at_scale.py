"""Assembly kernels on the twice-refined L_50_R_5 mesh (1.1 M cells): k_element (direct vs staged record stores), k_jac_gather,
k_res_gather — back-to-back launches, HIP events (gmpnp_time_kernel).  Bytes: element records 1,936 B per cell written by
k_element; k_jac_gather reads the 1,648-B Jacobian records and writes the SELL matrix.

    python tools/assembly_at_scale.py [refine=2] [out.json]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd import backend
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, _ = pore_problem(pp, mesh, refine=R)
nv, nc = prob.coords.shape[0], prob.cells.shape[0]
rng = np.random.default_rng(0)
u = np.concatenate([rng.uniform(.5, 1.5, (nv, 8)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
rows = {"refine": R, "n_vertices": nv, "n_cells": nc}
for mode, name in ((1, "direct"), (2, "staged")):
    with backend.DeviceSolver(prob, element_stores=mode) as dev:
        dev.set_state(u, u)
        dev.assemble(True)
        t_el = dev.time_kernel(1, 10)
        rec_bytes = nc * (206 + 36) * 8
        rows["k_element_" + name] = {"us": t_el, "record_bytes": rec_bytes, "TB_per_s": rec_bytes / t_el / 1e6}
        if mode == 2 or R == 0:
            t_g = dev.time_kernel(2, 10)
            nb = dev.n_blocks
            gb = nb * 81 * 8 + nc * 206 * 8
            rows["k_jac_gather"] = {"us": t_g, "matrix_bytes": nb * 81 * 8, "record_bytes_read_once": nc * 206 * 8, "TB_per_s_matrix_plus_records": gb / t_g / 1e6,
                                    "TB_per_s_algorithmic_matrix_only": nb * 81 * 8 / t_g / 1e6}
            rows["k_res_gather"] = {"us": dev.time_kernel(3, 10)}
print(json.dumps(rows, indent=1))
if len(sys.argv) > 2:
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
