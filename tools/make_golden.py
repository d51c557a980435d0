#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ with the in-repo CPU oracle.

The reference itself (FEniCS 2019.1.0) is not installed and not installable here, so these vectors are
ORACLE outputs, not reference outputs (see oracle/gmpnp_oracle.py header: parity of the steric quadrature is
unpinned).  They freeze the oracle and give the GPU tests full-size targets that take the oracle minutes.

  python tools/make_golden.py [elements] [pore10] [pore50] [edl1] [edl50]

edl1 = 1 um mesh, Cs, V = -5 (the BASELINE config-0 voltage -10 makes the undamped Newton of the first step
diverge in the oracle); edl50 = the reference README example (50 um, Cs, V = -10).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gmpnp_oracle as O
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, edl_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem, edl_problem
from gmpnp_amd.model import default_quadrature

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import EXTRA_PORE, EXTRA_EDL, EXTRA_RXN1D, EXTRA_RXN3D, RXN_NEWTON
what = set(sys.argv[1:]) or ({"elements", "pore10", "pore50", "edl1", "edl50"} | set(EXTRA_PORE) | set(EXTRA_EDL) | set(EXTRA_RXN1D) | set(EXTRA_RXN3D))


def pad_res(res):
    m = max(len(r) for r in res)
    return np.array([r + [np.nan] * (m - len(r)) for r in res])


if "elements" in what:
    rng = np.random.default_rng(0)
    pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
    X = rng.uniform(0, 0.05, (6, 4, 3)); U = np.concatenate([rng.uniform(.5, 1.5, (6, 4, 8)), rng.uniform(-1, 0, (6, 4, 1))], 2)
    Un = np.concatenate([rng.uniform(.5, 1.5, (6, 4, 8)), rng.uniform(-1, 0, (6, 4, 1))], 2)
    Fe, Je = O.element_residual_jacobian(pp.model, default_quadrature(3), X, U, Un)
    ep = edl_parameters(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    X1 = np.sort(rng.uniform(0, 1e-3, (6, 2, 1)), axis=1); U1 = np.concatenate([rng.uniform(.5, 1.5, (6, 2, 6)), rng.uniform(-1, 0, (6, 2, 1))], 2)
    Un1 = np.concatenate([rng.uniform(.5, 1.5, (6, 2, 6)), rng.uniform(-1, 0, (6, 2, 1))], 2)
    Fe1, Je1 = O.element_residual_jacobian(ep.model, default_quadrature(1), X1, U1, Un1)
    np.savez(os.path.join(G, "elements.npz"), X=X, U=U, Un=Un, Fe=Fe, Je=Je, X1=X1, U1=U1, Un1=Un1, Fe1=Fe1, Je1=Je1)
    print("elements done")

for key, (L, R, steps) in {"pore10": (10e-9, 5e-9, 3), "pore50": (50e-9, 5e-9, 2)}.items():
    if key not in what:
        continue
    t = time.time()
    pp = pore_parameters(concentration_elec=0.5, L=L, R=R)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
    prob, bnd = pore_problem(pp, mesh)
    out = O.pore_time_loop(pp, prob, bnd, steps, verbose=True)
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]), co2_bc=np.array(out["co2_bc"]),
                        args=np.array([0.5, L, R]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])

for key, kw, steps in (("edl1", dict(L_n=1e-6, cation="Cs", voltage_multiplier=-5.0), 5),
                       ("edl50", dict(cation="Cs", voltage_multiplier=-10.0), 3)):
    if key not in what:
        continue
    t = time.time()
    ep = edl_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), ep.mesh_name))
    prob = edl_problem(ep, mesh)
    out = O.edl_time_loop(ep, prob, steps, verbose=True)
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])

for key, (kw, steps) in EXTRA_PORE.items():
    if key not in what:
        continue
    t = time.time()
    pp = pore_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
    prob, bnd = pore_problem(pp, mesh)
    out = O.pore_time_loop(pp, prob, bnd, steps, verbose=True)
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]), co2_bc=np.array(out["co2_bc"]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])

for key, (kw, steps) in EXTRA_EDL.items():
    if key not in what:
        continue
    t = time.time()
    ep = edl_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), ep.mesh_name))
    prob = edl_problem(ep, mesh)
    out = O.edl_time_loop(ep, prob, steps, verbose=True,
                          stabilization=(kw.get("stabilization") == "Y" and kw.get("model") == "PNP"))
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])

for key, (kw, steps) in EXTRA_RXN1D.items():
    if key not in what:
        continue
    t = time.time()
    from gmpnp_amd.rxndiff1d import rxn_diff_parameters
    rp = rxn_diff_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), rp.mesh_name))
    prob = edl_problem(rp, mesh)
    out = O.edl_time_loop(rp, prob, steps, newton_kwargs=RXN_NEWTON, verbose=True)
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])

for key, (kw, steps) in EXTRA_RXN3D.items():
    if key not in what:
        continue
    t = time.time()
    from gmpnp_amd.rxnpore3d import rxn_pore_parameters
    pp = rxn_pore_parameters(**kw)
    mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
    prob, bnd = pore_problem(pp, mesh)
    out = O.pore_time_loop(pp, prob, bnd, steps, verbose=True, cation_from_electroneutrality=True)
    np.savez_compressed(os.path.join(G, key + "_steps.npz"), states=out["states"], newton_its=np.array(out["newton_its"]),
                        residuals=pad_res(out["residuals"]), co2_bc=np.array(out["co2_bc"]))
    print(key, "done in %.1fs" % (time.time() - t), out["newton_its"])
