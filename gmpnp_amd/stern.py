"""Stern-layer post-processor of the 1D runs (reference 1D/Stern_CO2ER.py; plain SciPy on the host — this
is what the reference is too: two coupled ODEs over 0.4 nm, nothing for a GPU).

Takes the potential, the electric field and the relative permittivity at the outer Helmholtz plane (OHP) of a GMPNP run
(``metadata.json`` keys ``field_OHP`` / ``eps_rel_OHP``, 1D:975-976) and integrates the ion-free Poisson equation across the Stern layer
(4 Angstrom, 1D/Stern_CO2ER.py:59) to the electrode surface: model ``BDM`` (permittivity varying linearly between the OHP value and 6
at the surface, Stern_CO2ER.py:82-103) or ``Stern_linear`` (constant field, :138-147).  Outputs as the reference writes them
(:101-110,155-163): ``stern_unscaled_BDM<V>.npz`` (arr_0 = the odeint solution), ``stern_scaled_BDM<V>.npz`` (arr_0 x [nm], arr_1
potential [V], arr_2 field [V/nm]) or ``stern_scaled_linear<V>.npz``, ``metadata.txt`` (:31-41), ``V_x.png`` / ``field_x.png`` when
matplotlib is there — under ``$GMPNP_OUT/Stern/<stamp>_experiment/voltage_scaled_OHP<V>/`` (the reference: an absolute path, Q10).

Reference behaviour kept on purpose (``as_published=True``, the default; ``--as_intended`` undoes both):
  S1  ``odeint(BDM, y0, x, args=(eps_rel_OHP, eps_rel_surface, L_stern))`` hands the two permittivities to a function whose parameters
      are declared in the OTHER order (:82,97): inside, "eps_rel_surface" is the OHP value and "eps_rel_OHP" is 6 — the profile runs
      from 6 at the OHP to the OHP value at the surface.
  S2  the state is (potential [V], field [V/nm]) and the independent variable is in METRES (:92-95): d(potential)/dx = field picks up
      1e-9 of the intended drop — the "voltage at the electrode" of metadata.txt is the OHP voltage to nine digits.
  S3  after defining everything ``main`` loops over the five recorded (voltage, field, permittivity) triples (:66-68,177-178) whatever
      the command line says; the flags only set defaults nobody uses.  Here: the same loop without arguments, ONE case when
      ``--field_OHP`` / ``--eps_rel_OHP`` / ``--from_run`` are given.
"""
from __future__ import annotations

import argparse
import json
import os
from datetime import datetime

import numpy as np

from .params import _load_yaml, utilities_dir

L_STERN = 4.0e-10          # [m] Stern_CO2ER.py:59
EPS_REL_SURFACE = 6.0      # rigid water at the catalyst surface, :79
# field_OHP [V/nm] and eps_rel_OHP of the 1D MPNP runs (K+, 0.1 M KHCO3) by voltage multiplier — 1D/Stern_CO2ER.py:66-68; the GPU
# solver reproduces them to <= 8e-11 (tests/test_gpu_parity.py::test_staged_schedule_reproduces_the_recorded_digits)
RECORDED = {-2.5: {"E": -0.08032108300135771, "eps": 74.56149297894756}, -5.0: {"E": -0.2524415478848975, "eps": 57.64572780716129},
            -7.5: {"E": -0.4612956299192668, "eps": 50.16243860179017}, -10.0: {"E": -0.6149631587776277, "eps": 49.311548142969336},
            -12.5: {"E": -0.7310301485096051, "eps": 49.2556833480052}}


def thermal_voltage(params_file="parameters"):
    nat = _load_yaml(os.path.join(utilities_dir(), params_file + ".yaml"))["nat_const"]
    return nat["k_B"] * nat["T"] / nat["e_0"]


def stern_grid(dx, xmax):
    """The reference's grid: ``np.linspace(0, xmax, abs(int(xmax / dx)))`` (:90-92,141-143) — the point count is what int() makes
    of the quotient, reproduced as written."""
    return np.linspace(0, xmax, abs(int(xmax / dx)))


def bdm_rhs(Y, x, eps_at_x0, eps_at_surface, L):
    """d(potential)/dx = field, d(field)/dx = -field * eps'(x) / eps(x) for eps linear from eps_at_x0 (x = 0) to eps_at_surface (x = -L)
    (:82-87, written there as (a - b) / (x (a - b) + a L) with a = eps_at_x0, b = eps_at_surface)."""
    return [Y[1], -Y[1] * ((eps_at_x0 - eps_at_surface) / (x * (eps_at_x0 - eps_at_surface) + eps_at_x0 * L))]


def bdm_closed_form(x, v0, f0, eps_at_x0, eps_at_surface, L):
    """The same ODE solved by hand (what the tests hold the integrator against): field = f0 b / (a x + b), potential = v0 +
    f0 (b / a) ln((a x + b) / b) with a = eps_at_x0 - eps_at_surface, b = eps_at_x0 L."""
    a, b = eps_at_x0 - eps_at_surface, eps_at_x0 * L
    return v0 + f0 * (b / a) * np.log((a * x + b) / b), f0 * b / (a * x + b)


def stern(voltage_scaled_OHP, field_OHP, eps_rel_OHP, model="BDM", as_published=True, params_file="parameters"):
    """One case; returns a dict with the arrays and the two surface values the reference writes to metadata.txt."""
    from scipy.integrate import odeint
    voltage_OHP = voltage_scaled_OHP * thermal_voltage(params_file)
    if model == "BDM":
        x = stern_grid(1.0e-11, -L_STERN)                       # metres, going backwards from the OHP (:89-92)
        if as_published:                                          # S1 + S2
            eps0, eps1, f0, xs = EPS_REL_SURFACE, eps_rel_OHP, -field_OHP, x
        else:                                                     # eps_rel_OHP at the OHP, 6 at the surface; x in nm like the field
            eps0, eps1, f0, xs = eps_rel_OHP, EPS_REL_SURFACE, -field_OHP, x * 1.0e9
        Lx = L_STERN if as_published else L_STERN * 1.0e9
        sol = odeint(bdm_rhs, [voltage_OHP, f0], xs, args=(eps0, eps1, Lx))
        pot, field = sol[:, 0], sol[:, 1] * -1
        return {"model": model, "voltage_OHP": voltage_OHP, "x_nm": x * 1.0e9, "potential": pot, "field": field, "sol": sol,
                "voltage_electrode": float(pot[-1]), "field_surf": float(field[-1])}
    if model == "Stern_linear":
        x = stern_grid(1.0e-2, -L_STERN * 1.0e9)               # nanometres (:141-143)
        pot = -field_OHP * x + voltage_OHP
        return {"model": model, "voltage_OHP": voltage_OHP, "x_nm": x, "potential": pot, "field": np.full_like(x, field_OHP), "sol": None,
                "voltage_electrode": float(voltage_OHP - (-field_OHP * (L_STERN * 1.0e9))), "field_surf": float(field_OHP)}
    raise ValueError("model %r: BDM or Stern_linear" % model)   # (the reference silently does nothing)


def write_outputs(res, voltage_scaled_OHP, field_OHP, eps_rel_OHP, stamp=None, plots=True):
    stamp = stamp or datetime.now().strftime("%y-%m-%d-%H-%M-%S")
    newpath = os.path.join(os.environ.get("GMPNP_OUT", os.path.join(os.getcwd(), "out")), "Stern", stamp + "_experiment",
                           "voltage_scaled_OHP" + str(voltage_scaled_OHP))
    os.makedirs(newpath, exist_ok=True)
    if res["model"] == "BDM":
        np.savez(os.path.join(newpath, "stern_unscaled_BDM%s.npz" % voltage_scaled_OHP), res["sol"])
        np.savez(os.path.join(newpath, "stern_scaled_BDM%s.npz" % voltage_scaled_OHP), res["x_nm"], res["potential"], res["field"])
    else:
        np.savez(os.path.join(newpath, "stern_scaled_linear%s.npz" % voltage_scaled_OHP), res["x_nm"], res["potential"])
    with open(os.path.join(newpath, "metadata.txt"), "w") as f:   # the seven lines of :33-39, text and units as there
        f.write("model=" + res["model"] + "\n")
        f.write("voltage_OHP=" + str(res["voltage_OHP"]) + "V\n")
        f.write("field_OHP=" + str(field_OHP) + "V/nm\n")
        f.write(f"Relative permittivity at the OHP is {eps_rel_OHP} \n")
        f.write(f"voltage at the electrode is {res['voltage_electrode']} \n")
        f.write(f"Electric field at the surface is {res['field_surf']} m\n")
        f.write(f"Stern length is {L_STERN} m\n")
    if plots:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            return newpath
        for name, y, label in (("V_x.png", res["potential"], "potential in V"), ("field_x.png", res["field"], "electric field in V/nm")):
            if name == "field_x.png" and res["model"] != "BDM":
                continue
            plt.figure()
            plt.plot(res["x_nm"], y)
            plt.xlabel("distance (nm)"); plt.ylabel(label); plt.title("voltage_multiplier: " + str(voltage_scaled_OHP))
            plt.xticks(rotation=90); plt.tight_layout()
            plt.savefig(os.path.join(newpath, name)); plt.close()
    return newpath


def ohp_of_run(run_dir):
    """(voltage multiplier, field_OHP [V/nm], eps_rel_OHP) of a 1D run directory written by ``1D/MPNP_CO2ER_EDL.py`` (its metadata.json)."""
    with open(os.path.join(run_dir, "metadata.json")) as fh:
        md = json.load(fh)
    return float(md["voltage_multiplier"]), float(md["field_OHP"]), float(md["eps_rel_OHP"])


def main(argv=None):
    p = argparse.ArgumentParser(description="experiment parameters")
    p.add_argument("--voltage_scaled_OHP", metavar="voltage multiplier", required=False, help="float val", default=-2.5, type=float)
    p.add_argument("--model", metavar="model_type", required=False, help="str, BDM/Stern_linear", default="BDM", type=str)
    p.add_argument("--field_OHP", metavar="electric field at the OHP", required=False, help="float val, -0.5", default=None, type=float)
    p.add_argument("--eps_rel_OHP", metavar="relative permittivity at the OHP", required=False, help="float, 80.0", default=None, type=float)
    p.add_argument("--from_run", default=None, help="output directory of a 1D run: voltage, field_OHP and eps_rel_OHP from its metadata.json")
    p.add_argument("--as_intended", action="store_true", help="permittivities in their declared roles, x in nm (see S1, S2 in the module text)")
    p.add_argument("--no_plots", action="store_true")
    a = p.parse_args(argv)
    if a.from_run:
        cases = [ohp_of_run(a.from_run)]
    elif a.field_OHP is not None or a.eps_rel_OHP is not None:
        cases = [(a.voltage_scaled_OHP, -0.5 if a.field_OHP is None else a.field_OHP, 80.0 if a.eps_rel_OHP is None else a.eps_rel_OHP)]
    else:
        cases = [(v, d["E"], d["eps"]) for v, d in RECORDED.items()]      # S3
    stamp = datetime.now().strftime("%y-%m-%d-%H-%M-%S")
    out = []
    for v, e, eps in cases:
        res = stern(v, e, eps, model=a.model, as_published=not a.as_intended)
        path = write_outputs(res, v, e, eps, stamp=stamp, plots=not a.no_plots)
        print("voltage multiplier %6.2f: voltage at the OHP %.6f V, at the electrode %.6f V, field at the surface %.6f V/nm -> %s"
              % (v, res["voltage_OHP"], res["voltage_electrode"], res["field_surf"], path))
        out.append(path)
    return out


if __name__ == "__main__":
    main()
