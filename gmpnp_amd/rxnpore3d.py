"""3D reaction-diffusion driver for the cylindrical pore on the MI355X backend — same CLI flags, YAML/XML inputs and
output layout as reference 3D/rxn_diff_CO2ER_pore.py (``solveEDL`` :95-786, CLI :788-942).

The reference script is the pore model WITHOUT the potential, the migration terms, the steric terms and the cation
(seven species: H, OH, HCO3, CO32, CO2, CO, H2, :363), WITH the wall / pore-exit flux terms in F (:483-511) and with the
cation concentration of the Sechenov correction estimated from electroneutrality (:564-568).  Everything before the
FEniCS part is line for line what 3D/MPNP_CO2ER_pore.py computes (same effective diffusivities, scaling, boundary
markers, time step), so this driver takes ``params.pore_parameters`` and switches the model: valences 0, steric off,
no cation exit flux.  The backend still carries 8 species + potential: the cation is an inert placeholder (stays 1),
the potential is pinned to 0 by its Dirichlet sets on S1, S2, S3 (``voltage_multiplier`` = 0).

Differences, all explicit: ``$GMPNP_UTILITIES`` / ``$GMPNP_OUT`` roots; ``--num_steps`` (not in the
reference) shortens the 1000-step loop."""
from __future__ import annotations

import argparse
import copy
import json
import os
from datetime import datetime

import numpy as np

from .mesh import read_dolfin_xml, resolve_mesh_path
from .params import pore_parameters, utilities_dir
from .pore3d import SOLVER_PARAMETERS, output_root, scale_conc_time
from .problem import pore_dirichlet, pore_problem
from .solver import GMPNPSystem, column_medians
from .vtk import write_pvd

SOLVED = ["H", "OH", "HCO3", "CO32", "CO2", "CO", "H2"]  # the reference's MixedElement([P3]*7), :363


def rxn_pore_parameters(**kwargs):
    """``pore_parameters`` of the MPNP pore script with the reaction-diffusion switches (reference :115-480)."""
    assert "voltage_multiplier" not in kwargs, "the reaction-diffusion script has no voltage"
    pp = pore_parameters(voltage_multiplier=0.0, as_published=False, **kwargs)  # flux terms ARE in this script's F
    m = copy.deepcopy(pp.model)
    ns = len(pp.species)
    icat = pp.species.index(pp.cation)
    m.z = np.zeros(ns)                  # no migration; the Poisson row decouples
    m.a = np.zeros(ns)
    m.steric = False
    m.q, m.eps0, m.epsc = 0.0, 1.0, np.zeros(ns)
    m.exit_kappa = np.array(m.exit_kappa, dtype=float)
    m.exit_kappa[icat] = 0.0            # no J_pore_exit_cat: the cation is not a solved species here
    assert not np.asarray(m.rc1)[icat].any() and not np.asarray(m.rc2)[icat].any() and m.wall_flux[icat] == 0.0
    pp.model = m
    pp.voltage_scaled = 0.0
    return pp


class RxnPoreRun:
    """State of one run; ``step()`` is one pass of the reference's time loop body (:523-598)."""

    def __init__(self, num_steps=None, device_kwargs=None, solver_parameters=None, **kwargs):
        self.kwargs = kwargs
        self.pp = rxn_pore_parameters(**kwargs)
        self.mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), self.pp.mesh_name))
        self.problem, self.bnd = pore_problem(self.pp, self.mesh)
        self.sys = GMPNPSystem(self.problem, **(device_kwargs or {}))
        self.solver_parameters = solver_parameters or SOLVER_PARAMETERS  # :531-538 = the MPNP pore script's dict
        self.tot_num_steps = self.pp.tot_num_steps if num_steps is None else int(num_steps)
        nv = self.mesh.num_vertices
        self.sys.initialise([1.0] * 8 + [0.0])
        self.history = [np.ones((nv, 7))]
        self.CO2_min = None
        self.co2_bc = None
        self.n, self.t = 0, 0.0
        self.newton_its = []

    def step(self, verbose=True):
        pp = self.pp
        self.t += pp.dt
        st = self.sys.solve(self.solver_parameters)
        vals = self.sys.vertex_values()
        b, cat = pp.bulk_conc, pp.cation
        med = dict(zip(SOLVED[:4], (float(m) for m in column_medians(vals, range(4)))))
        # assuming electroneutrality to estimate the concentration of cations (:564-568)
        conc_cat = med["HCO3"] * b["HCO3"] + 2 * med["CO32"] * b["CO32"] + med["OH"] * b["OH"] - med["H"] * b["H"]
        co2 = pp.sechenov_co2_scaled(med["OH"], med["HCO3"], med["CO32"], conc_cat / b[cat])
        self.co2_bc = co2
        self.sys.set_bcs(*pore_dirichlet(pp, self.bnd, co2))  # bc1 rebuilt (:577-580); the potential pins ride along
        self.history.append(vals[:, :7].copy())
        self.CO2_min = float(np.amin(vals[:, 4]))
        self.sys.assign_previous()
        self.newton_its.append(st["iterations"])
        if verbose:
            print(self.CO2_min)
            print(datetime.now().strftime("%y-%m-%d-%H-%M-%S"))
            print(self.n)
        self.n += 1
        return st

    def run(self, verbose=True):
        for _ in range(self.n, self.tot_num_steps):
            self.step(verbose)
        return self

    def write_outputs(self, stamp=None):
        """solution_*.pvd, arrays_unscaled.npz, arrays_scaled.npz, metadata.json as in :600-786."""
        pp, mesh, k = self.pp, self.mesh, self.kwargs
        stamp = stamp or datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        end_time = datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        L, R = pp.L, pp.R
        identifier = ("L_" + str(int(L * 1e+9)) + "_R_" + str(int(R * 1e+9)) + "_P_g_" + str(k.get("press_gas", 1.0))
                      + "_D_eff_" + str(k.get("pore_geom_multiplier", 1.0)) + "_Re_"
                      + str(k.get("electrolyte_flow_geom_multiplier", 1.0)) + "_rough_" + str(k.get("roughness_factor", 150.0)))
        newpath = os.path.join(output_root(), stamp + "_experiment", identifier) + "/"
        os.makedirs(newpath, exist_ok=True)
        hist = np.stack(self.history)  # (steps+1, nv, 7)
        H = {nme: hist[:, :, i] for i, nme in enumerate(SOLVED)}
        last = hist[-1]
        for fname, col in (("CO", 5), ("H2", 6), ("CO2", 4), ("OH", 1), ("H", 0), ("HCO3", 2), ("CO32", 3)):
            write_pvd(os.path.join(newpath, "solution_" + fname + ".pvd"), mesh.coords, mesh.cells, last[:, col], "f_" + fname)
        grads = {nme: self.sys.project_gradient(last[:, i]).T.ravel() for i, nme in enumerate(SOLVED)}
        tau_array = np.linspace(0, pp.T, self.tot_num_steps)
        np.savez(newpath + "arrays_unscaled.npz", coor=mesh.coords, tau=tau_array, **H,
                 **{nme + "_grad": grads[nme] for nme in SOLVED})
        sc = {nme: scale_conc_time(species=nme, C=H[nme], grad_c=grads[nme], bulk_conc=pp.bulk_conc, tau=tau_array,
                                   diff_coeff_eff=pp.diff_coeff_eff, L=L) for nme in SOLVED}
        c = {nme: sc[nme][0] for nme in SOLVED}
        out = {"coor_scaled": mesh.coords * L, "c_cat": c["HCO3"] + 2 * c["CO32"] + c["OH"] - c["H"]}
        for nme in SOLVED:
            out["t_" + nme], out["c_" + nme], out[nme + "_grad"] = sc[nme][1], sc[nme][0], sc[nme][2]
        np.savez(newpath + "arrays_scaled.npz", **out)
        meta = {"concentration_elec": k.get("concentration_elec", 1.0), "cation": pp.cation, "H2_FE": k.get("H2_FE", 0.05),
                "L": L, "R": R, "time_step": pp.time_step, "total_sim_time": pp.total_sim_time,
                "porosity": k.get("porosity_eff", 0.5), "tortuosity": k.get("tortuosity_eff", 1.5),
                "constrictivity": k.get("constrictivity_eff", 0.9), "y_CO2": k.get("y_CO2", 0.95),
                "press_gas": k.get("press_gas", 1.0), "pore_geom_multiplier": k.get("pore_geom_multiplier", 1.0),
                "electrolyte_flow_geom_multiplier": k.get("electrolyte_flow_geom_multiplier", 1.0), "end_time": end_time,
                "eq_conc_CO": pp.eq_conc_CO, "eq_conc_H2": pp.eq_conc_H2, "current_planar": pp.current_planar,
                "CO2_min": self.CO2_min,
                # additions of this backend (new keys only)
                "newton_iterations": int(sum(self.newton_its)), "krylov_iterations": int(self.sys.krylov_iterations),
                "num_steps_run": int(self.n)}
        with open(newpath + "metadata.json", "w") as fh:
            fh.write(json.dumps(meta, indent=0))
        return newpath


def solveEDL(concentration_elec=1.0, H2_FE=0.05, current_rough=3000.0, L=100.0e-9, cation="K", R=5.0e-9, press_gas=1.0,
             pore_geom_multiplier=1.0, porosity_eff=0.5, tortuosity_eff=1.5, constrictivity_eff=0.9,
             params_file="parameters_pore", y_CO2=0.95, electrolyte_flow_geom_multiplier=1.0, roughness_factor=150.0,
             num_steps=None, verbose=True):
    """Same keyword surface as the reference's ``solveEDL`` (:95-110); returns the output directory."""
    run = RxnPoreRun(num_steps=num_steps, concentration_elec=concentration_elec, H2_FE=H2_FE, current_rough=current_rough,
                     L=L, cation=cation, R=R, press_gas=press_gas, pore_geom_multiplier=pore_geom_multiplier,
                     porosity_eff=porosity_eff, tortuosity_eff=tortuosity_eff, constrictivity_eff=constrictivity_eff,
                     params_file=params_file, y_CO2=y_CO2, electrolyte_flow_geom_multiplier=electrolyte_flow_geom_multiplier,
                     roughness_factor=roughness_factor)
    try:
        run.run(verbose=verbose)
        return run.write_outputs()
    finally:
        run.sys.close()


def build_parser():
    from .pore3d import build_parser as pore_parser
    p = pore_parser()  # same flags as the MPNP pore script ...
    for act in list(p._actions):
        if "--voltage_multiplier" in act.option_strings or "--as_published" in act.option_strings:
            p._remove_action(act)  # ... minus the voltage (reference :788-942 has none)
            for s in act.option_strings:
                p._option_string_actions.pop(s, None)
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    return solveEDL(concentration_elec=a.concentration_elec, H2_FE=a.H2_FE, current_rough=a.current_rough, L=a.L, cation=a.cation,
                    R=a.R, press_gas=a.press_gas, pore_geom_multiplier=a.pore_geom_multiplier, porosity_eff=a.porosity_eff,
                    tortuosity_eff=a.tortuosity_eff, constrictivity_eff=a.constrictivity_eff, params_file=a.params_file,
                    y_CO2=a.y_CO2, electrolyte_flow_geom_multiplier=a.electrolyte_flow_geom_multiplier,
                    roughness_factor=a.roughness_factor, num_steps=a.num_steps)
