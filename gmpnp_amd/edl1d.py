"""1D electrical-double-layer GMPNP/PNP driver on the MI355X backend — same CLI flags, inputs and output layout as
reference 1D/MPNP_CO2ER_EDL.py (``solve_EDL`` 1D:66-989, CLI 1D:992-1118).

Kept quirks: ``--dry_run`` is ``type=bool`` so any value given on the command line is truthy; in
non-dry-run mode the form keeps the first time step while ``t`` advances by the second (Q2) — here: the model's
``inv_dt`` is never changed after the first stage; the run then ends with NameError for ``time_step`` as the
reference does (Q3).  ``--stabilization Y`` (1D:597-722): for MPNP the reference only prints a warning and solves the
unstabilised form, which is what happens here; for PNP the SUPG terms are added on the device (``gmpnp_set_supg``) with
the nodal parameters recomputed every step from the previous potential (``solver.supg_parameters``).  Paths: ``$GMPNP_UTILITIES`` / ``$GMPNP_OUT`` (Q10)."""
from __future__ import annotations

import argparse
import copy
import json
import math
import os
from datetime import datetime

import numpy as np

from .mesh import read_dolfin_xml, resolve_mesh_path
from .params import edl_parameters, utilities_dir
from .problem import edl_problem
from .solver import GMPNPSystem, supg_parameters

SOLVER_PARAMETERS = {  # reference 1D:357-364
    "nonlinear_solver": "newton",
    "newton_solver": {"maximum_iterations": 50, "relative_tolerance": 1.0e-4, "absolute_tolerance": 1.0e-4},
}


def scale(species="H", tau=None, C=None, initial_conc=None, diff_coeff=None, L_n=0.0, L_debye=0.0):
    """reference 1D:51-63"""
    t = (tau * L_debye * L_n) / diff_coeff[species]
    c = C * initial_conc[species]
    return t, c


def output_root():
    return os.environ.get("GMPNP_OUT", os.path.join(os.getcwd(), "out"))


class EDLRun:
    def __init__(self, num_steps=None, device_kwargs=None, solver_parameters=None, **kwargs):
        self.kwargs = kwargs
        self.ep = edl_parameters(**kwargs)
        ep = self.ep
        stab = kwargs.get("stabilization", "N") == "Y"
        self.supg = stab and ep.model_name == "PNP"        # reference 1D:687-722
        self.warn_stab = stab and ep.model_name != "PNP"   # "Warning:stabilization not implemented for MPNP!", 1D:724-727
        self.h_vertex = None
        self.mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), ep.mesh_name))
        self.problem = edl_problem(ep, self.mesh)
        self.model = copy.deepcopy(self.problem.model)
        self.sys = GMPNPSystem(self.problem, **(device_kwargs or {}))
        self.solver_parameters = solver_parameters or SOLVER_PARAMETERS
        self.tot_num_steps = ep.tot_num_steps if num_steps is None else int(num_steps)
        nv = self.mesh.num_vertices
        self.sys.initialise([1.0] * 6 + [0.0])
        self.history = [np.concatenate([np.ones((nv, 6)), np.zeros((nv, 1))], axis=1)]
        self.current_H_frac = ep.current_H_frac
        self.n, self.t, self.dt = 0, 0.0, ep.dts[0]
        self.newton_its = []

    def step(self, verbose=True):
        ep = self.ep
        if verbose:
            if ep.dry_run:
                print(int(self.t / self.dt))
            else:
                if self.t >= ep.stage_T[0]:
                    self.dt = ep.dts[1]  # only the clock changes: the form keeps the first Constant (Q2)
                    print(int(ep.stage_steps[0] + (self.t - ep.stage_T[0]) / self.dt))
                else:
                    print(int(self.t / self.dt))
        elif not ep.dry_run and self.t >= ep.stage_T[0]:
            self.dt = ep.dts[1]
        self.t += self.dt
        if self.warn_stab and verbose:
            print("Warning:stabilization not implemented for MPNP!")
        if self.supg:  # rho_i from the previous step's potential (u_n), OH's strong residual with grad(u_H)
            rho, self.h_vertex = supg_parameters(self.mesh.coords, self.mesh.cells, self.model.z, self.history[-1][:, 6],
                                                 self.sys.project_cellwise, self.h_vertex)
            w = np.arange(6, dtype=np.int32)
            w[ep.species.index("OH")] = ep.species.index("H")
            self.sys.dev.set_supg(rho, w)
        st = self.sys.solve(self.solver_parameters)
        vals = self.sys.vertex_values()
        self.history.append(vals)
        H_OHP_frac = vals[0, 0]
        H_OHP = ep.H_OHP
        if H_OHP is not None:  # reference 1D:770-793
            f = self.current_H_frac
            if H_OHP_frac < 0:
                f = f / 1.1
            elif H_OHP_frac < (H_OHP - 0.05):
                f = f / 1.05
            elif H_OHP_frac < (H_OHP - 0.025):
                f = f / 1.01
            elif (H_OHP_frac > H_OHP and H_OHP_frac <= (H_OHP + 0.4) and f <= 1.0):
                f = f * 1.04
            elif H_OHP_frac > (H_OHP + 0.4) and f <= 1.0:
                f = f * 1.15
            self.current_H_frac = f
            if verbose:
                print(H_OHP_frac)
                print(f)
            JH, JOH = ep.ohp_fluxes(f)
            self.model.point_flux[ep.species.index("H")] = JH
            self.model.point_flux[ep.species.index("OH")] = JOH
            self.sys.set_model(self.model)
        self.sys.assign_previous()
        self.newton_its.append(st["iterations"])
        self.n += 1
        return st

    def run(self, verbose=True):
        for _ in range(self.n, self.tot_num_steps):
            self.step(verbose)
        return self

    def ohp_summary(self):
        """field_OHP [V/nm] and eps_rel_OHP of the current state, as the reference derives them for metadata.json
        (1D:802-805 projection of -grad(p), 1D:893-954 rescaling): the two quantities 1D/Stern_CO2ER.py:66-68 records."""
        ep, mesh = self.ep, self.mesh
        last = self.history[-1]
        field = self.sys.project_gradient(last[:, 6], sign=-1.0)[:, 0] * ep.thermal_voltage / ep.L_n
        c_cat = last[0, 5] * ep.initial_conc[ep.cation]
        c_H = last[0, 0] * ep.initial_conc["H"]
        w = (ep.n_water[ep.cation] * c_cat + ep.n_water["H"] * c_H) * 1.0e-3
        return {"field_OHP": float(field[0] * 1.0e-9), "eps_rel_OHP": float(ep.eps_rel * ((55 - w) / 55) + 6 * (w / 55)),
                "potential_OHP": float(last[0, 6] * ep.thermal_voltage)}

    def write_outputs(self, stamp=None):
        ep, mesh, k = self.ep, self.mesh, self.kwargs
        stamp = stamp or datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        end_time = datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        identifier = ("voltage_" + str(ep.voltage_scaled) + "_H2_FE_" + str(k.get("H2_FE", 0.2)) + "_current_"
                      + str(ep.current_OHP_ss) + "_H_OHP_" + str(ep.H_OHP) + "_cation_" + ep.cation)
        newpath = os.path.join(output_root(), ep.model_name, stamp + "_experiment", identifier)
        os.makedirs(newpath, exist_ok=True)
        hist = np.stack(self.history)
        names = ["H", "OH", "HCO3", "CO32", "CO2", "cat", "p"]
        Hh = {nme: hist[:, :, i] for i, nme in enumerate(names)}
        field_values = self.sys.project_gradient(hist[-1][:, 6], sign=-1.0)[:, 0]
        field_values_rescaled = field_values * ep.thermal_voltage / ep.L_n
        field_OHP = field_values_rescaled[0] * 1.0e-9
        if ep.dry_run:
            tau_array = np.linspace(0, ep.stage_T[0], self.tot_num_steps)
        else:
            tau_array = np.concatenate((np.linspace(0, ep.stage_T[0], ep.stage_steps[0]),
                                        np.linspace(ep.stage_T[0] + ep.dts[1], ep.stage_T[1], ep.stage_steps[1])))
        np.savez(newpath + "/arrays_unscaled.npz", H=Hh["H"], OH=Hh["OH"], HCO3=Hh["HCO3"], CO32=Hh["CO32"],
                 CO2=Hh["CO2"], cat=Hh["cat"], p=Hh["p"], coor=mesh.coords, tau=tau_array, field_values=field_values)
        sc = {nme: scale(species=sp, tau=tau_array, C=Hh[nme], initial_conc=ep.initial_conc, diff_coeff=ep.diff_coeff,
                         L_n=ep.L_n, L_debye=ep.L_debye) for nme, sp in zip(names[:6], ep.species)}
        c = {nme: sc[nme][1] for nme in sc}
        psi = Hh["p"] * ep.thermal_voltage
        pH_OHP = -math.log10(c["H"][-1][0] / 1000)
        w = (ep.n_water[ep.cation] * c["cat"] + ep.n_water["H"] * c["H"]) * 1.0e-3
        eps_rel_conc_ss = ep.eps_rel * ((55 - w) / 55) + 6 * (w / 55)
        eps_rel_OHP = eps_rel_conc_ss[-1][0]
        charge_density = c["cat"][-1] - c["HCO3"][-1] - 2 * c["CO32"][-1] - c["OH"][-1] + c["H"][-1]
        np.savez(newpath + "/arrays_scaled.npz", x=mesh.coords * ep.L_n, psi=psi, t_H=sc["H"][0], c_H=c["H"],
                 t_OH=sc["OH"][0], c_OH=c["OH"], t_HCO3=sc["HCO3"][0], c_HCO3=c["HCO3"], t_CO32=sc["CO32"][0],
                 c_CO32=c["CO32"], t_CO2=sc["CO2"][0], c_CO2=c["CO2"], t_cat=sc["cat"][0], c_cat=c["cat"],
                 eps_rel=eps_rel_conc_ss, field_values=field_values_rescaled, charge_density=charge_density)
        potential_OHP = float(psi[-1][0])
        CO2_OHP_frac = c["CO2"][-1][0] / ep.initial_conc["CO2"]
        pH_overpotential = -0.059 * (ep.bulk_pH - pH_OHP) * 1.0e+3
        CO2_overpotential = (0.059 / 2) * math.log10(1 / CO2_OHP_frac) * 1.0e+3
        current_H = self.current_H_frac * ep.current_OHP_ss
        if not ep.dry_run:
            # reference 1D:971-972: time_step / total_sim_time are undefined outside the dry-run branch (Q3)
            raise NameError("name 'time_step' is not defined")
        metadata_dict = {
            "concentration_elec": k.get("concentration_elec", 0.1), "cation": ep.cation, "model": ep.model_name,
            "stabilization": k.get("stabilization", "N"), "voltage_multiplier": ep.voltage_scaled,
            "H2_FE": k.get("H2_FE", 0.2), "L_n_EDL": ep.L_n, "time_constant": ep.time_constant,
            "time_step": ep.time_step, "total_sim_time": ep.total_sim_time, "mesh_number": ep.mesh_number,
            "mesh_structure": ep.mesh_structure, "eps_rel_OHP": float(eps_rel_OHP), "field_OHP": float(field_OHP),
            "current_OHP_ss": ep.current_OHP_ss, "current_H": current_H, "H_OHP_vs_bulk": ep.H_OHP,
            "potential_OHP": potential_OHP, "pH_OHP": pH_OHP, "CO2_OHP_frac": float(CO2_OHP_frac),
            "pH_overpotential": pH_overpotential, "CO2_overpotential": CO2_overpotential, "end_time": end_time,
            "newton_iterations": int(sum(self.newton_its)), "krylov_iterations": int(self.sys.krylov_iterations),
            "num_steps_run": int(self.n)}
        with open(newpath + "/metadata.json", "w") as fh:
            fh.write(json.dumps(metadata_dict, indent=0))
        return newpath


def solve_EDL(concentration_elec=0.1, model="MPNP", voltage_multiplier=-1.0, H2_FE=0.2, mesh_structure="variable",
              current_OHP_ss=10.0, L_n=50.0e-6, stabilization="N", H_OHP=None, cation="K", params_file="parameters",
              dry_run=True, num_steps=None, verbose=True):
    """Same keyword surface as the reference's ``solve_EDL`` (1D:66-79); returns the output directory."""
    run = EDLRun(num_steps=num_steps, concentration_elec=concentration_elec, model=model,
                 voltage_multiplier=voltage_multiplier, H2_FE=H2_FE, mesh_structure=mesh_structure,
                 current_OHP_ss=current_OHP_ss, L_n=L_n, stabilization=stabilization, H_OHP=H_OHP, cation=cation,
                 params_file=params_file, dry_run=dry_run)
    try:
        run.run(verbose)
        return run.write_outputs()
    finally:
        run.sys.close()


def build_parser():
    """Flags, defaults and types of reference 1D:993-1101 (``--dry_run`` keeps ``type=bool``, Q3)."""
    p = argparse.ArgumentParser(description="experiment parameters")
    p.add_argument("--concentration_elec", required=False, default=0.1, type=float)
    p.add_argument("--model", required=False, default="MPNP", type=str)
    p.add_argument("--voltage_multiplier", required=False, default=-1.0, type=float)
    p.add_argument("--mesh_structure", required=False, default="variable", type=str)
    p.add_argument("--H2_FE", required=False, default=0.2, type=float)
    p.add_argument("--current_OHP_ss", required=False, default=10.0, type=float)
    p.add_argument("--L_n", required=False, default=50e-6, type=float)
    p.add_argument("--stabilization", required=False, default="N", type=str)
    p.add_argument("--H_OHP", required=False, default=None, type=float)
    p.add_argument("--cation", required=False, default="K", type=str)
    p.add_argument("--params_file", required=False, default="parameters", type=str)
    p.add_argument("--dry_run", required=False, default=True, type=bool)
    p.add_argument("--num_steps", required=False, default=None, type=int, help="(addition) run only the first N steps")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    return solve_EDL(concentration_elec=a.concentration_elec, model=a.model, voltage_multiplier=a.voltage_multiplier,
                     H2_FE=a.H2_FE, mesh_structure=a.mesh_structure, current_OHP_ss=a.current_OHP_ss, L_n=a.L_n,
                     stabilization=a.stabilization, H_OHP=a.H_OHP, cation=a.cation, params_file=a.params_file,
                     dry_run=a.dry_run, num_steps=a.num_steps)


if __name__ == "__main__":
    main()
