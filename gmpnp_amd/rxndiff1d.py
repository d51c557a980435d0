"""1D reaction-diffusion driver (planar electrode, no double layer) on the MI355X backend — same CLI flags, YAML/XML
inputs and output layout as reference 1D/rxn_diff_planar.py (``solve_rxn_diff`` 1D/rxn_diff_planar.py:87-486, CLI
:488-561).

The reference solves five species (H, OH, HCO3, CO32, CO2) with diffusion, the three homogeneous reactions and the OHP
fluxes of OH and CO2 (:232-318).  That is the GMPNP weak form with the potential, the migration and the steric terms
switched off, so it runs on the very same kernels: the 1D backend carries 6 species + potential; here every valence
is 0 (no migration, the Poisson row decouples and keeps p = 0 between its two Dirichlet ends), the steric flag is off,
and the sixth species (the cation, not solved for by the reference) has no source and stays at 1.  The 7 x 7 node
blocks make the solve ~2x the arithmetic a 5 x 5 layout would need; the 1D case is latency bound either way.

Differences, all explicit: input/output roots come from ``$GMPNP_UTILITIES`` / ``$GMPNP_OUT``;
``--num_steps`` (not in the reference) shortens the 500-step loop; ``--cation`` is accepted (the reference's
``solve_rxn_diff`` has the keyword, its CLI does not pass it)."""
from __future__ import annotations

import argparse
import json
import math
import os
from dataclasses import dataclass
from datetime import datetime

import numpy as np

from .mesh import read_dolfin_xml, resolve_mesh_path
from .model import Model
from .params import _load_yaml, _reaction_tables, utilities_dir
from .problem import edl_problem
from .solver import GMPNPSystem

SOLVER_PARAMETERS = {  # reference 1D/rxn_diff_planar.py:326-335 (relaxation 1.0 and the default LU are [3P] defaults)
    "nonlinear_solver": "newton",
    "newton_solver": {"maximum_iterations": 100, "relative_tolerance": 1.0e-6, "absolute_tolerance": 1.0e-6},
}
SOLVED = ["H", "OH", "HCO3", "CO32", "CO2"]  # the reference's MixedElement([P1]*5), :221


def scale(species="H", tau=None, C=None, initial_conc=None, diff_coeff=None, L_n=0.0):
    """reference 1D/rxn_diff_planar.py:54-66"""
    t = (tau * L_n ** 2) / diff_coeff[species]
    c = C * initial_conc[species]
    return t, c


def output_root():
    return os.environ.get("GMPNP_OUT", os.path.join(os.getcwd(), "out"))


@dataclass
class RxnDiffParameters:
    """Everything reference 1D/rxn_diff_planar.py:97-246 computes before the FEniCS part."""
    model: Model
    species: list            # the backend's six species (the five solved ones + the cation placeholder)
    cation: str
    concentration_KHCO3: float
    H2_FE: float
    CO_FE: float
    L_n: float
    current_OHP_ss: float
    mesh_structure: str
    mesh_name: str
    diff_coeff: dict
    initial_conc: dict
    bulk_pH: float
    time_constant: float
    total_sim_time: float
    time_step: float
    T: float
    dt: float
    num_steps: int
    scalars: dict
    voltage_scaled: float = 0.0    # read by problem.edl_problem: p = 0 at both ends
    H_OHP: object = None           # read by the oracle's 1D time loop: no proton-flux controller here
    current_H_frac: float = 0.0


def rxn_diff_parameters(concentration_KHCO3=0.1, H2_FE=0.2, L_n=50.0e-6, mesh_structure="variable", current_OHP_ss=10.0,
                        cation="K", params_file="parameters", utilities=None) -> RxnDiffParameters:
    """Signature and defaults of reference ``solve_rxn_diff`` (1D/rxn_diff_planar.py:87-95)."""
    utilities = utilities or utilities_dir()
    data = _load_yaml(os.path.join(utilities, params_file + ".yaml"))
    k = data["rate_constants"]
    cat = cation
    species = SOLVED + [cat]
    diff_coeff = {i: data["diff_coef"]["D_" + i] for i in species}
    farad = data["nat_const"]["F"]
    conc_data = _load_yaml(os.path.join(utilities, "bulk_soln_" + str(concentration_KHCO3) + "KHCO3.yaml"))
    bulk_pH = conc_data["bulk_conc_post_CO2"]["final_pH"]
    initial_conc = {i: conc_data["bulk_conc_post_CO2"]["concentrations"]["C0_" + i] for i in species}
    time_constant = L_n ** 2 / diff_coeff["CO32"]                                   # :151
    scale_R = {i: (L_n ** 2) / (diff_coeff[i] * initial_conc[i]) for i in species}  # :157-158
    J_OH_prefactor = L_n / (diff_coeff["OH"] * initial_conc["OH"] * farad)          # :161-162
    J_CO2_prefactor = L_n / (diff_coeff["CO2"] * initial_conc["CO2"] * farad)

    L_sys = int(L_n * 1.0e+6)
    mesh_number = None
    if mesh_structure == "variable":                                               # :169-183
        mesh_structure = mesh_structure + "_" + str(L_sys) + "um"
        mesh_number = {1: 1090, 5: 1490, 10: 1990, 50: 5990}.get(L_sys)
    elif mesh_structure == "uniform":
        mesh_number = 1000
    if mesh_number is None:  # the reference leaves ``mesh_number`` unbound here
        raise UnboundLocalError("local variable 'mesh_number' referenced before assignment")
    mesh_name = "1D_" + mesh_structure + "_mesh_" + str(mesh_number) + ".xml.gz"

    total_sim_time, time_step = 10, 2.0e-2                                         # :201-202
    T = total_sim_time / time_constant
    dt = time_step / time_constant
    num_steps = int(T / dt)                                                        # range(int(num_steps)), :320

    CO_FE = 1 - H2_FE
    J_CO2 = J_CO2_prefactor * current_OHP_ss * 0.5 * CO_FE                         # :243-244
    J_OH = J_OH_prefactor * current_OHP_ss * (-1.0)

    ns = len(species)
    rc0, rc1, bil, rc2 = _reaction_tables(species, initial_conc, scale_R, k)       # R_H ... R_CO2, :253-288
    point_flux = np.zeros(ns)
    point_flux[species.index("CO2")] = J_CO2                                       # "+ J_OH v_OH ds + J_CO2 v_CO2 ds", :305
    point_flux[species.index("OH")] = J_OH
    mdl = Model(dim=1, species=species, z=np.zeros(ns), bulk=np.array([initial_conc[i] for i in species]), a=np.zeros(ns),
                inv_dt=1.0 / dt, q=0.0, eps0=1.0, epsc=np.zeros(ns), rc0=rc0, rc1=rc1, bil=bil, rc2=rc2, steric=False,
                point_flux=point_flux)
    scalars = {"dt": dt, "T": T, "J_OH": J_OH, "J_CO2": J_CO2, "time_constant": time_constant}
    return RxnDiffParameters(model=mdl, species=species, cation=cat, concentration_KHCO3=concentration_KHCO3, H2_FE=H2_FE,
                             CO_FE=CO_FE, L_n=L_n, current_OHP_ss=current_OHP_ss, mesh_structure=mesh_structure,
                             mesh_name=mesh_name, diff_coeff=diff_coeff, initial_conc=initial_conc, bulk_pH=bulk_pH,
                             time_constant=time_constant, total_sim_time=total_sim_time, time_step=time_step, T=T, dt=dt,
                             num_steps=num_steps, scalars=scalars)


class RxnDiffRun:
    """State of one run; ``step()`` is one pass of the reference's time loop body (1D/rxn_diff_planar.py:320-360)."""

    def __init__(self, num_steps=None, device_kwargs=None, solver_parameters=None, **kwargs):
        self.kwargs = kwargs
        self.rp = rxn_diff_parameters(**kwargs)
        self.mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), self.rp.mesh_name))
        # Dirichlet: every field at x = 1 (bulk = 1, p = 0), p = 0 at x = 0; point fluxes at the x = 0 vertex (the
        # reference's ``ds`` also covers x = 1, where the Dirichlet rows replace the equations)
        self.problem = edl_problem(self.rp, self.mesh)
        self.sys = GMPNPSystem(self.problem, **(device_kwargs or {}))
        self.solver_parameters = solver_parameters or SOLVER_PARAMETERS
        self.tot_num_steps = self.rp.num_steps if num_steps is None else int(num_steps)
        nv = self.mesh.num_vertices
        self.sys.initialise([1.0] * 6 + [0.0])
        self.history = [np.ones((nv, 5))]  # H = np.ones(num_vertices) ..., :308-312
        self.n, self.t = 0, 0.0
        self.newton_its = []

    def step(self, verbose=True):
        self.t += self.rp.dt
        st = self.sys.solve(self.solver_parameters)
        self.history.append(self.sys.vertex_values()[:, :5].copy())
        self.sys.assign_previous()
        self.newton_its.append(st["iterations"])
        if verbose:
            print(self.n)
        self.n += 1
        return st

    def run(self, verbose=True):
        for _ in range(self.n, self.tot_num_steps):
            self.step(verbose)
        return self

    def write_outputs(self, stamp=None):
        """arrays_unscaled.npz, arrays_scaled.npz, metadata.json as in 1D/rxn_diff_planar.py:362-486."""
        rp, mesh = self.rp, self.mesh
        stamp = stamp or datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        identifier = ("H2_FE_" + str(rp.H2_FE) + "_current_" + str(rp.current_OHP_ss) + "_L_n_" + str(rp.L_n)
                      + "_cation_" + rp.cation)
        newpath = os.path.join(output_root(), stamp + "_experiment", identifier)
        os.makedirs(newpath, exist_ok=True)
        hist = np.stack(self.history)
        Hh = {nme: hist[:, :, i] for i, nme in enumerate(SOLVED)}
        tau_array = np.linspace(0, rp.T, self.tot_num_steps)
        np.savez(os.path.join(newpath, "arrays_unscaled.npz"), coor_array=mesh.coords, tau_array=tau_array, **Hh)
        sc = {nme: scale(species=nme, tau=tau_array, C=Hh[nme], initial_conc=rp.initial_conc, diff_coeff=rp.diff_coeff,
                         L_n=rp.L_n) for nme in SOLVED}
        c = {nme: sc[nme][1] for nme in SOLVED}
        c_cat = c["HCO3"] + 2 * c["CO32"] + c["OH"] - c["H"]  # electroneutrality, :428
        pH_OHP = -math.log10(c["H"][-1][0] / 1000)
        out = {"x": mesh.coords * rp.L_n, "c_cat": c_cat}
        for nme in SOLVED:
            out["t_" + nme], out["c_" + nme] = sc[nme]
        np.savez(os.path.join(newpath, "arrays_scaled.npz"), **out)
        CO2_surf = c["CO2"][-1][0]
        meta = {"concentration_KHCO3": rp.concentration_KHCO3, "L_n": rp.L_n, "bulk_pH": rp.bulk_pH,
                "time_constant": rp.time_constant, "total_sim_time": rp.total_sim_time, "time_step": rp.time_step,
                "mesh_structure": rp.mesh_structure, "H2_FE": rp.H2_FE, "CO_FE": rp.CO_FE,
                "current_OHP_ss": rp.current_OHP_ss, "pH_OHP": pH_OHP,
                "pH_overpotential": -0.059 * (rp.bulk_pH - pH_OHP) * 1.0e+3,
                "CO2_overpotential": (0.059 / 2) * math.log10(rp.initial_conc["CO2"] / CO2_surf) * 1.0e+3,
                "CO2_OHP_frac": CO2_surf / rp.initial_conc["CO2"], "num_steps_run": int(self.n)}
        with open(os.path.join(newpath, "metadata.json"), "w") as fh:
            fh.write(json.dumps(meta, indent=0))
        return newpath


def solve_rxn_diff(concentration_KHCO3=0.1, H2_FE=0.2, L_n=50.0e-6, mesh_structure="variable", current_OHP_ss=10.0,
                   cation="K", params_file="parameters", num_steps=None, verbose=True):
    """Same keyword surface as the reference's ``solve_rxn_diff``; returns the output directory."""
    run = RxnDiffRun(num_steps=num_steps, concentration_KHCO3=concentration_KHCO3, H2_FE=H2_FE, L_n=L_n,
                     mesh_structure=mesh_structure, current_OHP_ss=current_OHP_ss, cation=cation, params_file=params_file)
    try:
        run.run(verbose=verbose)
        return run.write_outputs()
    finally:
        run.sys.close()


def build_parser():
    p = argparse.ArgumentParser(description="experiment parameters")  # reference :489-546
    p.add_argument("--concentration_KHCO3", metavar="electrolyte_concentration", required=False, help="float val, 0.1 M",
                   default=0.1, type=float)
    p.add_argument("--mesh_structure", metavar="bias in mesh structure", required=False, help="str, uniform/variable",
                   default="variable", type=str)
    p.add_argument("--H2_FE", metavar="faradaic efficiency for hydrogen in fraction", required=False, help="float val, 0.2",
                   default=0.2, type=float)
    p.add_argument("--L_n", metavar="Nernst boundary layer thickness", required=False, help="float val, 50.0e-6",
                   default=50.0e-6, type=float)
    p.add_argument("--current_OHP_ss", metavar="steady state current in A/m2", required=False, help="float val, 10.0",
                   default=10.0, type=float)
    p.add_argument("--params_file", metavar="yaml file with parameter values", required=False, help="str, parameters",
                   default="parameters", type=str)
    p.add_argument("--cation", required=False, default="K", type=str, help="str, K/Li/Na/Cs (keyword of solve_rxn_diff)")
    p.add_argument("--num_steps", required=False, default=None, type=int, help="int, time steps to run (default: all 500)")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    return solve_rxn_diff(concentration_KHCO3=a.concentration_KHCO3, H2_FE=a.H2_FE, L_n=a.L_n, mesh_structure=a.mesh_structure,
                          current_OHP_ss=a.current_OHP_ss, cation=a.cation, params_file=a.params_file, num_steps=a.num_steps)
