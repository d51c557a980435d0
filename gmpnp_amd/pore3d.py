"""3D cylindrical-pore GMPNP driver on the MI355X backend — same CLI flags, YAML/XML inputs and output layout as
reference 3D/MPNP_CO2ER_pore.py (``solveEDL`` 3D:96-1085, CLI 3D:1088-1253).

Differences, all explicit: input/output roots come from ``$GMPNP_UTILITIES`` / ``$GMPNP_OUT`` instead of the
author's hard-coded macOS paths; ``--num_steps`` (not in the reference) shortens the 1000-step loop;
``--as_published`` drops the ds(2)/ds(3) flux terms that the published script never adds to F; ``--refine N`` /
``--multilevel`` run on the N times uniformly refined mesh (with the multilevel term of the preconditioner)."""
from __future__ import annotations

import argparse
import json
import os
from datetime import datetime

import numpy as np

from .mesh import read_dolfin_xml, resolve_mesh_path
from .params import pore_parameters, utilities_dir
from .problem import pore_dirichlet, pore_problem
from .solver import GMPNPSystem, column_medians
from .vtk import write_pvd

SOLVER_PARAMETERS = {  # reference 3D:789-798
    "nonlinear_solver": "newton",
    "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": 1.0e-4,
                      "absolute_tolerance": 1.0e-4, "relaxation_parameter": 0.9},
}


def scale_conc_time(species="H", C=None, grad_c=None, bulk_conc=None, tau=None, diff_coeff_eff=None, L=0.0):
    """reference 3D:56-67"""
    c = C * bulk_conc[species]
    t = tau * (L ** 2) / diff_coeff_eff[species]
    grad_c_scaled = grad_c * bulk_conc[species] / L
    return c, t, grad_c_scaled


def output_root():
    return os.environ.get("GMPNP_OUT", os.path.join(os.getcwd(), "out"))


class PoreRun:
    """State of one pore simulation; ``step()`` is one pass of the reference's time loop body (3D:783-858)."""

    def __init__(self, num_steps=None, as_published=False, device_kwargs=None, solver_parameters=None, refine=0,
                 partition=None, multilevel=False, ml_theta=2.0, ml_sweeps=4, **kwargs):
        """``partition`` = (nparts, rank): solve this ONE problem across `nparts` mesh partitions (rank None: all of them in
        this process on one GPU; rank r: this process is rank r of a ``torch.distributed`` job, RCCL inside the library).
        ``multilevel`` (with ``refine`` > 0): the preconditioner gets the geometric multilevel term over the nested meshes
        (gmpnp_attach_coarse_level) — not a reference feature; it changes iteration counts of the linear solves, not results."""
        self.kwargs = kwargs
        self.pp = pore_parameters(as_published=as_published, **kwargs)
        self.mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), self.pp.mesh_name))
        self._levels = None
        if multilevel and refine > 0 and not partition:
            from .problem import pore_hierarchy
            self._levels = pore_hierarchy(self.pp, self.mesh, refine)
            self.problem, self.bnd = self._levels[0][0], self._levels[0][1]
        else:
            self.problem, self.bnd = pore_problem(self.pp, self.mesh, refine=refine)
        if refine:  # uniformly refined copy of the reference mesh (not a reference feature: roofline studies)
            from .mesh import Mesh
            self.mesh = Mesh(dim=3, coords=self.problem.coords, cells=self.problem.cells)
        if partition:
            from .solver import PartitionedSystem
            self.sys = PartitionedSystem(self.problem, partition[0], rank=partition[1], **(device_kwargs or {}))
        else:
            self.sys = GMPNPSystem(self.problem, **(device_kwargs or {}))
            if self._levels:   # coarser levels: ordinary handles of the parent meshes, attached below the finest one
                from . import backend
                dev_id = (device_kwargs or {}).get("device_id", 0)
                finer = self.sys.dev
                for k in range(1, len(self._levels)):
                    coarse = backend.DeviceSolver(self._levels[k][0], device_id=dev_id, shared_device=1)
                    finer.attach_coarse_level(coarse, self._levels[k - 1][2], theta=ml_theta, sweeps=ml_sweeps)
                    finer = coarse
        self.solver_parameters = solver_parameters or SOLVER_PARAMETERS
        self.tot_num_steps = self.pp.tot_num_steps if num_steps is None else int(num_steps)
        nv = self.mesh.num_vertices
        self.sys.initialise([1.0] * 8 + [0.0])
        # history rows: initial ones/zeros (3D:771-779), one row appended per step (3D:842-850)
        self.history = [np.concatenate([np.ones((nv, 8)), np.zeros((nv, 1))], axis=1)]
        self.CO2_min = None
        self.co2_bc = None  # CO2 Dirichlet value in force (None = the equilibrium value of the set-up)
        self.n = 0
        self.t = 0.0
        self.newton_its = []

    def step(self, verbose=True):
        self.t += self.pp.dt
        st = self.sys.solve(self.solver_parameters)
        vals = self.sys.vertex_values()
        # medians of the scaled ion concentrations -> Sechenov -> new CO2 Dirichlet value at S1 (3D:817-838)
        co2 = self.pp.sechenov_co2_scaled(*column_medians(vals, (1, 2, 3, 7)))
        self.co2_bc = co2
        self.sys.set_bcs(*pore_dirichlet(self.pp, self.bnd, co2))
        self.history.append(vals)
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

    # ---- outputs (3D:860-1085) -------------------------------------------------------------------
    def write_outputs(self, stamp=None):
        pp, mesh, k = self.pp, self.mesh, self.kwargs
        stamp = stamp or datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        end_time = datetime.now().strftime("%y-%m-%d-%H-%M-%S")
        L, R = pp.L, pp.R
        identifier = ("v_" + str(pp.voltage_scaled) + "_L_" + str(int(L * 1e+9)) + "_R_" + str(int(R * 1e+9))
                      + "_P_g_" + str(k.get("press_gas", 1.0)) + "_D_eff_" + str(k.get("pore_geom_multiplier", 1.0))
                      + "_Re_" + str(k.get("electrolyte_flow_geom_multiplier", 1.0))
                      + "_rough_" + str(k.get("roughness_factor", 150.0)))
        newpath = os.path.join(output_root(), stamp + "_experiment", identifier) + "/"
        os.makedirs(newpath, exist_ok=True)
        hist = np.stack(self.history)  # (steps+1, nv, 9)
        names = ["H", "OH", "HCO3", "CO32", "CO2", "CO", "H2", "cat", "p"]
        H = {nme: hist[:, :, i] for i, nme in enumerate(names)}
        last = hist[-1]
        for fname, col in (("CO", 5), ("K", 7), ("H2", 6), ("CO2", 4), ("OH", 1), ("H", 0), ("HCO3", 2), ("CO32", 3),
                           ("p", 8)):
            write_pvd(os.path.join(newpath, "solution_" + fname + ".pvd"), mesh.coords, mesh.cells, last[:, col],
                      "f_" + fname)
        # project(+-grad(u_n), W).compute_vertex_values(): flat, component-major (3D:884-909)
        grads = {}
        for i, nme in enumerate(names[:8]):
            grads[nme] = self.sys.project_gradient(last[:, i]).T.ravel()
        field_values = self.sys.project_gradient(last[:, 8], sign=-1.0).T.ravel()
        tau_array = np.linspace(0, pp.T, self.tot_num_steps)
        np.savez(newpath + "arrays_unscaled.npz", H=H["H"], OH=H["OH"], HCO3=H["HCO3"], CO32=H["CO32"], CO2=H["CO2"],
                 CO=H["CO"], H2=H["H2"], cat=H["cat"], p=H["p"], coor=mesh.coords, tau=tau_array,
                 field_values=field_values, H_grad=grads["H"], OH_grad=grads["OH"], HCO3_grad=grads["HCO3"],
                 CO32_grad=grads["CO32"], CO2_grad=grads["CO2"], CO_grad=grads["CO"], H2_grad=grads["H2"],
                 cat_grad=grads["cat"])
        sc = {}
        for nme, sp in zip(names[:8], pp.species):
            sc[nme] = scale_conc_time(species=sp, C=H[nme], grad_c=grads[nme], bulk_conc=pp.bulk_conc, tau=tau_array,
                                      diff_coeff_eff=pp.diff_coeff_eff, L=L)
        c = {nme: sc[nme][0] for nme in sc}
        psi = H["p"] * pp.thermal_voltage
        nw = pp.n_water
        w = (nw[pp.cation] * c["cat"] + nw["H"] * c["H"]) * 1.0e-3
        eps_rel_conc_ss = pp.eps_rel * ((55 - w) / 55) + 6 * (w / 55)
        charge_density = c["cat"][-1] - c["HCO3"][-1] - 2 * c["CO32"][-1] - c["OH"][-1] + c["H"][-1]
        np.savez(newpath + "arrays_scaled.npz", coor_scaled=mesh.coords * L, psi=psi,
                 t_H=sc["H"][1], c_H=c["H"], t_OH=sc["OH"][1], c_OH=c["OH"], t_HCO3=sc["HCO3"][1], c_HCO3=c["HCO3"],
                 t_CO32=sc["CO32"][1], c_CO32=c["CO32"], t_CO2=sc["CO2"][1], c_CO2=c["CO2"], t_CO=sc["CO"][1],
                 c_CO=c["CO"], t_H2=sc["H2"][1], c_H2=c["H2"], t_cat=sc["cat"][1], c_cat=c["cat"],
                 eps_rel=eps_rel_conc_ss, field_values=field_values * pp.thermal_voltage / L,
                 charge_density=charge_density, H_grad=sc["H"][2], OH_grad=sc["OH"][2], HCO3_grad=sc["HCO3"][2],
                 CO32_grad=sc["CO32"][2], CO2_grad=sc["CO2"][2], CO_grad=sc["CO"][2], H2_grad=sc["H2"][2],
                 cat_grad=sc["cat"][2])
        metadata_dict = {
            "concentration_elec": k.get("concentration_elec", 1.0), "cation": pp.cation,
            "voltage_multiplier": pp.voltage_scaled, "H2_FE": k.get("H2_FE", 0.05), "L": L, "R": R,
            "time_step": pp.time_step, "total_sim_time": pp.total_sim_time, "porosity": k.get("porosity_eff", 0.5),
            "tortuosity": k.get("tortuosity_eff", 1.5), "constrictivity": k.get("constrictivity_eff", 0.9),
            "y_CO2": k.get("y_CO2", 0.95), "press_gas": k.get("press_gas", 1.0),
            "pore_geom_multiplier": k.get("pore_geom_multiplier", 1.0),
            "electrolyte_flow_geom_multiplier": k.get("electrolyte_flow_geom_multiplier", 1.0), "end_time": end_time,
            "eq_conc_CO": pp.eq_conc_CO, "eq_conc_H2": pp.eq_conc_H2, "current_planar": pp.current_planar,
            "CO2_min": self.CO2_min,
            # additions of this backend (new keys only)
            "newton_iterations": int(sum(self.newton_its)), "krylov_iterations": int(self.sys.krylov_iterations),
            "num_steps_run": int(self.n)}
        with open(newpath + "metadata.json", "w") as fh:
            fh.write(json.dumps(metadata_dict, indent=0))
        return newpath


def solveEDL(concentration_elec=1.0, voltage_multiplier=-1.0, H2_FE=0.05, current_rough=3000.0, L=100.0e-9,
             cation="K", R=5.0e-9, press_gas=1.0, pore_geom_multiplier=1.0, porosity_eff=0.5, tortuosity_eff=1.5,
             constrictivity_eff=0.9, params_file="parameters_pore", y_CO2=0.95, electrolyte_flow_geom_multiplier=1.0,
             roughness_factor=150.0, num_steps=None, as_published=False, verbose=True, refine=0, multilevel=False):
    """Same keyword surface as the reference's ``solveEDL`` (3D:96-113); returns the output directory.  Additions:
    ``num_steps``, ``as_published``, ``refine`` (uniform refinements of the mesh file), ``multilevel`` (with ``refine`` > 0: the
    geometric multilevel term of the preconditioner)."""
    run = PoreRun(num_steps=num_steps, as_published=as_published, refine=refine, multilevel=multilevel, concentration_elec=concentration_elec,
                  voltage_multiplier=voltage_multiplier, H2_FE=H2_FE, current_rough=current_rough, L=L, cation=cation,
                  R=R, press_gas=press_gas, pore_geom_multiplier=pore_geom_multiplier, porosity_eff=porosity_eff,
                  tortuosity_eff=tortuosity_eff, constrictivity_eff=constrictivity_eff, params_file=params_file,
                  y_CO2=y_CO2, electrolyte_flow_geom_multiplier=electrolyte_flow_geom_multiplier,
                  roughness_factor=roughness_factor)
    try:
        run.run(verbose)
        return run.write_outputs()
    finally:
        run.sys.close()


def build_parser():
    """Flags, defaults and types of reference 3D:1089-1233."""
    p = argparse.ArgumentParser(description="experiment parameters")
    for name, default in (("concentration_elec", 1.0), ("voltage_multiplier", -1.0), ("H2_FE", 0.05),
                          ("current_rough", 3000.0), ("L", 100e-9), ("R", 5e-9)):
        p.add_argument("--" + name, required=False, default=default, type=float)
    p.add_argument("--cation", required=False, default="K", type=str)
    for name, default in (("porosity_eff", 0.5), ("tortuosity_eff", 1.5), ("constrictivity_eff", 0.9),
                          ("press_gas", 1.0), ("pore_geom_multiplier", 1.0), ("electrolyte_flow_geom_multiplier", 1.0)):
        p.add_argument("--" + name, required=False, default=default, type=float)
    p.add_argument("--params_file", required=False, default="parameters_pore", type=str)
    p.add_argument("--y_CO2", required=False, default=0.95, type=float)
    p.add_argument("--roughness_factor", required=False, default=150.0, type=float)
    # additions (not in the reference)
    p.add_argument("--num_steps", required=False, default=None, type=int, help="run only the first N time steps")
    p.add_argument("--as_published", action="store_true", help="drop the ds(2)/ds(3) flux terms (SURVEY Q1)")
    p.add_argument("--refine", required=False, default=0, type=int, help="uniform (red) refinements of the mesh file, markers inherited")
    p.add_argument("--multilevel", action="store_true", help="with --refine > 0: geometric multilevel term of the preconditioner over the nested meshes")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    return solveEDL(concentration_elec=a.concentration_elec, voltage_multiplier=a.voltage_multiplier, H2_FE=a.H2_FE,
                    current_rough=a.current_rough, L=a.L, R=a.R, press_gas=a.press_gas, cation=a.cation,
                    porosity_eff=a.porosity_eff, tortuosity_eff=a.tortuosity_eff,
                    constrictivity_eff=a.constrictivity_eff, params_file=a.params_file, y_CO2=a.y_CO2,
                    pore_geom_multiplier=a.pore_geom_multiplier,
                    electrolyte_flow_geom_multiplier=a.electrolyte_flow_geom_multiplier,
                    roughness_factor=a.roughness_factor, num_steps=a.num_steps, as_published=a.as_published, refine=a.refine,
                    multilevel=a.multilevel)


if __name__ == "__main__":
    main()
