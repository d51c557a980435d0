"""Parameter and scaling layer: YAML tables -> scaled model constants.

Restates, scalar for scalar, reference 3D/MPNP_CO2ER_pore.py:115-325,469-499 (``pore_parameters``)
and 1D/MPNP_CO2ER_EDL.py:81-228,256-290,366-375 (``edl_parameters``).  Pure Python floats; the
operation order of the reference is kept so the constants agree to the last bit.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np
import yaml

from .model import Model

_HERE = os.path.dirname(os.path.abspath(__file__))


def utilities_dir() -> str:
    """Where ``<params_file>.yaml``, ``bulk_soln_*.yaml`` and the meshes live.  The reference
    hard-codes an absolute macOS path (3D:118, 1D:85); here: ``$GMPNP_UTILITIES`` or the
    repo's ``data/utilities``."""
    return os.environ.get("GMPNP_UTILITIES", os.path.join(os.path.dirname(_HERE), "data", "utilities"))


def _load_yaml(path):
    with open(path) as fh:  # missing file -> FileNotFoundError as in the reference's bare open()
        return yaml.safe_load(fh)


def co2_conc(temp, fugacity_CO2, conc_ions, h_sechenov):
    """Sechenov-corrected CO2 solubility [mol/m3] (reference 3D:70-93)."""
    lnK_H_CO2 = 93.4517 * (100 / temp) - 60.2409 + 23.3585 * math.log(temp / 100)
    h_CO2 = h_sechenov["CO2_0"] + h_sechenov["CO2_T"] * (temp - 298.15)
    sechenov = 0.0
    for ion in conc_ions.keys():
        sechenov += (h_sechenov[ion] + h_CO2) * (conc_ions[ion] / 1000)
    K_H_CO2 = math.exp(lnK_H_CO2)
    return fugacity_CO2 * K_H_CO2 * 1000 * 10 ** (-sechenov)


def _reaction_tables(species, conc, scale_R, k):
    """Coefficient tables of ``-R_i`` (what enters F as ``- R_i v``), reference 3D:505-532 / 1D:383-410.

    monomials: T1 = u_H u_OH, T2 = u_OH u_HCO3, T3 = u_CO2 u_OH; linear u_CO32, u_HCO3; constant kw1."""
    ns = len(species)
    ix = {s: i for i, s in enumerate(species)}
    H, OH, HCO3, CO32, CO2 = ix["H"], ix["OH"], ix["HCO3"], ix["CO32"], ix["CO2"]
    bil = [(H, OH), (OH, HCO3), (CO2, OH)]
    t1 = k["kw2"] * conc["H"] * conc["OH"]
    t2 = k["ka1"] * conc["OH"] * conc["HCO3"]
    t3 = k["kb1"] * conc["CO2"] * conc["OH"]
    l1 = k["ka2"] * conc["CO32"]  # * u_CO32
    l2 = k["kb2"] * conc["HCO3"]  # * u_HCO3
    rc0, rc1, rc2 = np.zeros(ns), np.zeros((ns, ns)), np.zeros((ns, len(bil)))
    s = scale_R
    # -R_H = s_H (T1 - kw1)
    rc2[H, 0] = s["H"] * t1
    rc0[H] = -s["H"] * k["kw1"]
    # -R_OH = s_OH (T1 + T2 + T3 - kw1 - l1 u_CO32 - l2 u_HCO3)
    rc2[OH, 0], rc2[OH, 1], rc2[OH, 2] = s["OH"] * t1, s["OH"] * t2, s["OH"] * t3
    rc0[OH] = -s["OH"] * k["kw1"]
    rc1[OH, CO32], rc1[OH, HCO3] = -s["OH"] * l1, -s["OH"] * l2
    # -R_HCO3 = s (T2 + l2 u_HCO3 - l1 u_CO32 - T3)
    rc2[HCO3, 1], rc2[HCO3, 2] = s["HCO3"] * t2, -s["HCO3"] * t3
    rc1[HCO3, HCO3], rc1[HCO3, CO32] = s["HCO3"] * l2, -s["HCO3"] * l1
    # -R_CO32 = s (l1 u_CO32 - T2)
    rc1[CO32, CO32] = s["CO32"] * l1
    rc2[CO32, 1] = -s["CO32"] * t2
    # -R_CO2 = s (T3 - l2 u_HCO3)
    rc2[CO2, 2] = s["CO2"] * t3
    rc1[CO2, HCO3] = -s["CO2"] * l2
    return rc0, rc1, bil, rc2


def _permittivity_tables(species, conc, n_water, eps_rel, cat):
    """eps(u) = eps_rel (55-w)/55 + 6 w/55, w = (n_cat c_cat + n_H c_H) 1e-3  (3D:752-760, 1D:412-420)
    rewritten as eps0 + sum_j epsc_j u_j (affine, identical up to rounding)."""
    epsc = np.zeros(len(species))
    for name in (cat, "H"):
        j = species.index(name)
        epsc[j] = (6.0 - eps_rel) / 55.0 * n_water[name] * conc[name] * 1.0e-3
    return float(eps_rel), epsc


# ---------------------------------------------------------------------------------------------
@dataclass
class PoreParameters:
    """Everything reference 3D:115-325,358-365,469-499 computes before the FEniCS part."""

    model: Model
    species: list
    cation: str
    L: float
    R: float
    aspect_pore: float
    bulk_conc: dict
    diff_coeff: dict
    diff_coeff_eff: dict
    time_constant: float
    time_step: float
    total_sim_time: float
    dt: float  # scaled step (3D:362)
    T: float
    tot_num_steps: int
    thermal_voltage: float
    voltage_scaled: float
    eq_conc_CO2_scaled: float
    eq_conc_CO_scaled: float
    eq_conc_H2_scaled: float
    eq_conc_CO: float
    eq_conc_H2: float
    current_planar: float
    temp: float
    fugacity_CO2: float
    h_sechenov: dict
    n_water: dict
    eps_rel: float
    mesh_name: str
    scalars: dict = field(default_factory=dict)  # named intermediates for the golden-scalar tests

    def sechenov_co2_scaled(self, med_OH, med_HCO3, med_CO32, med_cat) -> float:
        """Per-step CO2 Dirichlet value from median ion concentrations (reference 3D:817-835)."""
        b, c = self.bulk_conc, self.cation
        conc_ions = {"OH": med_OH * b["OH"], "HCO3": med_HCO3 * b["HCO3"], "CO32": med_CO32 * b["CO32"],
                     c: med_cat * b[c]}
        return co2_conc(self.temp, self.fugacity_CO2, conc_ions, self.h_sechenov) / b["CO2"]


def pore_parameters(concentration_elec=1.0, voltage_multiplier=-1.0, H2_FE=0.05, current_rough=3000.0,
                    L=100.0e-9, cation="K", R=5.0e-9, press_gas=1.0, pore_geom_multiplier=1.0,
                    porosity_eff=0.5, tortuosity_eff=1.5, constrictivity_eff=0.9,
                    params_file="parameters_pore", y_CO2=0.95, electrolyte_flow_geom_multiplier=1.0,
                    roughness_factor=150.0, as_published=False, utilities=None) -> PoreParameters:
    """Signature and defaults of reference ``solveEDL`` (3D:96-113).

    ``as_published=True`` drops the ds(2)/ds(3) flux terms, which the published script never adds to
    ``F`` (missing line continuations); the default keeps the intended physics."""
    utilities = utilities or utilities_dir()
    data = _load_yaml(os.path.join(utilities, params_file + ".yaml"))
    k = data["rate_constants"]
    cat = cation
    species = ["H", "OH", "HCO3", "CO32", "CO2", "CO", "H2", cat]
    diff_coeff = {i: data["diff_coef"]["D_" + i] for i in species}
    diff_coeff_eff = {
        i: (diff_coeff[i] * porosity_eff * constrictivity_eff * pore_geom_multiplier) / tortuosity_eff ** 2
        for i in species}
    n_water = {"H": data["Hydration_number"]["w_H"], cat: data["Hydration_number"]["w_" + cat]}
    solv_size = {i: data["solv_size"]["a_" + i] for i in species}
    nc = data["nat_const"]
    farad, k_B, e_0, eps_0, eps_rel, R_gas, N_A = (nc["F"], nc["k_B"], nc["e_0"], nc["eps_0"], nc["eps_rel"],
                                                   nc["R"], nc["N_A"])
    H_CO2, H_CO, H_H2 = (data["Henrys_const"][n] for n in ("H_CO2", "H_CO", "H_H2"))
    sp = data["sys_params"]
    temp, density_e, viscosity_e = sp["T"], sp["density_e"], sp["viscosity_e"]
    L_electrode, vel_e, A_cross_e, L_cross_e = sp["L_electrode"], sp["vel_e"], sp["A_cross_e"], sp["L_cross_e"]
    sc = data["sechonov_const"]
    h_sechenov = {"OH": sc["h_ion_OH"], "HCO3": sc["h_ion_HCO3"], "CO32": sc["h_ion_CO32"],
                  cat: sc["h_ion_" + cat], "CO2_0": sc["h_CO2_0"], "CO2_T": sc["h_CO2_T"]}

    y_CO = 0.9 * (1 - y_CO2)
    y_H2 = 1 - y_CO2 - y_CO
    fugacity_CO2 = y_CO2 * press_gas

    conc_data = _load_yaml(os.path.join(utilities, "bulk_soln_" + str(concentration_elec) + "KHCO3.yaml"))
    z = {"H": 1, "OH": -1, "HCO3": -1, "CO32": -2, "CO2": 0, "CO": 0, "H2": 0, cat: 1}
    bulk_conc = {}
    for i in species:  # KeyError for 0.1 M (no C0_CO/C0_H2) and for non-K cations: SURVEY Q9
        bulk_conc[i] = conc_data["bulk_conc_pre_CO2"]["concentrations"]["C0_" + i]

    eq_conc_CO2 = H_CO2 * press_gas * y_CO2 * density_e
    eq_conc_CO = H_CO * press_gas * y_CO * density_e
    eq_conc_H2 = H_H2 * press_gas * y_H2 * density_e
    bulk_conc["CO"] = 0.01 * eq_conc_CO
    bulk_conc["H2"] = 0.01 * eq_conc_H2
    eq_conc_CO2_scaled = eq_conc_CO2 / bulk_conc["CO2"]
    eq_conc_CO_scaled = eq_conc_CO / bulk_conc["CO"]
    eq_conc_H2_scaled = eq_conc_H2 / bulk_conc["H2"]

    aspect_pore = R / L
    thermal_voltage = (k_B * temp) / e_0
    time_constant = L ** 2 / diff_coeff_eff["CO32"]
    scale_R = {i: (L ** 2) / (diff_coeff_eff[i] * bulk_conc[i]) for i in species}
    q = (farad ** 2 * L ** 2) / (eps_0 * R_gas * temp)
    scale_vol = {i: solv_size[i] ** 3 * bulk_conc[i] * N_A for i in species}
    J_prefactor = {i: L / (diff_coeff_eff[i] * bulk_conc[i]) for i in species}
    Re = (density_e * (vel_e / A_cross_e) * L_electrode * electrolyte_flow_geom_multiplier) / viscosity_e
    Sc, Sh, k_elec = {}, {}, {}
    for i in species:
        Sc[i] = viscosity_e / (density_e * diff_coeff[i])
        Sh[i] = 1.017 * ((L_electrode * 2 / L_cross_e) * Re * Sc[i]) ** (1.0 / 3)
        k_elec[i] = (diff_coeff[i] / L_electrode) * Sh[i]

    time_step, total_sim_time = 1.0e-3, 1.0
    T = total_sim_time / time_constant
    dt = time_step / time_constant
    tot_num_steps = int(total_sim_time / time_step)

    CO_FE = 1 - H2_FE
    current_planar = current_rough / roughness_factor
    J_wall = {
        "CO2": (J_prefactor["CO2"] / farad) * current_planar * 0.5 * (CO_FE),
        "CO": (J_prefactor["CO"] / farad) * current_planar * 0.5 * (CO_FE) * (-1.0),
        "H2": (J_prefactor["H2"] / farad) * current_planar * 0.5 * (H2_FE) * (-1.0),
        "OH": (J_prefactor["OH"] / farad) * current_planar * (-1.0),
    }
    kappa = {i: J_prefactor[i] * k_elec[i] * bulk_conc[i] for i in species}

    ns = len(species)
    rc0, rc1, bil, rc2 = _reaction_tables(species, bulk_conc, scale_R, k)
    eps0, epsc = _permittivity_tables(species, bulk_conc, n_water, eps_rel, cat)
    wall_flux, exit_kappa = np.zeros(ns), np.zeros(ns)
    if not as_published:
        for name, val in J_wall.items():
            wall_flux[species.index(name)] = val
        for i, name in enumerate(species):
            exit_kappa[i] = kappa[name]
    model = Model(dim=3, species=species, z=np.array([z[i] for i in species], dtype=float),
                  bulk=np.array([bulk_conc[i] for i in species]), a=np.array([scale_vol[i] for i in species]),
                  inv_dt=1.0 / dt, q=q, eps0=eps0, epsc=epsc, rc0=rc0, rc1=rc1, bil=bil, rc2=rc2, steric=True,
                  wall_flux=wall_flux, exit_kappa=exit_kappa)
    mesh_name = "L_" + str(int(L * 1e+9)) + "_R_" + str(int(R * 1e+9)) + ".xml"  # 3D:330-331 (int() truncation: Q4)
    scalars = {"q": q, "time_constant": time_constant, "dt": dt, "Re": Re, "J_wall": J_wall, "kappa": kappa,
               "scale_vol_sum": float(sum(scale_vol.values())), "scale_R": scale_R, "k_elec": k_elec,
               "J_prefactor": J_prefactor}
    return PoreParameters(model=model, species=species, cation=cat, L=L, R=R, aspect_pore=aspect_pore,
                          bulk_conc=bulk_conc, diff_coeff=diff_coeff, diff_coeff_eff=diff_coeff_eff,
                          time_constant=time_constant, time_step=time_step, total_sim_time=total_sim_time, dt=dt,
                          T=T, tot_num_steps=tot_num_steps, thermal_voltage=thermal_voltage,
                          voltage_scaled=float(voltage_multiplier), eq_conc_CO2_scaled=eq_conc_CO2_scaled,
                          eq_conc_CO_scaled=eq_conc_CO_scaled, eq_conc_H2_scaled=eq_conc_H2_scaled,
                          eq_conc_CO=eq_conc_CO, eq_conc_H2=eq_conc_H2, current_planar=current_planar,
                          temp=temp, fugacity_CO2=fugacity_CO2, h_sechenov=h_sechenov, n_water=n_water,
                          eps_rel=eps_rel, mesh_name=mesh_name, scalars=scalars)


# ---------------------------------------------------------------------------------------------
@dataclass
class EDLParameters:
    """Everything reference 1D:81-290,366-375 computes before the FEniCS part."""

    model: Model
    species: list
    cation: str
    model_name: str
    L_n: float
    L_debye: float
    L_D: float
    initial_conc: dict
    diff_coeff: dict
    time_constant: float
    dry_run: bool
    dts: list  # scaled step sizes of the stages (1 entry in dry-run)
    stage_steps: list  # number of steps per stage
    stage_T: list  # scaled end time per stage
    tot_num_steps: int
    time_step: float
    total_sim_time: float
    thermal_voltage: float
    voltage_scaled: float
    bulk_pH: float
    current_OHP_ss: float
    current_H_frac: float
    H_OHP: object
    J_H_prefactor: float
    J_OH_prefactor: float
    J_CO2_prefactor: float
    n_water: dict
    eps_rel: float
    mesh_name: str
    mesh_number: int
    mesh_structure: str
    scalars: dict = field(default_factory=dict)

    def ohp_fluxes(self, current_H_frac):
        """(J_H, J_OH) at the OHP for a proton-current fraction (reference 1D:372-375, 789-793)."""
        J_OH = self.J_OH_prefactor * self.current_OHP_ss * (1 - current_H_frac) * (-1.0)
        J_H = self.J_H_prefactor * self.current_OHP_ss * current_H_frac
        return J_H, J_OH


def edl_parameters(concentration_elec=0.1, model="MPNP", voltage_multiplier=-1.0, H2_FE=0.2,
                   mesh_structure="variable", current_OHP_ss=10.0, L_n=50.0e-6, stabilization="N", H_OHP=None,
                   cation="K", params_file="parameters", dry_run=True, utilities=None) -> EDLParameters:
    """Signature and defaults of reference ``solve_EDL`` (1D:66-79)."""
    utilities = utilities or utilities_dir()
    data = _load_yaml(os.path.join(utilities, params_file + ".yaml"))
    k = data["rate_constants"]
    cat = cation
    n_water = {"H": 10.0, cat: 0.0}
    if cat == "K":
        n_water[cat] = 4
    elif cat == "Li":
        n_water[cat] = 5
    elif cat == "Cs":
        n_water[cat] = 3
    elif cat == "Na":
        n_water[cat] = 5
    species = ["H", "OH", "HCO3", "CO32", "CO2", cat]
    diff_coeff = {i: data["diff_coef"]["D_" + i] for i in species}
    solv_size = {i: data["solv_size"]["a_" + i] for i in species}
    nc = data["nat_const"]
    farad, temp, k_B, e_0, eps_0, eps_rel, R, N_A = (nc["F"], nc["T"], nc["k_B"], nc["e_0"], nc["eps_0"],
                                                     nc["eps_rel"], nc["R"], nc["N_A"])
    conc_data = _load_yaml(os.path.join(utilities, "bulk_soln_" + str(concentration_elec) + "KHCO3.yaml"))
    bulk_pH = conc_data["bulk_conc_post_CO2"]["final_pH"]
    z = {"H": 1, "OH": -1, "HCO3": -1, "CO32": -2, "CO2": 0, cat: 1}
    initial_conc = {i: conc_data["bulk_conc_post_CO2"]["concentrations"]["C0_" + i] for i in species}
    current_H_frac = 0.0 if H_OHP is None else 0.001

    L_debye = math.sqrt((eps_0 * eps_rel * k_B * temp) / (2 * e_0 ** 2 * concentration_elec * 1.0e+3 * N_A))
    L_D = L_debye / L_n
    thermal_voltage = (k_B * temp) / e_0
    time_constant = L_debye * L_n / diff_coeff["CO32"]
    scale_R = {i: (L_n ** 2) / (diff_coeff[i] * initial_conc[i]) for i in species}
    q = (farad ** 2 * L_n ** 2) / (eps_0 * R * temp)
    scale_vol = {i: solv_size[i] ** 3 * initial_conc[i] * N_A for i in species}
    J_H_prefactor = L_n / (diff_coeff["H"] * initial_conc["H"] * farad)
    J_OH_prefactor = L_n / (diff_coeff["OH"] * initial_conc["OH"] * farad)
    J_CO2_prefactor = L_n / (diff_coeff["CO2"] * initial_conc["CO2"] * farad)

    L_sys = int(L_n * 1.0e+6)
    mesh_number = None
    if mesh_structure == "variable":
        mesh_structure = mesh_structure + "_" + str(L_sys) + "um"
        if L_sys == 1:
            mesh_number = 1090
        elif L_sys == 5:
            mesh_number = 1490
        elif L_sys == 10:
            mesh_number = 1990
        elif L_sys == 50:
            mesh_number = 5990
    elif mesh_structure == "uniform":
        mesh_number = 1000
    if mesh_number is None:
        # reference: ``mesh_number`` is unbound here -> UnboundLocalError at 1D:233
        raise UnboundLocalError("local variable 'mesh_number' referenced before assignment")
    mesh_name = "1D_" + mesh_structure + "_mesh_" + str(mesh_number) + ".xml.gz"

    if dry_run:
        time_step, total_sim_time = 1.0e-5, 1.0e-3
        T = total_sim_time / time_constant
        dt = time_step / time_constant
        dts, stage_steps, stage_T = [dt], [int(total_sim_time / time_step)], [T]
    else:
        time_step_1, time_step_2 = 1.0e-5, 1.0e-3
        total_sim_time_1, total_sim_time_2 = 0.1, 10.1
        T_1, dt_1 = total_sim_time_1 / time_constant, time_step_1 / time_constant
        num_steps_1 = int(total_sim_time_1 / time_step_1)
        T_2, dt_2 = total_sim_time_2 / time_constant, time_step_2 / time_constant
        num_steps_2 = int((total_sim_time_2 - total_sim_time_1) / time_step_2)
        dts, stage_steps, stage_T = [dt_1, dt_2], [num_steps_1, num_steps_2], [T_1, T_2]
        time_step, total_sim_time = time_step_1, total_sim_time_2
    tot_num_steps = sum(stage_steps)

    CO_FE = 1 - H2_FE
    J_CO2 = J_CO2_prefactor * current_OHP_ss * 0.5 * (CO_FE)
    J_OH = J_OH_prefactor * current_OHP_ss * (1 - current_H_frac) * (-1.0)
    J_H = J_H_prefactor * current_OHP_ss * current_H_frac

    ns = len(species)
    rc0, rc1, bil, rc2 = _reaction_tables(species, initial_conc, scale_R, k)
    eps0, epsc = _permittivity_tables(species, initial_conc, n_water, eps_rel, cat)
    point_flux = np.zeros(ns)
    point_flux[species.index("CO2")] = J_CO2
    point_flux[species.index("OH")] = J_OH
    point_flux[species.index("H")] = J_H
    mdl = Model(dim=1, species=species, z=np.array([z[i] for i in species], dtype=float),
                bulk=np.array([initial_conc[i] for i in species]), a=np.array([scale_vol[i] for i in species]),
                inv_dt=1.0 / (dts[0] * L_D), q=q, eps0=eps0, epsc=epsc, rc0=rc0, rc1=rc1, bil=bil, rc2=rc2,
                steric=(model == "MPNP"), point_flux=point_flux)
    if model not in ("MPNP", "PNP"):
        # reference: neither branch defines F -> NameError at 1D:738
        raise NameError("name 'F' is not defined")
    scalars = {"q": q, "L_debye": L_debye, "dt": dts[0], "J_OH": J_OH, "J_CO2": J_CO2, "J_H": J_H,
               "time_constant": time_constant, "scale_vol_sum": float(sum(scale_vol.values()))}
    return EDLParameters(model=mdl, species=species, cation=cat, model_name=model, L_n=L_n, L_debye=L_debye,
                         L_D=L_D, initial_conc=initial_conc, diff_coeff=diff_coeff, time_constant=time_constant,
                         dry_run=bool(dry_run), dts=dts, stage_steps=stage_steps, stage_T=stage_T,
                         tot_num_steps=tot_num_steps, time_step=time_step, total_sim_time=total_sim_time,
                         thermal_voltage=thermal_voltage, voltage_scaled=float(voltage_multiplier), bulk_pH=bulk_pH,
                         current_OHP_ss=current_OHP_ss, current_H_frac=current_H_frac, H_OHP=H_OHP,
                         J_H_prefactor=J_H_prefactor, J_OH_prefactor=J_OH_prefactor,
                         J_CO2_prefactor=J_CO2_prefactor, n_water=n_water, eps_rel=eps_rel, mesh_name=mesh_name,
                         mesh_number=mesh_number, mesh_structure=mesh_structure, scalars=scalars)
