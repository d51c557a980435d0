"""ctypes front end of oracle/edl1d_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as gmpnp_oracle.py: only
tests/, tools/, smoke() and bench.py's cpu_baseline leg may import it).

Independent of gmpnp_amd on purpose: the parameter block below is transcribed from reference 1D/MPNP_CO2ER_EDL.py:89-290,366-375
with the reference's own names, the mesh is read with a few lines of regex, and the C side evaluates the published integrands
literally at Gauss points.  What it shares with the product is the YAML / mesh DATA under data/utilities only.
"""
from __future__ import annotations

import ctypes
import gzip
import os
import re
import subprocess
from math import sqrt

import numpy as np
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "edl1d_oracle.c")
LIBS = {"double": os.path.join(_HERE, "libedl1d_oracle.so"), "long double": os.path.join(_HERE, "libedl1d_oracle_ld.so")}

SPECIES = ["H", "OH", "HCO3", "CO32", "CO2", "cat"]
NF = 7
BAND = 27

# 1D/Stern_CO2ER.py:66-68 (OHP_dict): voltage_multiplier -> (field_OHP [V/nm], eps_rel_OHP); data, not code
RECORDED = {-2.5: (-0.08032108300135771, 74.56149297894756), -5.0: (-0.2524415478848975, 57.64572780716129),
            -7.5: (-0.4612956299192668, 50.16243860179017), -10.0: (-0.6149631587776277, 49.311548142969336),
            -12.5: (-0.7310301485096051, 49.2556833480052)}


class EdlParams(ctypes.Structure):
    _fields_ = [("nv", ctypes.c_int), ("steric", ctypes.c_int), ("nq_f", ctypes.c_int), ("nq_j", ctypes.c_int),
                ("max_it", ctypes.c_int), ("rtol", ctypes.c_double), ("atol", ctypes.c_double), ("relax", ctypes.c_double),
                ("conc", ctypes.c_double * 6), ("z", ctypes.c_double * 6), ("scale_R", ctypes.c_double * 6),
                ("scale_vol", ctypes.c_double * 6),
                ("kw1", ctypes.c_double), ("kw2", ctypes.c_double), ("ka1", ctypes.c_double), ("ka2", ctypes.c_double),
                ("kb1", ctypes.c_double), ("kb2", ctypes.c_double),
                ("eps_rel", ctypes.c_double), ("n_water_cat", ctypes.c_double), ("n_water_H", ctypes.c_double),
                ("q", ctypes.c_double), ("del_t", ctypes.c_double), ("L_D", ctypes.c_double),
                ("J_CO2", ctypes.c_double), ("J_OH", ctypes.c_double), ("J_H", ctypes.c_double), ("voltage", ctypes.c_double)]


def build(force=False):
    """gcc the C restatement (double and x87 extended precision builds)."""
    for kind, lib in LIBS.items():
        if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(SRC):
            cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-lm", "-o", lib]
            if kind == "long double":
                cmd.insert(1, "-DEDL_REAL_LONG_DOUBLE")
            subprocess.run(cmd, check=True)


_loaded = {}


def load(kind="double"):
    if kind not in _loaded:
        build()
        lib = ctypes.CDLL(LIBS[kind])
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        pp = ctypes.POINTER(EdlParams)
        lib.edl_residual_jacobian.argtypes = [pp, dp, dp, dp, dp, dp]
        lib.edl_residual_jacobian.restype = None
        lib.edl_newton_solve.argtypes = [pp, dp, dp, dp, dp]
        lib.edl_newton_solve.restype = ctypes.c_int
        lib.edl_run.argtypes = [pp, dp, ctypes.c_int, dp, dp, ip, dp]
        lib.edl_run.restype = ctypes.c_int
        lib.edl_project_neg_gradient.argtypes = [ctypes.c_int, dp, dp, dp]
        lib.edl_project_neg_gradient.restype = None
        lib.edl_real_mantissa.restype = ctypes.c_int
        _loaded[kind] = lib
    return _loaded[kind]


def utilities_dir():
    return os.environ.get("GMPNP_UTILITIES", os.path.join(_ROOT, "data", "utilities"))


def read_interval_mesh(path):
    """DOLFIN-XML interval mesh -> ascending vertex coordinates (reference 1D:231-234)."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        txt = fh.read()
    x = np.array([float(m) for m in re.findall(r'<vertex index="\d+" x="([^"]+)"', txt)])
    cells = np.array([[int(a), int(b)] for a, b in re.findall(r'<interval index="\d+" v0="(\d+)" v1="(\d+)"', txt)])
    assert np.all(np.diff(x) > 0) and np.all(cells[:, 1] - cells[:, 0] == 1), "file order is path order on the shipped meshes"
    return x


class Setup:
    """Everything reference 1D:89-290,366-375 computes before the FEniCS part, with the reference's names."""

    def __init__(self, concentration_elec=0.1, model="MPNP", voltage_multiplier=-1.0, H2_FE=0.2, mesh_structure="variable",
                 current_OHP_ss=10.0, L_n=50.0e-6, cation="K", params_file="parameters", dry_run=True, nq_f=2, nq_j=2):
        base = utilities_dir()
        with open(os.path.join(base, params_file + ".yaml")) as fh:
            data = yaml.safe_load(fh)
        rate_constants = data["rate_constants"]
        cat_str = cation
        n_water = {"H": 10.0, cat_str: {"K": 4, "Li": 5, "Cs": 3, "Na": 5}[cat_str]}
        species = ["H", "OH", "HCO3", "CO32", "CO2", cat_str]
        diff_coeff = {i: data["diff_coef"]["D_" + i] for i in species}
        solv_size = {i: data["solv_size"]["a_" + i] for i in species}
        nc = data["nat_const"]
        farad, temp, k_B, e_0, eps_0, eps_rel, R, N_A = (nc["F"], nc["T"], nc["k_B"], nc["e_0"], nc["eps_0"], nc["eps_rel"],
                                                            nc["R"], nc["N_A"])
        with open(os.path.join(base, "bulk_soln_" + str(concentration_elec) + "KHCO3.yaml")) as fh:
            bdata = yaml.safe_load(fh)
        z = {"H": 1, "OH": -1, "HCO3": -1, "CO32": -2, "CO2": 0, cat_str: 1}
        initial_conc = {i: bdata["bulk_conc_post_CO2"]["concentrations"]["C0_" + i] for i in species}
        current_H_frac = 0.0  # H_OHP is None
        L_debye = sqrt((eps_0 * eps_rel * k_B * temp) / (2 * e_0 ** 2 * concentration_elec * 1.0e+3 * N_A))
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
        mesh_number = {1: 1090, 5: 1490, 10: 1990, 50: 5990}[L_sys]
        mesh_name = "1D_" + mesh_structure + "_" + str(L_sys) + "um" + "_mesh_" + str(mesh_number) + ".xml.gz"
        if dry_run:
            time_step, total_sim_time = 1.0e-5, 1.0e-3
            dt = time_step / time_constant
            tot_num_steps = int(total_sim_time / time_step)
        else:  # the form keeps Constant(dt_1) for all 20,000 solves
            time_step_1, time_step_2, total_sim_time_1, total_sim_time_2 = 1.0e-5, 1.0e-3, 0.1, 10.1
            dt = time_step_1 / time_constant
            tot_num_steps = int(total_sim_time_1 / time_step_1) + int((total_sim_time_2 - total_sim_time_1) / time_step_2)
        CO_FE = 1 - H2_FE
        J_CO2 = J_CO2_prefactor * current_OHP_ss * 0.5 * (CO_FE)
        J_OH = J_OH_prefactor * current_OHP_ss * (1 - current_H_frac) * (-1.0)
        J_H = J_H_prefactor * current_OHP_ss * current_H_frac

        self.x = read_interval_mesh(os.path.join(base, mesh_name))
        self.nv = len(self.x)
        self.tot_num_steps = tot_num_steps
        self.thermal_voltage, self.L_n, self.eps_rel, self.n_water, self.initial_conc = thermal_voltage, L_n, eps_rel, n_water, initial_conc
        self.cat_str = cat_str
        p = EdlParams()
        p.nv, p.steric, p.nq_f, p.nq_j, p.max_it = self.nv, int(model == "MPNP"), nq_f, nq_j, 50
        p.rtol, p.atol, p.relax = 1.0e-4, 1.0e-4, 1.0  # reference 1D:357-364; relaxation_parameter default [3P] = 1
        for k, s in enumerate(species):
            p.conc[k], p.z[k], p.scale_R[k], p.scale_vol[k] = initial_conc[s], z[s], scale_R[s], scale_vol[s]
        p.kw1, p.kw2, p.ka1, p.ka2, p.kb1, p.kb2 = (rate_constants[k] for k in ("kw1", "kw2", "ka1", "ka2", "kb1", "kb2"))
        p.eps_rel, p.n_water_cat, p.n_water_H = eps_rel, n_water[cat_str], n_water["H"]
        p.q, p.del_t, p.L_D = q, dt, L_D
        p.J_CO2, p.J_OH, p.J_H, p.voltage = J_CO2, J_OH, J_H, voltage_multiplier
        self.params = p

    def initial_state(self):
        """u = Function(V) (zeros, 1D:320); u_n = project(u_0) = (1,..,1,0) (1D:322-326)."""
        un = np.tile(np.r_[np.ones(6), 0.0], self.nv)
        return np.zeros(self.nv * NF), un

    def ohp_summary(self, u):
        """field_OHP [V/nm] and eps_rel_OHP as reference 1D:802-805,893-954 derive them from the last state."""
        lib = load("double")
        last = np.ascontiguousarray(u).reshape(self.nv, NF)
        p = np.ascontiguousarray(last[:, 6])
        field = np.zeros(self.nv)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.edl_project_neg_gradient(self.nv, self.x.ctypes.data_as(dp), p.ctypes.data_as(dp), field.ctypes.data_as(dp))
        field_values_rescaled = field * self.thermal_voltage / self.L_n
        c_cat = last[0, 5] * self.initial_conc[self.cat_str]
        c_H = last[0, 0] * self.initial_conc["H"]
        w = (self.n_water[self.cat_str] * c_cat + self.n_water["H"] * c_H) * 1.0e-3
        return {"field_OHP": float(field_values_rescaled[0] * 1.0e-9), "eps_rel_OHP": float(self.eps_rel * ((55 - w) / 55) + 6 * (w / 55))}


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def residual_jacobian(setup, u, un, want_jacobian=True, kind="double"):
    lib = load(kind)
    n = setup.nv * NF
    F = np.zeros(n)
    J = np.zeros((n, BAND)) if want_jacobian else None
    u, un = np.ascontiguousarray(u, dtype=float), np.ascontiguousarray(un, dtype=float)
    lib.edl_residual_jacobian(ctypes.byref(setup.params), _dp(setup.x), _dp(u), _dp(un), _dp(F), _dp(J) if want_jacobian else None)
    return F, J


def newton_solve(setup, u, un, kind="double"):
    """One solve(F == 0, u, bcs); returns (u_new, iterations (negative: not converged), residual history)."""
    lib = load(kind)
    u = np.array(u, dtype=float)
    un = np.ascontiguousarray(un, dtype=float)
    res = np.zeros(setup.params.max_it + 1)
    rc = lib.edl_newton_solve(ctypes.byref(setup.params), _dp(setup.x), _dp(u), _dp(un), _dp(res))
    its = rc if rc >= 0 else -rc - 1
    return u, rc, res[: its + 1]


def run(setup, nsteps, u, un, kind="double"):
    """nsteps time steps; returns (u, un, its array, completed steps, residual history of the last solve)."""
    lib = load(kind)
    u, un = np.array(u, dtype=float), np.array(un, dtype=float)
    its = np.zeros(nsteps, dtype=np.int32)
    res = np.zeros(setup.params.max_it + 1)
    done = lib.edl_run(ctypes.byref(setup.params), _dp(setup.x), nsteps, _dp(u), _dp(un), its.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(res))
    return u, un, its, done, res
