"""Extra golden cases: name -> (driver, keyword arguments of the reference CLI, number of time steps).

Shared by tools/make_golden.py (which runs the CPU oracle and writes tests/golden/<name>_steps.npz) and by the GPU
parity tests.  They walk the flag surface of the two scripts: bulk concentration file, published/intended weak form,
voltage, pore length, cation (hydration numbers 1D:106-115), the PNP model (no steric term), the proton-flux controller
(--H_OHP) and the H2 Faradaic efficiency."""

EXTRA_PORE = {
    "pore10_pub_v25": (dict(concentration_elec=0.5, L=10e-9, R=5e-9, voltage_multiplier=-2.5, as_published=True), 3),
    "pore10_1M": (dict(concentration_elec=1.0, L=10e-9, R=5e-9), 2),
    "pore25_fe": (dict(concentration_elec=0.5, L=25e-9, R=5e-9, H2_FE=0.2, current_rough=1000.0), 2),
    # thinnest pore of the sweep (BASELINE configs[4]): from time step 3 on BiCGStab no longer converges and the
    # block-banded LU takes over, as MUMPS does in the reference
    "pore50_r1": (dict(concentration_elec=0.5, L=50e-9, R=1e-9), 6),
    # the other radii of the sweep (BASELINE configs[4]) at V = -1.  R = 2.5 nm loads the R = 2 nm mesh (the reference
    # builds the file name with int(): SURVEY Q4) with the R = 2.5 nm scaling and marking radius; R = 7.5 nm asks for
    # L_50_R_7.xml, which does not exist (tests assert the error).
    "pore50_r2": (dict(concentration_elec=0.5, L=50e-9, R=2e-9), 2),
    "pore50_r2p5": (dict(concentration_elec=0.5, L=50e-9, R=2.5e-9), 2),
    "pore50_r4": (dict(concentration_elec=0.5, L=50e-9, R=4e-9), 2),
    "pore50_r10": (dict(concentration_elec=0.5, L=50e-9, R=10e-9), 2),
}

EXTRA_EDL = {
    "edl1_pnp": (dict(L_n=1e-6, model="PNP", voltage_multiplier=-2.5), 4),
    "edl1_li": (dict(L_n=1e-6, cation="Li", voltage_multiplier=-2.5), 3),
    "edl5_na": (dict(L_n=5e-6, cation="Na", voltage_multiplier=-2.5), 3),
    "edl50_default": (dict(), 3),
    "edl10_hohp": (dict(L_n=10e-6, voltage_multiplier=-2.5, H_OHP=0.5, H2_FE=0.4), 4),
    # PNP + SUPG stabilisation (reference 1D:597-722, --model PNP --stabilization Y)
    "edl1_pnp_supg": (dict(L_n=1e-6, model="PNP", stabilization="Y", voltage_multiplier=-2.5), 4),
    "edl10_pnp_supg_cs": (dict(L_n=10e-6, model="PNP", stabilization="Y", voltage_multiplier=-2.5, cation="Cs"), 4),
}

# 1D reaction-diffusion driver (reference 1D/rxn_diff_planar.py) on the same backend
EXTRA_RXN1D = {
    "rxn1d_default": (dict(), 6),
    "rxn1d_10um_05M": (dict(L_n=10e-6, concentration_KHCO3=0.5, H2_FE=0.4, current_OHP_ss=50.0), 5),
}
RXN_NEWTON = dict(maximum_iterations=100, relative_tolerance=1e-6, absolute_tolerance=1e-6)  # rxn_diff_planar.py:329-333

# 3D reaction-diffusion driver (reference 3D/rxn_diff_CO2ER_pore.py) on the same backend
EXTRA_RXN3D = {
    "rxn3d_pore10": (dict(concentration_elec=0.5, L=10e-9, R=5e-9), 3),
    "rxn3d_pore10_1M": (dict(concentration_elec=1.0, L=10e-9, R=5e-9, H2_FE=0.2), 2),
}
