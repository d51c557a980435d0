"""How much does the one unpinned ingredient — WHICH quadrature points the rational steric term u_i / (1 - S) is sampled at on a
tetrahedron — move a solution?  CPU oracle, first time step of the 3D pore problem (full physics, u = 0 start
as in the reference) on a generated cylinder, with
  default   5-point degree-3 rule in F, 14-point degree-4 rule in J (what FFC/FIAT are restated to use)
  deg4      the 14-point rule in both
  deg7      a 125-point conical Gauss rule (exact to degree 7) in both
Prints the Newton iteration counts and the largest difference of the converged states relative to each field's range."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import copy
import numpy as np
import gmpnp_oracle as O
from gmpnp_amd.mesh import mark_pore_boundaries
from gmpnp_amd.meshgen import cylinder_mesh
from gmpnp_amd.model import Quadrature, default_quadrature
from gmpnp_amd.params import pore_parameters
from gmpnp_amd.problem import Problem, pore_dirichlet


def conical(n):
    x, w = np.polynomial.legendre.leggauss(n)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    pts, wts = [], []
    for a, wa in zip(x, w):
        for b, wb in zip(x, w):
            for c, wc in zip(x, w):
                p = (a, b * (1 - a), c * (1 - a) * (1 - b))
                pts.append((1 - sum(p),) + p); wts.append(wa * wb * wc * (1 - a) ** 2 * (1 - b) * 6.0)
    return np.array(pts), np.array(wts)


for rings, layers in ((3, 6), (5, 10)):
    pp = pore_parameters(concentration_elec=0.5, L=10e-9, R=5e-9)
    mesh = cylinder_mesh(pp.aspect_pore, rings, layers)
    sag = pp.aspect_pore ** 2 * (1.0 - np.cos(np.pi / (6 * rings)) ** 2)
    bnd = mark_pore_boundaries(mesh, pp.aspect_pore, 1.5 * sag)
    dofs, vals = pore_dirichlet(pp, bnd)
    dq = default_quadrature(3)
    l7, w7 = conical(5)
    assert abs(w7.sum() - 1.0) < 1e-12
    rules = {"default": dq, "deg4": Quadrature(dq.lam_j, dq.w_j, dq.lam_j, dq.w_j), "deg7": Quadrature(l7, w7, l7, w7)}
    nv = mesh.num_vertices
    sols = {}
    for name, qd in rules.items():
        prob = Problem(coords=mesh.coords, cells=mesh.cells, model=pp.model, quad=qd, wall_facets=bnd.ds_facets[2], exit_facets=bnd.ds_facets[3],
                       bc_dofs=dofs, bc_vals=vals)
        u, st = O.newton_solve(prob, np.zeros(prob.ndof), np.tile(np.r_[np.ones(8), 0.0], nv), relaxation_parameter=0.9,
                               relative_tolerance=1e-12, absolute_tolerance=1e-12, maximum_iterations=60)
        sols[name] = (u.reshape(nv, 9), st.iterations)
    ref = sols["deg7"][0]
    rng = ref.max(0) - ref.min(0)
    for name in ("default", "deg4"):
        d = np.abs(sols[name][0] - ref).max(0) / rng
        print("cylinder %d rings x %d layers (%d vertices): %-8s Newton %d (deg7: %d)  max |u - u_deg7| / range per field: %s" %
              (rings, layers, nv, name, sols[name][1], sols["deg7"][1], " ".join("%.1e" % x for x in d)))
