"""Tetrahedral meshes of the pore cylinder.

The reference reads DOLFIN-XML meshes "generated using a separate script" that was never published
(3D/MPNP_CO2ER_pore.py:20); seven of the meshes it names are not in the release either (``.MISSING_LARGE_BLOBS``:
L_100_R_10/20/50, L_10_R_10/50, L_20_R_5, L_25_R_20).  This module generates meshes of the same kind — the cylinder
z in [0, 1], radius R/L, quasi-uniform tetrahedra, coordinates already scaled by L as in the shipped files — so that those
cases (BASELINE configs[3] = L_100_R_50 among them) can run:

* the disc is triangulated by concentric rings (ring k carries 6k vertices: near-equilateral triangles of edge R/(L n_rings));
* the triangulation is extruded in n_layers layers and every prism is cut into three tetrahedra with the diagonal of each
  quadrilateral face running from the lower-numbered bottom vertex to the higher-numbered top vertex, which makes the cut
  conforming across neighbouring prisms without any search;
* the densities default to what the shipped meshes have (App. E: edges 0.012-0.049 at aspect 0.1) and are raised where the
  reference's wall test needs it: a lateral facet is marked as ds(2) only if its three vertices AND its centroid satisfy
  |x^2 + y^2 - (R/L)^2| < tol (3D:350-356), i.e. the outer ring must be fine enough for the chord error to stay below tol.

    python -m gmpnp_amd.meshgen --L 100e-9 --R 50e-9            # writes data/utilities/L_100_R_50.xml.gz
    python -m gmpnp_amd.meshgen --missing                        # the files of .MISSING_LARGE_BLOBS (six are shipped in
                                                                 # data/utilities; L_10_R_50 — 0.86 M vertices, 63 MB — is not)

Generated meshes are not the author's meshes: results on them are comparable with the reference at discretisation level only.
"""
from __future__ import annotations

import argparse
import math
import os

import numpy as np

from .mesh import Mesh, write_dolfin_xml

MISSING = [(100, 10), (100, 20), (100, 50), (10, 10), (10, 50), (20, 5), (25, 20)]   # (L nm, R nm) of .MISSING_LARGE_BLOBS


def disc_triangulation(radius: float, n_rings: int):
    """Vertices (n, 2) and triangles (m, 3) of the ring triangulation of a disc; the last 6 n_rings vertices are the rim."""
    pts = [np.zeros((1, 2))]
    start = [0, 1]
    for k in range(1, n_rings + 1):
        ang = 2.0 * math.pi * np.arange(6 * k) / (6 * k)
        pts.append(radius * k / n_rings * np.stack([np.cos(ang), np.sin(ang)], axis=1))
        start.append(start[-1] + 6 * k)
    tris = []
    for k in range(1, n_rings + 1):
        no, ni = 6 * k, max(6 * (k - 1), 1)
        o0, i0 = start[k], start[k - 1]
        for sct in range(6):
            for j in range(k):
                a = o0 + (sct * k + j) % no
                b = o0 + (sct * k + j + 1) % no
                c = i0 + (sct * (k - 1) + j) % ni if k > 1 else 0
                tris.append((a, b, c))
                if j < k - 1:
                    d = i0 + (sct * (k - 1) + j + 1) % ni
                    tris.append((c, b, d))
    return np.concatenate(pts), np.array(tris, dtype=np.int64)


def cylinder_mesh(aspect: float, n_rings: int, n_layers: int) -> Mesh:
    """Tetrahedral mesh of {x^2 + y^2 <= aspect^2, 0 <= z <= 1}."""
    p2, tri = disc_triangulation(aspect, n_rings)
    npl = p2.shape[0]
    z = np.linspace(0.0, 1.0, n_layers + 1)
    z[0], z[-1] = 0.0, 1.0   # the reference marks the caps with near(x[2], 0 | 1, 1e-12)
    coords = np.concatenate([np.column_stack([p2, np.full(npl, zk)]) for zk in z])
    ts = np.sort(tri, axis=1)   # p0 < p1 < p2: diagonals run from the lower-numbered bottom to the higher-numbered top vertex
    cells = []
    for k in range(n_layers):
        b, t = ts + k * npl, ts + (k + 1) * npl
        cells.append(np.stack([b[:, 0], b[:, 1], b[:, 2], t[:, 2]], axis=1))
        cells.append(np.stack([b[:, 0], b[:, 1], t[:, 1], t[:, 2]], axis=1))
        cells.append(np.stack([b[:, 0], t[:, 0], t[:, 1], t[:, 2]], axis=1))
    cells = np.concatenate(cells)
    X = coords[cells]
    det = np.linalg.det(X[:, 1:, :] - X[:, :1, :])
    flip = det < 0
    cells[flip, 0], cells[flip, 1] = cells[flip, 1].copy(), cells[flip, 0].copy()   # positive orientation, as the shipped meshes
    return Mesh(dim=3, coords=coords, cells=cells.astype(np.int32))


def wall_tolerance(L: float, R: float) -> float:
    """The reference's geometric tolerance of the wall test (3D:350-355)."""
    return 5.0e-3 if (R == 5.0e-9 or R == 50.0e-9) and L == 10.0e-9 else 1.0e-3


def default_density(L: float, R: float, h: float = None):
    """(n_rings, n_layers): edge length like the shipped meshes (about 0.025 at aspect 0.1, 0.09 at aspect 0.5), refined
    until the centroid of a lateral facet passes the reference's wall test with a safety factor of 2."""
    aspect = R / L
    if h is None:
        h = min(0.09, max(0.02, 0.25 * aspect))
    n_rings = max(2, int(round(aspect / h)))
    tol = wall_tolerance(L, R)
    # centroid of a lateral facet: r^2 (5 + 4 cos(dtheta)) / 9  ->  deficit r^2 (4/9) (1 - cos dtheta), dtheta = 2 pi / (6 n_rings)
    while aspect ** 2 * (4.0 / 9.0) * (1.0 - math.cos(2.0 * math.pi / (6 * n_rings))) > 0.5 * tol:
        n_rings += 1
    n_layers = max(4, int(round(1.0 / (aspect / n_rings))))
    n_layers = min(n_layers, 64)
    return n_rings, n_layers


def pore_mesh(L: float, R: float, h: float = None) -> Mesh:
    n_rings, n_layers = default_density(L, R, h)
    return cylinder_mesh(R / L, n_rings, n_layers)


def mesh_filename(L: float, R: float) -> str:
    """The reference's file name (3D:330-331, int() truncation and all)."""
    return "L_" + str(int(L * 1e+9)) + "_R_" + str(int(R * 1e+9)) + ".xml"


def main(argv=None):
    from .params import utilities_dir
    p = argparse.ArgumentParser(description="generate pore cylinder meshes (DOLFIN-XML, gzip)")
    p.add_argument("--L", type=float, default=None)
    p.add_argument("--R", type=float, default=None)
    p.add_argument("--h", type=float, default=None, help="target edge length in units of L")
    p.add_argument("--missing", action="store_true", help="generate the seven meshes of the reference's .MISSING_LARGE_BLOBS")
    p.add_argument("--out_dir", type=str, default=None)
    a = p.parse_args(argv)
    out_dir = a.out_dir or utilities_dir()
    todo = [(l * 1e-9, r * 1e-9) for l, r in MISSING] if a.missing else [(a.L, a.R)]
    for L, R in todo:
        if L is None or R is None:
            p.error("--L and --R (or --missing)")
        m = pore_mesh(L, R, a.h)
        path = os.path.join(out_dir, mesh_filename(L, R) + ".gz")
        write_dolfin_xml(m, path)
        print("%s: %d vertices, %d cells (rings %d, layers %d)" % ((path, m.num_vertices, m.num_cells) + default_density(L, R, a.h)))


if __name__ == "__main__":
    main()
