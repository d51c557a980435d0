"""Minimal VTK writer for ``File("solution_X.pvd") << function`` (reference 3D:863-880): one ASCII .vtu with the
P1 point data plus the .pvd collection that points at it."""
from __future__ import annotations

import os

import numpy as np


def write_pvd(path, coords, cells, values, name="f"):
    base = os.path.splitext(path)[0]
    vtu = base + "000000.vtu"
    nv, d = coords.shape
    nc, nn = cells.shape
    pts = np.zeros((nv, 3))
    pts[:, :d] = coords
    ctype = {2: 3, 4: 10}[nn]  # VTK_LINE / VTK_TETRA
    with open(vtu, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1">\n<UnstructuredGrid>\n')
        fh.write('<Piece NumberOfPoints="%d" NumberOfCells="%d">\n' % (nv, nc))
        fh.write('<Points>\n<DataArray type="Float64" NumberOfComponents="3" format="ascii">')
        fh.write(" ".join(repr(float(x)) for x in pts.ravel()))
        fh.write('</DataArray>\n</Points>\n<Cells>\n<DataArray type="UInt32" Name="connectivity" format="ascii">')
        fh.write(" ".join(str(int(v)) for v in cells.ravel()))
        fh.write('</DataArray>\n<DataArray type="UInt32" Name="offsets" format="ascii">')
        fh.write(" ".join(str(nn * (i + 1)) for i in range(nc)))
        fh.write('</DataArray>\n<DataArray type="UInt8" Name="types" format="ascii">')
        fh.write(" ".join([str(ctype)] * nc))
        fh.write('</DataArray>\n</Cells>\n<PointData Scalars="%s">\n<DataArray type="Float64" Name="%s" format="ascii">' % (name, name))
        fh.write(" ".join(repr(float(v)) for v in values))
        fh.write("</DataArray>\n</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
    with open(path, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n')
        fh.write('<DataSet timestep="0" part="0" file="%s" />\n</Collection>\n</VTKFile>\n' % os.path.basename(vtu))
