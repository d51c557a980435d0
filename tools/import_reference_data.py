#!/usr/bin/env python3
"""Import the reference's INPUT DATA (no code) into data/utilities/.

The drivers read the same inputs the reference reads from its ``utilities/`` folder
(reference 3D/MPNP_CO2ER_pore.py:123,224-226,329-332; 1D/MPNP_CO2ER_EDL.py:89,146-148,231-234):
YAML parameter tables and DOLFIN-XML meshes.  /root/reference does not exist on the GPU box, so
the data travels with the repo:

* YAML: values only, re-serialised through ``yaml.safe_load`` -> ``yaml.safe_dump`` (keys and
  numbers unchanged; the author's comments are not carried over).
* meshes: parsed with gmpnp_amd.mesh.read_dolfin_xml and re-written by write_dolfin_xml
  (same DOLFIN-XML dialect, ``repr`` floats = bit-exact coordinates, gzip-compressed).
  A round-trip check (coords and cells bitwise equal) runs for every file.

Run here (container with /root/reference):  python tools/import_reference_data.py
"""
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gmpnp_amd.mesh import read_dolfin_xml, write_dolfin_xml  # noqa: E402

SRC = os.environ.get("GMPNP_REFERENCE_UTILITIES", "/root/reference/utilities")
DST = os.path.join(ROOT, "data", "utilities")


def main():
    os.makedirs(DST, exist_ok=True)
    for name in sorted(os.listdir(SRC)):
        src = os.path.join(SRC, name)
        if name.endswith(".yaml"):
            with open(src) as fh:
                data = yaml.safe_load(fh)
            with open(os.path.join(DST, name), "w") as fh:
                yaml.safe_dump(data, fh, default_flow_style=False, sort_keys=False)
            with open(os.path.join(DST, name)) as fh:
                assert yaml.safe_load(fh) == data, name
            print("yaml ", name)
        elif name.endswith(".xml") or name.endswith(".xml.gz"):
            out = name if name.endswith(".gz") else name + ".gz"
            mesh = read_dolfin_xml(src)
            write_dolfin_xml(mesh, os.path.join(DST, out))
            back = read_dolfin_xml(os.path.join(DST, out))
            assert np.array_equal(back.coords, mesh.coords) and np.array_equal(back.cells, mesh.cells), name
            print("mesh ", out, mesh.num_vertices, mesh.num_cells)


if __name__ == "__main__":
    main()
