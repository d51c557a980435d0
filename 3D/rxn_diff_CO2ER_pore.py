#!/usr/bin/env python3
"""Reaction-diffusion model of the cylindrical pore on the MI355X backend: the reference's script name, flags and outputs
(see gmpnp_amd/rxnpore3d.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.rxnpore3d import main  # noqa: E402

if __name__ == "__main__":
    print(main())
