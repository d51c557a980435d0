#!/usr/bin/env python3
"""Entry point with the reference's script name: python 3D/MPNP_CO2ER_pore.py --L=50e-9 --R=5e-9 ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.pore3d import main  # noqa: E402

if __name__ == "__main__":
    main()
