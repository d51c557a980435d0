#!/usr/bin/env python3
"""Entry point with the reference's script name: python 1D/MPNP_CO2ER_EDL.py --voltage_multiplier=-10.0 --cation='Cs'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.edl1d import main  # noqa: E402

if __name__ == "__main__":
    main()
