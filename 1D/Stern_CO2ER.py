#!/usr/bin/env python3
"""Entry point with the reference's script name: python 1D/Stern_CO2ER.py [--model Stern_linear] [--from_run <1D output directory>]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.stern import main  # noqa: E402

if __name__ == "__main__":
    main()
