#!/usr/bin/env python3
"""Reaction-diffusion model of the planar CO2ER electrode on the MI355X backend: the reference's script name, flags and
outputs (see gmpnp_amd/rxndiff1d.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.rxndiff1d import main  # noqa: E402

if __name__ == "__main__":
    print(main())
