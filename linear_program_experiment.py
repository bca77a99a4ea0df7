#!/usr/bin/env python3
"""python linear_program_experiment.py --cfg linear_program_netlib.yaml  (same CLI as the reference)"""
import sys

from mllp_amd.experiment import main

if __name__ == "__main__":
    sys.exit(main())
