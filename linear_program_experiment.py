#!/usr/bin/env python3
"""python linear_program_experiment.py --cfg linear_program_netlib.yaml  (same CLI as the reference)"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch loads the HIP runtime (mllp_amd/__init__.py)

from mllp_amd.experiment import main

if __name__ == "__main__":
    sys.exit(main())
