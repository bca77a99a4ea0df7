"""Drop-in module name of the reference (`linear_program_data.py`): Netlib loaders of this build."""
from mllp_amd.data import (LPInstance, SUBSET5, get_netlib_dataset, get_netlib_dataset_dense,  # noqa: F401
                           load_instances, load_packed, synthetic_instance)
