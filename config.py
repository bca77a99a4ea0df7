"""Drop-in module name of the reference (`config.py`): re-exports the build's implementation."""
from mllp_amd.config import AttrDict, HOT_PATH_DEFAULTS, _merge_a_into_b, cfg_from_file, load_config  # noqa: F401
