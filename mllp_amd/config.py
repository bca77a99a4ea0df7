"""Experiment configuration: `--cfg <yaml>` -> attribute-style nested dict.

Mirrors the surface of the reference's `config.py` (reference `config.py:6-25`):
`load_config()` parses `--cfg/--config`, raises ``ValueError('Please specify path
to the configuration file!')`` when absent, and returns the yaml merged into an
attribute dict.  The reference depends on `easydict`; this build carries its own
small `AttrDict` so nothing outside the standard library + pyyaml is needed.

Keys read on the hot path (reference `linear_program_netlib.yaml:1-15`):
``train_data_type, train_lr, train_iter, methods, verbose``.  Optional keys
added by this build (all default to reference behaviour when absent) are listed
in `HOT_PATH_DEFAULTS`.
"""
import argparse

import yaml

# optional keys this build understands; absent -> reference semantics
HOT_PATH_DEFAULTS = {
    "device": "cuda",        # the reference hard-wires cpu (experiment.py:18); the product path is HIP
    "batch_size": 1,         # instances per Adam step; 1 == reference (experiment.py:123-144); 0 == whole dataset
    "instances": None,       # optional list of instance names (e.g. ['afiro.mps', ...]); None == all
    "log_every": 1,          # print per-instance metrics every k epochs
    "save_every": 0,         # extra checkpoints every k epochs (0 == only at the end, as the reference)
    "resume": False,         # resume from linear_program_<data>_<method>.ckpt if present
    "tiled_copies": "auto",  # LDS-tiled copies of the batch for the large-batch kernels: True | False | 'auto' (>= 32 M nonzeros)
    "use_hip_graph": "auto", # capture the training step into a hipGraph: True | False | 'auto' (= False: eager launches measured faster)
    "angle_feat_dim": 256,   # feat_dim of AngleModel for the 'angleNet' method (the reference hard-codes 256, experiment.py:83)
    "dtype": "f32",          # feature precision of the training step.  'f32' is the reference's arithmetic and the only one the
                             # step implements; 'bf16' exists for the plain SpMM only (LPBatch.spmm_bf16 / mllp_spmm_csr_bf16)
}


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted recursively."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            # a missing key is an AttributeError at use, as with easydict
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def get_default(self, k):
        """Value of an optional build key, falling back to HOT_PATH_DEFAULTS."""
        return self[k] if k in self else HOT_PATH_DEFAULTS[k]


def _merge_a_into_b(a, b, strict=False):
    """Merge mapping `a` into `b`, overwriting `b`'s entries (reference config.py:28-57).

    With ``strict`` every key of `a` must already exist in `b` with the same type
    (int -> float promotion allowed).
    """
    if not isinstance(a, AttrDict):
        return
    for key, val in a.items():
        if strict:
            if key not in b:
                raise KeyError("{} is not a valid config key".format(key))
            want, got = type(b[key]), type(val)
            if want is not got:
                if want is float and got is int:
                    val = float(val)
                elif key not in ("CLASS",):
                    raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(want, got, key))
        if isinstance(val, AttrDict) and isinstance(b.get(key), AttrDict):
            try:
                _merge_a_into_b(val, b[key], strict)
            except Exception:
                print("Error under config key: {}".format(key))
                raise
        else:
            b[key] = val


def cfg_from_file(filename, cfg=None):
    """Load a yaml file and merge it into `cfg` (a fresh AttrDict when None)."""
    with open(filename, "r") as fh:
        loaded = AttrDict(yaml.safe_load(fh) or {})
    if cfg is None:
        cfg = AttrDict()
    _merge_a_into_b(loaded, cfg)
    return cfg


def load_config(argv=None):
    parser = argparse.ArgumentParser(description="mllp learned-LP experiment protocol (MI355X build).")
    parser.add_argument("--cfg", "--config", dest="cfg_file", default=None, type=str,
                        help="path to the configuration file")
    args, _unknown = parser.parse_known_args(argv)
    if args.cfg_file is None:
        raise ValueError("Please specify path to the configuration file!")
    return cfg_from_file(args.cfg_file)
