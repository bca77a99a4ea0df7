"""Drop-in module name of the reference (`linear_program_methods.py`) for the sparse bipartite path:
set_seed, BipartiteData, build_graph_from_weights_sets, GNNModel, and for the `angleNet` method AngleModel,
build_graph_from_Q_sets, get_netlib_dataloader.  InvariantModel and the max-covering solvers are outside this build's
scope."""
import numpy as np  # noqa: F401  (the reference's `from linear_program_methods import *` exposes these)
import torch  # noqa: F401

from mllp_amd.angle import AngleModel, build_graph_from_Q_sets, get_netlib_dataloader  # noqa: F401
from mllp_amd.model import BipartiteData, GNNModel, build_graph_from_weights_sets, set_seed  # noqa: F401


def __getattr__(name):
    if name == "InvariantModel":
        raise NotImplementedError("InvariantModel belongs to the reference's 'invariant' research path, which this "
                                  "build does not implement")
    raise AttributeError(name)
