"""ctypes binding of libmllp_hip.so (C ABI declared in include/mllp_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises at import of the
symbol table (`lib()`), and every launch fails loudly without a HIP device.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmllp_hip.so")
HEADER_PATH = os.path.abspath(os.path.join(_HERE, "..", "include", "mllp_hip.h"))

NUM_PARAMS = 4721
ABI_VERSION = 3


class MllpError(RuntimeError):
    pass


_PROTOTYPES = {
    # name: (restype, [argtypes])
    "mllp_last_error": (c_char_p, []),
    "mllp_abi_version": (c_int, []),
    "mllp_graph_create_host": (c_int, [c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                       POINTER(c_int32), POINTER(c_double), c_int32, c_int32,
                                       POINTER(c_void_p)]),
    "mllp_graph_create_device": (c_int, [c_int64, POINTER(c_int64), POINTER(c_int64), c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int32, c_int32, c_void_p, POINTER(c_void_p)]),
    "mllp_graph_destroy": (c_int, [c_void_p]),
    "mllp_graph_dims": (c_int, [c_void_p, POINTER(c_int64)]),
    "mllp_graph_export": (c_int, [c_void_p, c_int, c_void_p, c_int64]),
    "mllp_graph_set_path": (c_int, [c_void_p, c_int]),
    "mllp_graph_invalidate_inputs": (c_int, [c_void_p]),
    "mllp_graph_build_spmm_copy": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mllp_graph_drop_spmm_copy": (c_int, [c_void_p, c_int]),
    "mllp_graph_spmm_copy_info": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "mllp_graph_export_spmm_copy": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64]),
    "mllp_graph_build_stream_copy": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "mllp_graph_drop_stream_copy": (c_int, [c_void_p, c_int, c_int]),
    "mllp_graph_stream_copy_info": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64)]),
    "mllp_graph_export_stream_copy": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64]),
    "mllp_csr_transpose_device": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "mllp_graph_build_tiled": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mllp_graph_tiled_info": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64)]),
    "mllp_graph_export_tiled": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mllp_spmm_csr_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mllp_spmm_csr_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mllp_angle_num_params": (c_int, [c_int, POINTER(c_int64)]),
    "mllp_angle_workspace_floats": (c_int, [c_int64, c_int, POINTER(c_int64)]),
    "mllp_angle_forward": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_angle_backward": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_tiled_geometry": (c_int, [c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "mllp_graph_attach_tiled": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "mllp_tconv_workspace_floats": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64)]),
    "mllp_tconv_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_tconv_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mllp_gnn_workspace_bytes": (c_int, [c_void_p, POINTER(c_int64)]),
    "mllp_gnn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_gnn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_gnn_loss_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_gnn_train_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "mllp_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int64,
                               c_void_p]),
    "mllp_metrics_scratch_bytes": (c_int, [c_void_p, POINTER(c_int64)]),
    "mllp_topm_metrics": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_mps_read": (c_int, [c_char_p, c_int, POINTER(c_void_p)]),
    "mllp_lp_dims": (c_int, [c_void_p, POINTER(c_int64)]),
    "mllp_lp_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mllp_lp_free": (c_int, [c_void_p]),
}

_lib = None


def build(verbose=False):
    """Compile libmllp_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise MllpError("building libmllp_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library with prototypes set; raises MllpError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MllpError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(or `make -C mllp_amd/csrc`). The product path has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(L, name)           # AttributeError if the .so misses a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.mllp_abi_version() != ABI_VERSION:
            raise MllpError("libmllp_hip.so ABI version mismatch")
        _lib = L
    return _lib


MLLP_ERANGE = -4          # include/mllp_hip.h


def check(rc):
    if rc != 0:
        msg = lib().mllp_last_error()
        raise MllpError(f"libmllp_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None) as c_void_p."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def np_ptr(a, ctype):
    return a.ctypes.data_as(POINTER(ctype))


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
