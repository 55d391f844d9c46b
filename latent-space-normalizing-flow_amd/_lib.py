"""ctypes binding of liblsnf_flow.so (C ABI: include/lsnf_flow.h).

The library is the product; there is no fallback.  If the shared object is missing or a
symbol is absent, import of the compute API raises -- nothing is silently routed elsewhere."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSNF_LIB_PATH: developer override to A/B alternative BUILDS of the same library (tools/); not a fallback.
LIB_PATH = os.environ.get("LSNF_LIB_PATH") or os.path.join(_HERE, "liblsnf_flow.so")

LSNF_PARAMS_PER_BLOCK = 12
ABI_VERSION = 5

class LsnfRng(ctypes.Structure):
    """include/lsnf_flow.h `LsnfRng`: in-kernel Philox noise of lsnf_langevin_step."""
    _fields_ = [("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("offset_dev", c_void_p), ("row0", ctypes.c_int64)]


# name -> (restype, argtypes); mirrors include/lsnf_flow.h one to one
_SIGNATURES = {
    "lsnf_abi_version": (c_int, []),
    "lsnf_last_error": (c_char_p, []),
    "lsnf_set_small_batch_max": (c_int, [c_int]),
    "lsnf_set_math_mode": (c_int, [c_int]),
    "lsnf_device_arch": (c_int, [c_int, c_char_p, c_size_t]),
    "lsnf_plan_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lsnf_prepare_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lsnf_prepare": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "lsnf_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsnf_params_fast_path": (c_int, []),
    "lsnf_act_saved_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lsnf_restash": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsnf_reverse": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsnf_backward_z": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "lsnf_langevin_step": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(LsnfRng), c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsnf_backward_params_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lsnf_backward_params": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                     c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                     c_void_p, c_void_p, c_void_p]),
}

_lib = None


class LsnfError(RuntimeError):
    """A call into liblsnf_flow.so returned a negative status."""


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """dlopen the in-tree library and attach signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LsnfError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C latent-space-normalizing-flow_amd/csrc`). There is no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.lsnf_abi_version()
    if v != ABI_VERSION:
        raise LsnfError(f"liblsnf_flow.so ABI {v} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().lsnf_last_error()
        raise LsnfError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
