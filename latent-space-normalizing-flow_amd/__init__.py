"""MI355X-native flow prior (drop-in for the reference's `model._netF`).

    from lsnf_amd import _netF          # same constructor / forward / state_dict as the reference

The compute path is liblsnf_flow.so (hand-written HIP, gfx950); see include/lsnf_flow.h."""
from ._lib import LsnfError, LIB_PATH, exported_symbols, load as load_library  # noqa: F401
from . import flow, parallel, langevin, netg  # noqa: F401
from .flow import FlowPlan, prepare, forward, reverse, backward_z, backward_params, langevin_step, params_from_state_dict  # noqa: F401
from .netf import _netF  # noqa: F401
from .netg import _netG  # noqa: F401

__all__ = ["LsnfError", "LIB_PATH", "load_library", "flow", "FlowPlan", "prepare", "forward", "reverse",
           "backward_z", "backward_params", "params_from_state_dict", "_netF", "_netG"]
