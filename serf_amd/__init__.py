"""serf_amd — MI355X-native bulk SWIM/Serf gossip simulator (hot path of al8n/serf).

The compute lives in ``serf_amd/csrc`` (hand-written HIP for gfx950 behind the C ABI of
``include/serf_sim.h``).  This package is the thin host side: a ctypes binding of that ABI with
``Serf``-shaped method names (serf-core/src/serf/api.rs) and the torch.distributed exchange for
sharded runs.  There is no CPU fallback: ``load()`` raises if the HIP library has not been built.
"""
import os

from . import _ffi
from ._ffi import (Config, Sim, SimError, SimLib, Stats, make_config)  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SERF_SIM_LIB: another build of the same HIP source (tools/ab.py variants, -DTICK_TIMING ...), for measurements
LIB_PATH = os.environ.get("SERF_SIM_LIB") or os.path.join(_HERE, "csrc", "libserf_sim.so")
_lib = None


def load() -> SimLib:
    """Load the HIP product library (fails loudly when it is missing)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: torch bundles its own libamdhip64; when torch is going to be
        # used at all (device buffers for the sharded exchange, streams) it has to be loaded first so
        # that this library binds to the same runtime instead of bringing in a second one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = SimLib(LIB_PATH, prefix="sim_")
    return _lib


def create(n_nodes, **kw) -> Sim:
    """Serf::new for a whole simulated cluster on the current GPU."""
    return Sim(load(), make_config(n_nodes, **kw))
