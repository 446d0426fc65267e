"""xflow_amd — MI355X-native sparse LR/FM trainer behind xflow's worker/server surface.

The product is `lib/libxflow_amd.so` (hand-written HIP kernels for gfx950 + host C++,
C ABI in include/xflow_amd.h).  This package is the thin Python binding (`capi`) and the
multi-GPU driver (`sharded`) that moves keys / weights / gradients between GPU shards with
torch.distributed (RCCL) — plumbing, no math.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
