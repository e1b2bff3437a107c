"""linearmpc.jl_amd -- MI355X-native batched QP backend for LinearMPC.jl's online path.

The directory name carries a dot, so import it through the repo-root alias:

    import linearmpc_jl_amd as lmpc

Contents: the HIP kernels + C ABI (csrc/, built to lib/liblmpc_hip.so), the ctypes binding
(_cabi), `BatchedQP` (solver), the host-side mirror of the reference interface (mpc) and the
one-process-per-GPU sharding helpers (shard).  Importing the package does not need a GPU;
setting up or solving does, and there is no CPU fallback.
"""
from ._cabi import LIB_PATH, SYMBOLS, LmpcError, Settings, default_settings, default_settings_f32, lib  # noqa: F401
from .solver import BatchedQP, MultiQP, transform, transform_avi  # noqa: F401
from .mpc import MPC, MPQP, GeneratedController  # noqa: F401
from .shard import gather_shards, shard_bounds, shard_counts, solve_sharded  # noqa: F401
from . import explicit  # noqa: F401

__all__ = ["BatchedQP", "MultiQP", "transform", "transform_avi", "MPC", "MPQP", "GeneratedController", "Settings", "default_settings", "default_settings_f32", "LmpcError",
           "gather_shards", "shard_bounds", "shard_counts", "solve_sharded", "explicit", "lib", "LIB_PATH",
           "SYMBOLS"]
