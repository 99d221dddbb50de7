"""nmfk.jl_amd -- MI355X-native implementation of the NMFk.jl `execute(...; method=:simple)` hot path.

The directory name contains a dot, so it is imported through the `nmfk_jl_amd` shim at the repository root:

    import nmfk_jl_amd as NMFk
    W, H, fit, robustness, aic, kopt = NMFk.execute(X, range(2, 6), save=False, load=False)
"""
from ._lib import (COMPUTE_F32, COMPUTE_F64, STOP_CONSISTENCY, STOP_MAXITER, STOP_STAGNATION, STOP_TOL, Comm, Context,
                   Multi, NMFkError, build, default_params, device_count, lib)
from .execute import ExecuteOptions, execute, execute_run, getk, input_checks, run_seed, signalorder
from . import parallel
from .cluster import robustkmeans, sortclustering

__all__ = ["ExecuteOptions", "robustkmeans", "sortclustering", "execute", "execute_run", "getk", "signalorder", "input_checks", "run_seed", "Context", "Comm", "Multi", "NMFkError",
           "build", "lib", "device_count", "default_params", "parallel"]
