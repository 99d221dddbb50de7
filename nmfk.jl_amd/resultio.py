"""On-disk results of the path, in the reference's format: JLD files (HDF5) with the reference's names and keys, so
that `NMFk.load` / `postprocess` on the Julia side can read what the GPU path wrote and the other way round.

  <case>_<n>_<m>_<nk>_<nNMF>.jld       W, H, fit, robustness, aic                     src/NMFkExecute.jl:323-327
  <case>_<n>_<m>_<nk>_<nNMF>-all.jld   W, H (all restarts), Wmean, Hmean, Wvar, Hvar, Wbest, Hbest, fit,
                                       "Cluster Silhouettes", "Cluster assignments", "Cluster centroids"
                                                                                      src/NMFkExecute.jl:650-654
The HDF5 layer is nmfk.jl_amd/jldfile.py (no h5py in the image).  Files are written to a temporary name and renamed,
so a reader never sees a partial file."""
import os

import numpy as np

from . import jldfile

EXT = ".jld"


def save(filename, **variables):
    tmp = f"{filename}.tmp{os.getpid()}"
    jldfile.save(tmp, variables)
    os.replace(tmp, filename)


def load(filename, *names):
    return jldfile.load(filename, *names)


def as_julia(v, dtype=np.float32):
    """what the reference's variable would be: Matrix{T} for arrays, T for scalars (fit values are T = eltype(X))"""
    return np.asarray(v, dtype=dtype)
