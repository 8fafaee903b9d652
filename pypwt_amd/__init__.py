"""pypwt_amd -- MI355X-native (gfx950) discrete wavelet transform, drop-in for pycudwt's `Wavelets`.

    from pypwt_amd import Wavelets
    W = Wavelets(img, "db2", 3); W.forward(); W.soft_threshold(10); W.inverse(); W.coeffs; W.image

The compute path is the hand-written HIP library pypwt_amd/libpypwt_amd.so (C ABI in
include/pypwt_amd.h, built by `python -m pypwt_amd.build`; `Wavelets64` binds the fp64 build
libpypwt_amd_f64.so).  There is no CPU fallback.
"""
from .sharded import ShardedBatch, partition_images  # noqa: F401
from .wavelets import BatchedWavelets, BatchedWavelets64, DeviceArray, Wavelets, Wavelets64  # noqa: F401



def trim_pool():
    """Release the device memory and streams that destroyed plans left in the library's pool (all loaded variants)."""
    from . import _lib
    return sum(lib.pdwt_trim_pool() for lib in list(_lib._libs.values()))


__version__ = "0.1.0"
