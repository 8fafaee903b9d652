"""The compiled binding of the C ABI (Cython): ``Wavelets`` with the method set of the reference's class, src/pypwt.pyx:64-615.
Built by ``python -m pypwt_amd.build`` from INTEGRATION.md's ``cdef extern`` block and wavelets_class.pyx.in; importing this
package raises ImportError when the extension has not been built (the ctypes class of pypwt_amd.wavelets is always there)."""
from ._wavelets import PdwtError, Wavelets, device_count  # noqa: F401
