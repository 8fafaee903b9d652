"""pycudwt -- the import name of the reference package (README.md:60-81: ``from pycudwt import Wavelets``), served by the
MI355X-native library of pypwt_amd.

``Wavelets`` is the compiled (Cython) binding when it has been built, the ctypes binding otherwise; the environment variable
PYPWT_AMD_BINDING = "cython" | "ctypes" forces one (and fails loudly when it is not available).  Both are the same class to a
user of the reference: constructor, attributes, methods, coefficient layout and state rules of src/pypwt.pyx:64-615.  There is
no CPU fallback: without the HIP library, or without a GPU, construction fails."""
import os as _os


def _pick():
    want = _os.environ.get("PYPWT_AMD_BINDING", "").lower()
    if want not in ("", "cython", "ctypes"):
        raise ImportError("PYPWT_AMD_BINDING must be 'cython' or 'ctypes', not %r" % want)
    if want != "ctypes":
        try:
            from pypwt_amd._cy import Wavelets as W
            return W, "cython"
        except ImportError:
            if want == "cython":
                raise
    from pypwt_amd.wavelets import Wavelets as W
    return W, "ctypes"


Wavelets, binding = _pick()
from pypwt_amd import __version__  # noqa: E402,F401

version = Wavelets.version
__all__ = ["Wavelets", "binding", "version"]
