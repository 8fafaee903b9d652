"""pypwt -- the name the reference's own tests import (test/test_wavelets.py:23: ``from pypwt import Wavelets``); the same
objects as ``pycudwt``."""
from pycudwt import Wavelets, binding, version, __version__  # noqa: F401

__all__ = ["Wavelets", "binding", "version"]
