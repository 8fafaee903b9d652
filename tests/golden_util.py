"""Helpers shared by the oracle tests and the GPU parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def load_digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        return json.load(f)


def ndim_of(kind):
    return 2 if kind.endswith("2") else 1


def swt_of(kind):
    return 1 if "swt" in kind else 0


def band_tol(level_index_1based, scale=255.0, base=3e-4):
    """The reference's absolute rule (test/test_wavelets.py:103,235,247):
    3e-4 * 2^level on 0..255-range data."""
    return base * (2 ** level_index_1based) * (scale / 255.0)


def rel_err(a, b):
    """max|a-b| / max|b|  -- the north-star criterion (1e-4 relative per band)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / den


def reconstruction_tol(x, wname, levels, ndim=2, do_swt=0, ora=None):
    """The bound on |inverse(forward(x)) - x| the GPU tests use: the reference's 7e-4 (test/test_wavelets.py:545, stated on its
    0..255 test image), or twice what the reference's own fp32 arithmetic -- the CPU oracle on the same input -- achieves where that
    is more (deep plans on 0..255 data, 40-tap banks, ill-conditioned biorthogonal banks).  One rule for every test: no per-test
    constant.  `ora`: the oracle's forward coefficients of x, when the caller has them already."""
    import numpy as np
    from oracle import oracle
    if ora is None:
        ora = oracle.forward(x, wname, levels, ndim=ndim, do_swt=do_swt)
    own = float(np.abs(oracle.inverse(ora, x.shape, wname, levels, ndim=ndim, do_swt=do_swt) - x).max())
    return max(7e-4, 2.0 * own)
