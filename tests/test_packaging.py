"""The drop-in as a package (missing 3): `pip install .` gives `from pycudwt import Wavelets` (the reference's
README, README.md:60-81) and `from pypwt import Wavelets` (the reference's tests, test/test_wavelets.py:23); the class is the
compiled Cython binding built from INTEGRATION.md's `cdef extern` block when the extension is there, the ctypes class otherwise.
CPU: the build, the import names, the method set, loud failure without a GPU, an install into a clean directory.  GPU: a script
in the shape of the reference's denoising example (doc/denoising.rst:85-141) through BOTH bindings against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the public surface of the reference's class (src/pypwt.pyx:93-615): attributes, then methods / properties
REF_ATTRS = ["Nr", "Nc", "sizes", "wname", "levels", "do_cycle_spinning", "do_swt", "do_separable", "ndim", "batched1d"]
REF_METHODS = ["info", "coeff_only", "coeffs", "image", "set_image", "forward", "inverse", "soft_threshold", "hard_threshold", "shrink",
               "norm1", "norm2sq", "add_wavelet", "set_coeff", "set_wavelets_filters", "image_int_ptr", "coeff_int_ptr", "cleanup",
               "version", "div2", "_checkarray", "_compute_sizes", "__repr__", "__str__"]


@pytest.fixture(scope="module")
def built():
    from pypwt_amd import build
    build.build_library(verbose=False)
    so = build.build_cython(verbose=False)
    assert so and os.path.exists(so)
    return so


def _run(code, env=None, cwd=ROOT, path=None):
    e = dict(os.environ)
    e.update(env or {})
    if path is not None:
        e["PYTHONPATH"] = path
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=cwd, timeout=600)


def test_the_extension_is_built_from_the_document(built):
    from pypwt_amd import build
    block = build.cython_extern_block()
    assert 'cdef extern from "pypwt_amd.h"' in block and "pdwt_soft_threshold_norms_async" in block
    assert open(os.path.join(ROOT, "pypwt_amd", "_cy", "pdwt_decl.pxi")).read() == block  # the copy an sdist builds from
    gen = open(os.path.join(ROOT, "build", "obj", "cy", "_wavelets.pyx")).read()
    assert block in gen and "cdef class Wavelets:" in gen


def test_import_names_and_binding_choice(built):
    r = _run("import pycudwt, pypwt; print(pycudwt.binding, pypwt.Wavelets is pycudwt.Wavelets, pycudwt.Wavelets.__module__)")
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[:2] == ["cython", "True"], r.stdout
    r = _run("import pycudwt; print(pycudwt.binding, pycudwt.Wavelets.__module__)", env={"PYPWT_AMD_BINDING": "ctypes"})
    assert r.returncode == 0 and r.stdout.split() == ["ctypes", "pypwt_amd.wavelets"], (r.stdout, r.stderr)
    r = _run("import pycudwt", env={"PYPWT_AMD_BINDING": "pybind"})
    assert r.returncode != 0 and "PYPWT_AMD_BINDING" in r.stderr


def test_both_classes_have_the_reference_surface(built):
    from pypwt_amd._cy import Wavelets as Cy
    from pypwt_amd.wavelets import Wavelets as Ct
    for name in REF_METHODS:
        assert hasattr(Cy, name), ("cython", name)
        assert hasattr(Ct, name), ("ctypes", name)
    for name in REF_ATTRS:  # cdef readonly attributes are descriptors on the extension type
        assert hasattr(Cy, name), ("cython", name)
    assert Cy.div2(5) == 3 and Ct.div2(5) == 3
    assert "pycudwt 1.0.3" in Cy.version()


def test_no_gpu_means_loud_failure_through_both_bindings(built):
    from pypwt_amd._cy import Wavelets as Cy, device_count
    if device_count() > 0:
        pytest.skip("a GPU is present")
    from pypwt_amd.wavelets import Wavelets as Ct
    img = np.zeros((32, 32), dtype=np.float32)
    for cls in (Cy, Ct):
        with pytest.raises(RuntimeError, match="no HIP device"):
            cls(img, "db2", 2)
        with pytest.raises((ValueError, RuntimeError)):
            cls(img, "no-such-wavelet", 2)
    with pytest.raises(NotImplementedError):
        Cy(np.zeros((2, 2, 2), dtype=np.float32), "haar", 1)


def test_pip_install_into_a_clean_directory(built, tmp_path):
    """`pip install --no-build-isolation .` (no venv module in this image: --target is the clean site); the installed tree alone --
    the repository is not on the path -- imports under the reference's names with the compiled binding and both HIP libraries."""
    target = str(tmp_path / "site")
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--quiet", "--target", target, ROOT],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    code = ("import sys, os; sys.path = [p for p in sys.path if os.path.abspath(p or '.') != %r]\n"
            "import pycudwt, pypwt, pypwt_amd\n"
            "d = os.path.dirname(pypwt_amd.__file__)\n"
            "print(pycudwt.binding, d, os.path.exists(os.path.join(d, 'libpypwt_amd.so')), os.path.exists(os.path.join(d, 'libpypwt_amd_f64.so')))\n"
            "import numpy as np\n"
            "try:\n    pycudwt.Wavelets(np.zeros((8, 8), np.float32), 'haar', 1); print('created')\n"
            "except RuntimeError as e:\n    print('refused:', e)\n") % ROOT
    r = _run(code, cwd=str(tmp_path), path=target)
    assert r.returncode == 0, r.stderr
    first = r.stdout.splitlines()[0].split()
    assert first[0] == "cython" and first[1].startswith(target) and first[2:] == ["True", "True"], r.stdout
    assert "created" in r.stdout or "no HIP device" in r.stdout


DENOISE = r'''
import sys
import numpy as np
from pycudwt import Wavelets, binding
sys.path.insert(0, %r)
from oracle import oracle

# the shape of doc/denoising.rst:85-141: a noisy image, an undecimated db2 transform, three levels, soft threshold, inverse
img = oracle.hash_input((256, 320), 9, 255.0)
noisy = (img + 20.0 * (oracle.hash_input((256, 320), 10, 2.0) - 1.0)).astype(np.float32)
W = Wavelets(noisy, "db2", 3, do_swt=1)
W.forward()
ref = oracle.forward(noisy, "db2", 3, do_swt=1)
flat = [W.coeffs[0]] + [b for lvl in W.coeffs[1:] for b in lvl]
assert all(np.abs(g - r).max() <= 2e-5 * 255 for g, r in zip(flat, ref)), "forward"
W.soft_threshold(15.0, 0, 1)
thr = oracle.threshold(ref, noisy.shape, 3, "soft", 15.0, 0, 1, do_swt=1)
W.inverse()
want = oracle.inverse(thr, noisy.shape, "db2", 3, do_swt=1)
err = float(np.abs(W.image - want).max())
assert err <= 2e-5 * 255, err
assert W.norm1 is not None and W.levels == 3 and W.do_swt == 1 and W.sizes[0] == (256, 320)
print(binding, "%%.3e" %% err, "%%.6f" %% float(W.image.mean()))
'''


@pytest.mark.gpu
def test_denoising_script_through_both_bindings_against_the_oracle(built):
    outs = {}
    for want in ("cython", "ctypes"):
        r = _run(DENOISE % ROOT, env={"PYPWT_AMD_BINDING": want})
        assert r.returncode == 0, (want, r.stderr[-3000:])
        got = r.stdout.split()
        assert got[0] == want, r.stdout
        outs[want] = got[2]
    assert outs["cython"] == outs["ctypes"]  # the same library underneath: the same image


@pytest.mark.gpu
def test_cython_class_against_the_ctypes_class_on_the_gpu(built):
    """Every method of the compiled class once, results equal to the ctypes class's (same library, same calls)."""
    from oracle import oracle
    from pypwt_amd._cy import Wavelets as Cy
    from pypwt_amd.wavelets import Wavelets as Ct
    x = oracle.hash_input((192, 160), 77, 255.0)
    a, b = Cy(x, "db3", 3), Ct(x, "db3", 3)
    assert (a.Nr, a.Nc, a.levels, a.hlen, a.sizes, a.shape) == (b.Nr, b.Nc, b.levels, b.hlen, b.sizes, b.shape)
    a.forward(); b.forward()
    for g, h in zip([a.coeffs[0]] + [c for l in a.coeffs[1:] for c in l], [b.coeffs[0]] + [c for l in b.coeffs[1:] for c in l]):
        assert np.array_equal(g, h)
    assert np.array_equal(a.coeff_only(5), b.coeff_only(5))
    assert a.norm1() == b.norm1() and a.norm2sq() == b.norm2sq()
    a.soft_threshold_norms(3.0); b.soft_threshold_norms(3.0)
    assert a.read_norms() == b.read_norms()
    a.hard_threshold(1.0, 1, 1); b.hard_threshold(1.0, 1, 1)
    a.shrink(0.1); b.shrink(0.1)
    a.proj_linf(500.0); b.proj_linf(500.0)
    band = np.ascontiguousarray(b.coeff_only(2)) * 0.5
    a.set_coeff(band, 2, check=True); b.set_coeff(band, 2, check=True)
    c2 = Cy(x, "db3", 3); c2.forward()
    d2 = Ct(x, "db3", 3); d2.forward()
    assert a.add_wavelet(c2, 0.25) == 0 and b.add_wavelet(d2, 0.25) == 0
    a.inverse(); b.inverse()
    assert np.array_equal(a.image, b.image)
    a.inverse()  # second call in a row: a warning, nothing else (wt.cu:272-279)
    assert a.image_int_ptr() != 0 and a.coeff_int_ptr(1) != 0
    rng = np.random.default_rng(2)
    f = [rng.standard_normal(6).astype(np.float32) for _ in range(4)]
    a.set_wavelets_filters("mine", *f); b.set_wavelets_filters("mine", *f)
    a.forward(x); b.forward(x)
    assert np.array_equal(a.coeffs[0], b.coeffs[0]) and a.wname == "mine" and a.hlen == 6
    s = Cy(x[0], "haar", 2, ndim=1)
    s.forward()
    assert s.coeffs[0].shape == (1, 40) and s.batched1d == 0
    a.info()
