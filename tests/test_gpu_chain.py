"""Several 2D DWT levels in ONE launch (pypwt_amd/csrc/dwt2_chain_kernels.hpp): every band against the CPU oracle, element
by element, through the C ABI -- forced on small shapes (tuning key "chain" = 2), with the hand-off timeout at 0 (every
wait that is not satisfied at once takes the compute-it-yourself path) and for batches (staggered steps).  The chained
launches are an experiment that measured no faster: they live in the test-only library
libpypwt_amd_lab.so, which the `lib` fixture selects for this module."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    """The chained launches are an experiment: they live in the test-only library libpypwt_amd_lab.so."""
    oracle.build()
    from pypwt_amd import _lib
    was = _lib.use_lab_kernels(True)
    yield _lib.load()
    _lib.use_lab_kernels(was)


@pytest.fixture
def chain(lib):
    """yields a function that sets the "chain" knob; restores the knobs afterwards"""
    prev = {}

    def set_(key, value):
        old = lib.pdwt_set_tuning(key.encode(), int(value))
        prev.setdefault(key, old)
    yield set_
    for k, v in prev.items():
        lib.pdwt_set_tuning(k.encode(), int(v))


def launches(plan):
    plan.enable_kernel_timing(True)
    plan.reset_kernel_times()
    plan.forward()
    plan.inverse()
    names = [n for n, _ in plan.kernel_times(cap=64)]
    plan.enable_kernel_timing(False)
    plan.reset_kernel_times()
    return names


def check_against_oracle(shape, wname, levels, batch, seed=5, want_chain=True):
    from pypwt_amd import BatchedWavelets
    Nr, Nc = shape
    plan = BatchedWavelets(batch, Nr, Nc, wname, levels)
    plan.fill_hash(seed, 255.0)
    names = launches(plan)
    if want_chain:
        assert "dwt2_fwd_chain" in names and "dwt2_inv_chain" in names, names
    x = oracle.hash_input((batch * Nr, Nc), seed).reshape(batch, Nr, Nc)
    for rep in range(2):  # the second pass runs on flags stamped by the first (epochs, no reset)
        plan.forward()
        for b in sorted({0, batch // 2, batch - 1}):
            ref = oracle.forward(x[b], wname, levels)
            for num, r in enumerate(ref):
                g = plan.coeff_at(num, b)
                assert g.shape == r.shape
                err = np.abs(g - r).max()
                assert err <= 1.5e-6 * (levels + 1) * max(np.abs(r).max(), 1.0), (shape, wname, levels, batch, b, num, err)
        plan.inverse()
        for b in sorted({0, batch - 1}):
            assert np.abs(plan.image_at(b) - x[b]).max() <= 2e-5 * 255, (shape, wname, batch, b)
    plan.cleanup()


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior2.2"])
def test_chain_every_band_vs_oracle(chain, wname):
    chain("chain", 2)
    hl = {"haar": 2, "db2": 4, "db3": 6, "db4": 8, "sym4": 8, "bior2.2": 6}[wname]
    assert hl <= 8
    check_against_oracle((256, 1024), wname, 3, 1)


@pytest.mark.parametrize("batch", [2, 3, 5, 9])
def test_chain_batches_staggered_and_not(chain, batch):
    chain("chain", 3)
    check_against_oracle((128, 512), "db4", 2, batch, seed=11)


def test_chain_self_help_path_is_exact(chain):
    """timeout 0: a tile whose producers have not ALL published at its first poll computes them itself (recursively)"""
    chain("chain", 3)
    chain("chain_timeout", 0)
    check_against_oracle((512, 1024), "db4", 4, 1, seed=3)
    check_against_oracle((128, 512), "db2", 2, 6, seed=4)


def test_chain_rectangular_and_partial_groups(chain):
    chain("chain", 2)
    check_against_oracle((1024, 512), "db4", 2, 1, seed=8)       # tall
    check_against_oracle((512, 2048), "db3", 5, 1, seed=9)       # 5 levels asked, the chain takes what divides, the rest follows
    check_against_oracle((96, 384), "db4", 2, 1, seed=10, want_chain=False)  # not whole tiles at level 2 -> classic path
