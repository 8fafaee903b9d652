"""Worker of tests/test_gpu_tiled.py: one rank of a TiledWavelets run (launched as a child process with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set).  Every rank computes the transform of the WHOLE image
with the CPU ORACLE and compares its own row slab of every sub-band (the whole band for the levels that
were gathered on rank 0) and of the reconstruction; prints 'OK <rank> tiled=<t> deep=<d>' on success.
No torch anywhere: the transport is pypwt_amd.comm.HostRing (TCP, staged on the host: the ranks of a test share ONE GPU, which RCCL
refuses), pypwt_amd.comm.Communicator (the library's RCCL calls; one rank as its own neighbour on this box) or none (one rank: the
ring closes on itself).  argv: wname levels Nr Nc [backend = ring | comm | none [do_swt [fuse_last]]]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle
    from pypwt_amd.tiled import TiledWavelets

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    wname, levels = sys.argv[1], int(sys.argv[2])
    Nr, Nc = int(sys.argv[3]), int(sys.argv[4])
    backend = sys.argv[5] if len(sys.argv) > 5 else "ring"
    swt = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    fuse = (int(sys.argv[7]) or None) if len(sys.argv) > 7 else None
    comm = ring = None
    if backend == "comm":  # the library's own RCCL transport (pdwt_comm_*)
        from pypwt_amd.comm import Communicator
        comm = Communicator.from_env(device=0)
        assert comm.size == world and comm.rank == rank
    elif world > 1:  # one rank without a transport: the ring closes on the rank itself
        from pypwt_amd.comm import HostRing
        ring = HostRing.from_env()
    x = oracle.hash_input((Nr, Nc), 555, scale=255.0)
    n = Nr // world
    tw = TiledWavelets(x[rank * n:(rank + 1) * n], wname, levels, do_swt=swt, comm=comm, ring=ring, device=0, fuse_last=fuse)
    tw.forward()
    flat = oracle.forward(x, wname, levels, do_swt=swt)  # [A, H1, V1, D1, H2, ...] from the CPU oracle
    ref = [flat[0]] + [flat[1 + 3 * l:4 + 3 * l] for l in range(levels)]
    got = tw.coeffs
    assert tw.tiled_levels + tw.deep_levels == levels and tw.tiled_levels >= 1

    def slab(a):
        k = a.shape[0] // world
        return a[rank * k:(rank + 1) * k]

    tol = 2e-6 * (levels + 1) * 255 * 4 ** levels  # SWT bands grow by 2 per level like the decimated ones
    for lvl in range(1, levels + 1):
        if lvl <= tw.tiled_levels:
            for g, r in zip(got[lvl], ref[lvl]):
                assert g.shape == slab(r).shape and np.abs(g - slab(r)).max() <= tol, ("level", lvl)
        elif rank == 0:  # gathered levels: the whole band lives on rank 0
            for g, r in zip(got[lvl], ref[lvl]):
                assert g.shape == r.shape and np.abs(g - r).max() <= tol, ("gathered level", lvl)
        else:
            assert got[lvl] is None
    if tw.deep_levels == 0:
        assert np.abs(got[0] - slab(ref[0])).max() <= tol, "A"
    elif rank == 0:
        assert got[0].shape == ref[0].shape and np.abs(got[0] - ref[0]).max() <= tol, "A (gathered)"
    else:
        assert got[0] is None
    tw.forward()  # a forward does not consume the slab: the same coefficients again
    again = tw.coeffs
    for a, b in zip(got[1], again[1]):
        assert np.array_equal(a, b), "second forward"
    tw.inverse()
    if swt:  # the reconstruction against the oracle's inverse of the oracle's coefficients too
        rec = oracle.inverse(flat, x.shape, wname, levels, do_swt=1)
        assert np.abs(tw.image - rec[rank * n:(rank + 1) * n]).max() <= 2e-3, "reconstruction vs oracle"
    assert np.abs(tw.image - x[rank * n:(rank + 1) * n]).max() <= 2e-3, "reconstruction"
    import warnings
    with warnings.catch_warnings(record=True) as caught:  # the image is current: a second inverse changes nothing
        warnings.simplefilter("always")                    # (the reference's W_INVERSE state) and says so
        tw.inverse()
    assert any("already been run" in str(w.message) for w in caught), "second inverse must warn"
    tw.forward()
    for lvl in tw.device_coeffs[1:tw.tiled_levels + 1]:  # zero-copy views of the plans' buffers: shrink the details in place
        for b in lvl:
            b.set(b.get() * np.float32(0.5))
    tw.inverse()
    flat2 = [f.copy() for f in flat]
    for k in range(1, 3 * tw.tiled_levels + 1):
        flat2[k] *= np.float32(0.5)
    rec2 = oracle.inverse(flat2, x.shape, wname, levels, do_swt=swt)[rank * n:(rank + 1) * n]
    assert np.abs(tw.image - rec2).max() <= 2e-3, "inverse of coefficients modified in place"
    # forward -> inverse -> edit in place -> inverse: refused (stale image) unless mark_coeffs_current() re-arms it
    for lvl in tw.device_coeffs[1:tw.tiled_levels + 1]:
        for b in lvl:
            b.set(b.get() * np.float32(0.5))
    tw.mark_coeffs_current()
    tw.inverse()
    flat3 = [f.copy() for f in flat2]
    for k in range(1, 3 * tw.tiled_levels + 1):
        flat3[k] *= np.float32(0.5)
    rec3 = oracle.inverse(flat3, x.shape, wname, levels, do_swt=swt)[rank * n:(rank + 1) * n]
    assert np.abs(tw.image - rec3).max() <= 2e-3, "inverse after mark_coeffs_current"
    tw.forward()
    tw.inverse()  # plans are reused: another round trip must work too
    assert np.abs(tw.image - rec3).max() <= 4e-3, "second round trip"
    groups = tw.groups
    tw.synchronize()
    tw.cleanup()
    if comm is not None:
        comm.close()
    elif ring is not None:
        ring.barrier()
        ring.close()
    assert "torch" not in sys.modules, "the tiled path must not need torch"
    print("OK %d tiled=%d deep=%d groups=%s%s" % (rank, tw.tiled_levels, tw.deep_levels, ",".join("%d+%d" % g for g in groups),
                                                  " comm-loopback" if comm is not None else ""))


if __name__ == "__main__":
    main()
