#!/usr/bin/env python3
"""What the tiled single-image path (pypwt_amd/tiled.py) costs per step on ONE rank: a slab of `rows` x `cols` with
the halos (a) copied on the device (no transport: the ring closes on the rank) and (b) sent through the library's RCCL calls to the
rank itself (a Communicator of one rank) -- one grouped send/recv per level group, all_gather + broadcast for the
gathered levels.  (b) - (a) is the price of the transport calls of one rank; the links themselves are not in it.
The plain single-GPU plan of the same slab is printed beside them; every level its own group (1+1,...) and the last two
slab levels as one group (round 5).

    python3 tools/tiledbench.py [rows cols wname levels] > profiles/r03_tiledbench.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, sync, n=20):
    """median of n calls, each device-synchronised (the first calls of a new message size set RCCL buffers up: warmed)"""
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[n // 2] * 1e6


def main():
    from pypwt_amd import Wavelets
    from pypwt_amd.tiled import TiledWavelets

    wname = sys.argv[3] if len(sys.argv) > 3 else "db4"
    levels = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4096, 4096), (16384, 16384)]
    for rows, cols in sizes:
        one_size(Wavelets, TiledWavelets, rows, cols, wname, levels)
    assert "torch" not in sys.modules


def one_size(Wavelets, TiledWavelets, rows, cols, wname, levels):
    from pypwt_amd.comm import Communicator
    x = (np.random.RandomState(3).rand(rows, cols) * 255).astype(np.float32)
    print("# tools/tiledbench.py: one rank's slab %d x %d, %s, %d levels (us per call, median of 20 synchronised calls; no torch in the process)" % (rows, cols, wname, levels))
    for swt in (0, 1):
        lv = levels if not swt else min(levels, 3)
        W = Wavelets(x, wname, lv, do_swt=swt)
        t_plain_f = timed(W.forward, W.synchronize)
        t_plain_fi = timed(lambda: (W.forward(), W.inverse()), W.synchronize)
        del W
        res = []
        for fuse in ((1, 2, None) if not swt else (None,)):
            tw = TiledWavelets(x, wname, lv, do_swt=swt, fuse_last=fuse)
            t_copy_f = timed(tw.forward, tw.synchronize)
            t_copy_fi = timed(lambda: (tw.forward(), tw.inverse()), tw.synchronize)
            tiled, deep, groups = tw.tiled_levels, tw.deep_levels, tw.groups
            tw.cleanup()
            del tw
            # the library's own RCCL calls (pdwt_comm_exchange on the plans' stream), one rank as its own neighbour
            comm = Communicator.single(0)
            tw = TiledWavelets(x, wname, lv, do_swt=swt, comm=comm, fuse_last=fuse)
            t_c_f = timed(tw.forward, tw.synchronize)
            t_c_fi = timed(lambda: (tw.forward(), tw.inverse()), tw.synchronize)
            tw.synchronize()
            tw.cleanup()
            del tw
            comm.close()
            res.append((groups, t_copy_f, t_copy_fi, t_c_f, t_c_fi))
        for groups, t_copy_f, t_copy_fi, t_c_f, t_c_fi in res:
            print("%s L=%d (slab levels %d, gathered %d, level groups %s): plain plan fwd %8.1f  fwd+inv %8.1f | tiled, halos copied fwd %8.1f  fwd+inv %8.1f"
                  " | tiled, halos through the library's RCCL calls (loopback) fwd %8.1f  fwd+inv %8.1f" % (
                      "swt2" if swt else "dwt2", lv, tiled, deep, ",".join("%d+%d" % g for g in groups), t_plain_f, t_plain_fi, t_copy_f, t_copy_fi,
                      t_c_f, t_c_fi), flush=True)


if __name__ == "__main__":
    main()
