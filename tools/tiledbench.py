#!/usr/bin/env python3
"""What the tiled single-image path (pypwt_amd/tiled.py) costs per step on ONE rank: a slab of `rows` x `cols` with
the halos (a) copied on the device (no process group: the ring closes on the rank) and (b) sent through RCCL to the
rank itself (backend nccl, world size 1, loopback) -- grouped send/recv per level, all_gather + broadcast for the
gathered levels.  (b) - (a) is the price of the transport calls of one rank; the links themselves are not in it.
The plain single-GPU plan of the same slab is printed beside them.

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
    import torch
    import torch.distributed as dist
    from pypwt_amd import Wavelets
    from pypwt_amd.tiled import TiledWavelets

    wname = sys.argv[3] if len(sys.argv) > 3 else "db4"
    levels = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    torch.cuda.set_device(0)
    sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4096, 4096), (16384, 16384)]
    for rows, cols in sizes:
        one_size(torch, dist, Wavelets, TiledWavelets, rows, cols, wname, levels)
    if dist.is_initialized():
        dist.destroy_process_group()


def one_size(torch, dist, Wavelets, TiledWavelets, rows, cols, wname, levels):
    x = (np.random.RandomState(3).rand(rows, cols) * 255).astype(np.float32)
    sync = torch.cuda.synchronize
    print("# tools/tiledbench.py: one rank's slab %d x %d, %s, %d levels (us per call, median of 20 synchronised calls)" % (rows, cols, wname, levels))
    for swt in (0, 1):
        lv = levels if not swt else min(levels, 3)
        W = Wavelets(x, wname, lv, do_swt=swt)
        t_plain_f = timed(W.forward, sync)
        t_plain_fi = timed(lambda: (W.forward(), W.inverse()), sync)
        del W
        tw = TiledWavelets(x, wname, lv, do_swt=swt)
        t_copy_f = timed(tw.forward, sync)
        t_copy_fi = timed(lambda: (tw.forward(), tw.inverse()), sync)
        tiled, deep = tw.tiled_levels, tw.deep_levels
        tw.cleanup()
        del tw
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1)
            dist.barrier()
        tw = TiledWavelets(x, wname, lv, do_swt=swt, loopback=True)
        t_rccl_f = timed(tw.forward, sync)
        t_rccl_fi = timed(lambda: (tw.forward(), tw.inverse()), sync)
        tw.cleanup()
        del tw
        # (c) round 4: the library's own RCCL calls (pdwt_comm_exchange on the plans' stream), one rank as its own neighbour
        from pypwt_amd.comm import Communicator
        comm = Communicator.single(0)
        tw = TiledWavelets(x, wname, lv, do_swt=swt, comm=comm)
        t_c_f = timed(tw.forward, sync)
        t_c_fi = timed(lambda: (tw.forward(), tw.inverse()), sync)
        tw.cleanup()
        del tw
        sync()
        comm.close()
        print("%s L=%d (slab levels %d, gathered %d): plain plan fwd %8.1f  fwd+inv %8.1f | tiled, halos copied fwd %8.1f  fwd+inv %8.1f"
              " | tiled, halos through torch.distributed RCCL (loopback) fwd %8.1f  fwd+inv %8.1f"
              " | tiled, halos through the library's RCCL calls (loopback) fwd %8.1f  fwd+inv %8.1f" % (
                  "swt2" if swt else "dwt2", lv, tiled, deep, t_plain_f, t_plain_fi, t_copy_f, t_copy_fi, t_rccl_f, t_rccl_fi,
                  t_c_f, t_c_fi), flush=True)


if __name__ == "__main__":
    main()
