#!/usr/bin/env python3
"""Proxies for two schedules (round 4), built from the existing API before anything is written in plan.cpp:

 A. batch of 4096^2 db4 L4 images, IMAGE-MAJOR: 128 plans of 1 image (64 of 2) on S streams round robin, so that every
    image's A1/A2/A3 are re-read while they still sit in the Infinity Cache and the latency-bound small levels of one
    image run under the bandwidth-bound level 1 of another.  Against ONE plan of 128 images (level-major).
 B. ONE 4096^2 image as S row slabs extended by halo rows (overlap-save: the slabs are independent through all levels),
    each on its own stream.  Right bytes and launches, wrong wrap rows (the proxy only times).

    python3 tools/slabproxy.py [A] [B]
"""
import ctypes as C
import sys
import time

sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets

hip = C.CDLL("libamdhip64.so.7")


def mkstream(prio=0):
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithPriority(C.byref(s), 1, int(prio))  # hipStreamNonBlocking
    assert rc == 0, rc
    return s.value


def prio_range():
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    return lo.value, hi.value  # (least, greatest): greatest is numerically lower


def timed(step, sync, steps, warm=3):
    for _ in range(warm):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps


def preheat(ms=300):
    p = BatchedWavelets(1, 4096, 4096, "db4", 4)
    p.fill_hash(3)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            p.forward(); p.inverse()
        p.synchronize()
    p.cleanup()


def batch_case(nplans, per_plan, nstreams, order="level", steps=12, prios=None):
    streams = [mkstream(prios[i] if prios else 0) for i in range(nstreams)] if nstreams else None
    plans = []
    for i in range(nplans):
        plans.append(BatchedWavelets(per_plan, 4096, 4096, "db4", 4, stream=streams[i % nstreams] if streams else None))
        plans[-1].fill_hash(7 + i)

    if order == "level":  # forward of every image, then inverse of every image (what forward() / inverse() of a batch must do)
        def step():
            for p in plans:
                p.forward()
            for p in plans:
                p.inverse()
    else:  # "image": forward + inverse of an image back to back (upper bound: coefficients never leave the cache)
        def step():
            for p in plans:
                p.forward(); p.inverse()

    def sync():
        for p in plans:
            p.synchronize()
    dt = timed(step, sync, steps)
    total = nplans * per_plan
    print("A: %3d plan(s) x %3d image(s), %s streams, order=%-5s prio=%s : %8.3f ms per %d images = %6.2f us per image"
          % (nplans, per_plan, nstreams or "own", order, prios, dt * 1e3, total, dt / total * 1e6), flush=True)
    for p in plans:
        p.cleanup()
    for s in streams or []:
        hip.hipStreamDestroy(C.c_void_p(s))


def slab_case(nslabs, rows, nstreams, steps=300, prios=None, label=""):
    streams = [mkstream(prios[i] if prios else 0) for i in range(nstreams)]
    plans = [BatchedWavelets(1, rows, 4096, "db4", 4, stream=streams[i % nstreams]) for i in range(nslabs)]
    for i, p in enumerate(plans):
        p.fill_hash(11 + i)

    def step():
        for p in plans:
            p.forward()
        for p in plans:
            p.inverse()

    def sync():
        for p in plans:
            p.synchronize()
    best = min(timed(step, sync, steps) for _ in range(3))
    print("B: %d slab(s) of %4d rows x 4096 on %d stream(s) prio=%s %s: %7.2f us per step (%s)"
          % (nslabs, rows, nstreams, prios, label, best * 1e6, plans[0].schedule().replace("\n", " | ")), flush=True)
    for p in plans:
        p.cleanup()
    for s in streams:
        hip.hipStreamDestroy(C.c_void_p(s))


if __name__ == "__main__":
    which = sys.argv[1:] or ["A", "B"]
    lo, hi = prio_range()
    print("stream priority range: least %d greatest %d" % (lo, hi), flush=True)
    preheat()
    if "B" in which:
        slab_case(1, 4096, 1, label="baseline")
        slab_case(2, 2048, 1, label="no halo, one stream")
        slab_case(2, 2048, 2, label="no halo")
        slab_case(2, 2240, 2, label="halo 96")
        slab_case(2, 2240, 2, prios=[0, hi], label="halo 96")
        slab_case(2, 2240, 1, label="halo 96, one stream")
        slab_case(3, 1568, 3, label="halo 96")
        slab_case(4, 1216, 4, label="halo 96")
        slab_case(4, 1216, 2, label="halo 96")
        slab_case(4, 1024, 4, label="no halo")
        slab_case(8, 512, 4, label="no halo")
        slab_case(1, 4096, 1, label="baseline again")
    if "A" in which:
        batch_case(1, 128, 0)
        for S in (1, 2, 3, 4, 8):
            batch_case(128, 1, S)
        batch_case(128, 1, 2, prios=[0, hi])
        for S in (2, 4):
            batch_case(64, 2, S)
        batch_case(32, 4, 2)
        for S in (2, 4):
            batch_case(128, 1, S, order="image")
        batch_case(1, 128, 0)
