#!/usr/bin/env python3
"""One GPU, 128 images of cfg2 per step: ONE plan of 128 images against TWO / FOUR plans of 64 / 32 images on their own streams,
launched interleaved (does a second stream fill the first one's kernel tails?).  python3 tools/twostreams.py"""
import sys, time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets


def run(parts, total=128, steps=30):
    B = total // parts
    plans = [BatchedWavelets(B, 4096, 4096, "db4", 4) for _ in range(parts)]
    for i, p in enumerate(plans):
        p.fill_hash(7 + i)
    def step():
        for p in plans:
            p.forward()
        for p in plans:
            p.inverse()
    for _ in range(5):
        step()
    for p in plans:
        p.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for p in plans:
        p.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%d plan(s) x %3d images: %.3f ms per step of %d images = %.1f us per image" % (parts, B, dt * 1e3, total, dt / total * 1e6), flush=True)
    for p in plans:
        p.cleanup()


for parts in (1, 2, 4, 1, 2):
    run(parts)
