"""ShardedBatch -- a batch of independent images sharded over the GPUs of ONE node, one process.

BASELINE.json's north star: "a batch of independent images shards one-per-GPU across the 8 x MI355X node with no
collectives".  The reference has no device selection at all (pdwt/TODO.txt:15: "multi-GPU"); a user of its `Wavelets`
class loops over images on one GPU.  Here the images are cut into contiguous blocks, one per device; every device gets
ONE batched plan (`BatchedWavelets`: all its images in the same launches), its own stream and its own host thread, so
that the per-device calls -- which release the GIL inside ctypes -- are enqueued and waited for concurrently.  Nothing
is exchanged between devices: no RCCL, no peer copies, no host staging; `coeff_at(num, b)` / `image_at(b)` are routed
to the device that owns image b.  The C ABI stays per device (`pdwt_create_batched(..., device_id, ...)`).

`bench.py --gpus N --single-process` measures this class; the default multi-GPU bench is one process per GPU
(the driver's launcher), which shards the same way (`bench.py: shard_images`).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .wavelets import BatchedWavelets


def partition_images(total, parts):
    """Contiguous blocks [lo, hi) of `total` images for `parts` owners, the remainder going to the first ones
    (the same rule as bench.py: shard_images).  Owners may end up empty when total < parts."""
    total, parts = int(total), int(parts)
    if total < 0 or parts < 1:
        raise ValueError("partition_images: need total >= 0 and parts >= 1")
    base, rem = divmod(total, parts)
    out, lo = [], 0
    for r in range(parts):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def owner_of(blocks, b):
    """(owner index, local index) of image b in the partition `blocks`."""
    for i, (lo, hi) in enumerate(blocks):
        if lo <= b < hi:
            return i, b - lo
    raise IndexError("image %d is outside the batch of %d" % (b, blocks[-1][1] if blocks else 0))


def device_count():
    """HIP devices visible to this process (0 without a GPU: there is no CPU path)."""
    return int(_lib.load().pdwt_device_count())


class ShardedBatch(object):
    """`batch` images of (Nr, Nc), contiguous blocks per device.

        S = ShardedBatch(1024, 4096, 4096, "db4", 4)          # every visible GPU
        S = ShardedBatch(64, 2048, 2048, "haar", 5, devices=[0, 1, 2, 3], do_swt=1)
        S.fill_hash(7); S.forward(); S.soft_threshold(10.0); S.inverse(); S.synchronize()
        S.coeff_at(0, 517)      # band 0 of image 517, from the GPU that owns it
        S.shards                # [(device, first image, one past the last)]

    `devices` may name a device more than once (two plans with their own streams on one GPU: how the one-GPU test box
    exercises the routing).  Devices that would own no image (batch < len(devices)) get no plan.
    """

    def __init__(self, batch, Nr, Nc, wname, levels, devices=None, do_swt=0, ndim=2):
        if devices is None:
            n = device_count()
            if n < 1:
                raise RuntimeError("ShardedBatch: no HIP device visible (this library has no CPU path)")
            devices = list(range(n))
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("ShardedBatch: empty device list")
        self.batch, self.Nr, self.Nc = int(batch), int(Nr), int(Nc)
        if self.batch < 1:
            raise ValueError("ShardedBatch: batch must be >= 1")
        blocks = [(d, lo, hi) for d, (lo, hi) in zip(devices, partition_images(self.batch, len(devices))) if hi > lo]
        self.shards = blocks
        self._blocks = [(lo, hi) for _, lo, hi in blocks]
        # one single-thread executor per shard: calls for one plan stay ordered, calls for different plans overlap
        self._pools = [ThreadPoolExecutor(max_workers=1) for _ in blocks]
        self.plans = self._each(lambda i: BatchedWavelets(blocks[i][2] - blocks[i][1], Nr, Nc, wname, levels, do_swt=do_swt,
                                                          ndim=ndim, device=blocks[i][0]), build=True)
        p0 = self.plans[0]
        self.levels, self.hlen, self.ndims, self.do_swt, self.nbands = p0.levels, p0.hlen, p0.ndims, p0.do_swt, p0.nbands

    # ---- plumbing
    def _each(self, fn, build=False):
        """fn(shard index) on every shard's own thread; returns the results in shard order (re-raises the first error)."""
        futs = [pool.submit(fn, i) for i, pool in enumerate(self._pools)]
        errs, out = [], []
        for f in futs:
            try:
                out.append(f.result())
            except Exception as e:  # noqa: BLE001 -- collected, the first one is re-raised below
                errs.append(e)
                out.append(None)
        if errs:
            if build:
                for p in out:
                    if p is not None:
                        p.cleanup()
            raise errs[0]
        return out

    def owner(self, b):
        """(shard index, index inside that shard's plan) of image b."""
        return owner_of(self._blocks, int(b))

    # ---- the Wavelets verbs, on every shard
    def fill_hash(self, seed, scale=255.0, index_offset=0):
        """Deterministic on-device input: image b of the batch is the same whatever the number of devices."""
        per = self.Nr * self.Nc
        self._each(lambda i: self.plans[i].fill_hash(seed, scale, index_offset + self._blocks[i][0] * per))

    def forward(self):
        self._each(lambda i: self.plans[i].forward())

    def inverse(self):
        self._each(lambda i: self.plans[i].inverse())

    def soft_threshold(self, beta, do_threshold_appcoeffs=0, normalize=0):
        self._each(lambda i: self.plans[i].soft_threshold(beta, do_threshold_appcoeffs, normalize))

    def synchronize(self):
        self._each(lambda i: self.plans[i].synchronize())

    def set_image(self, img):
        """img: host array (batch, Nr, Nc); every device uploads its own block."""
        img = np.ascontiguousarray(img, dtype=np.float32)
        if img.shape != (self.batch, self.Nr, self.Nc):
            raise ValueError("ShardedBatch.set_image: expected shape %s, got %s" % ((self.batch, self.Nr, self.Nc), img.shape))
        self._each(lambda i: self.plans[i].set_image(img[self._blocks[i][0]:self._blocks[i][1]]))

    def norm2sq(self):
        return float(sum(self._each(lambda i: self.plans[i].norm2sq())))

    # ---- results, routed to the owner
    def coeff_at(self, num, b):
        i, k = self.owner(b)
        return self._pools[i].submit(self.plans[i].coeff_at, num, k).result()

    def image_at(self, b):
        i, k = self.owner(b)
        return self._pools[i].submit(self.plans[i].image_at, k).result()

    def schedule(self):
        return [p.schedule() for p in self.plans]

    def cleanup(self):
        plans, self.plans = getattr(self, "plans", None) or [], []
        for p in plans:
            if p is not None:
                p.cleanup()
        for pool in getattr(self, "_pools", []):
            pool.shutdown(wait=True)
        self._pools = []

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
