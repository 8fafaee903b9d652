"""TiledWavelets: ONE 2D image split in row slabs over the GPUs of a torch.distributed group.

The batched case (independent images, one plan per GPU) needs no communication and is what bench.py
measures.  This module is the other multi-GPU case of the north star: a single image too large for one
GPU.  Rank r of G owns rows [r n, (r+1) n) of a (G n) x Nc image and the matching row slab of every
sub-band.  The row pass of a level is local; the column filters reach `hlen/2 - 1` rows into the
neighbouring slabs (analysis) and at most `hlen/4 + 1` coefficient rows (synthesis), so each level does
one ring halo exchange with the two neighbours -- point-to-point send/recv (RCCL over xGMI with the
"nccl" backend: neighbour traffic only, no collective) -- and then runs the ordinary single-GPU level
kernels on the slab extended by the halo rows, keeping the interior of the result: the periodic wrap of
the kernels only touches rows that are discarded.  The ring is periodic, like the transform
(reference semantics pdwt/src/separable.cu:114-121).

Restrictions: separable decimated 2D DWT, float32, rows-per-rank and columns divisible by 2^levels, and
a slab at the coarsest level at least as tall as the halo (beyond that the remaining approximation is
small enough to gather on one GPU, which this class does not do).

torch is plumbing here (device tensors, streams, torch.distributed); all arithmetic is done by the HIP
library through the C ABI, with zero-copy views of the plans' device buffers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PdwtInfo, check, handle_t


class _DeviceView(object):
    """__cuda_array_interface__ carrier for a borrowed device pointer (pdwt_image_ptr / pdwt_coeff_ptr)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TiledWavelets(object):
    def __init__(self, slab, wname, levels, group=None):
        import sys
        if "torch" not in sys.modules and _lib._libs:
            # PyTorch-ROCm bundles its own libamdhip64 under the same soname as /opt/rocm's: whichever is
            # loaded first serves both, and torch does not initialise on top of the system runtime
            raise RuntimeError("TiledWavelets: import torch before the first use of pypwt_amd in this process "
                               "(torch and libpypwt_amd.so must share torch's HIP runtime)")
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("TiledWavelets needs a HIP device (there is no CPU implementation)")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._via_host = dist.is_initialized() and dist.get_backend(group) != "nccl"  # gloo: stage halos on the host
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.slab = torch.as_tensor(np.ascontiguousarray(slab, dtype=np.float32) if isinstance(slab, np.ndarray)
                                    else slab, dtype=torch.float32, device=self.device).contiguous().clone()
        if self.slab.dim() != 2:
            raise ValueError("TiledWavelets: the slab must be a 2D array (rows of this rank x all columns)")
        self.n, self.Nc = int(self.slab.shape[0]), int(self.slab.shape[1])
        self.wname = str(wname)
        buf = (C.c_float * 160)()
        hlen = self._lib.pdwt_wavelet_filters(self.wname.encode("ASCII"), buf, 160)
        check(hlen, "TiledWavelets()", self._lib)
        self.hlen = int(hlen)
        self.levels = int(levels)
        if self.levels < 1 or self.n % (1 << self.levels) or self.Nc % (1 << self.levels):
            raise ValueError("TiledWavelets: rows per rank (%d) and columns (%d) must be divisible by 2^levels"
                             % (self.n, self.Nc))
        c = self.hlen // 2 - 1
        self._hp = c + (c & 1)                        # analysis halo rows (even)
        H2 = self.hlen // 2
        C2, S = H2 // 2, (0 if (H2 & 1) else 1)
        self._hq = max(C2, H2 - 1 - C2 + S)           # synthesis halo rows (coefficient rows)
        coarse = self.n >> (self.levels - 1)
        if coarse < self._hp or (coarse >> 1) < self._hq:
            raise ValueError("TiledWavelets: the slab at the last level (%d rows) is thinner than the halo (%d)"
                             % (coarse, max(self._hp, 2 * self._hq)))
        self._plans = {}
        self._bands = None   # after forward(): [A_L, (H1,V1,D1), ...] torch slabs
        # every plan runs on ONE side stream that torch also uses for its copies (a NULL stream handle would
        # mean "private stream" to pdwt_create_batched, unordered with torch's default stream)
        self._stream = torch.cuda.Stream(device=self.device)

    # ---- single-level plans on the extended slab, cached per shape
    def _plan(self, rows, cols):
        key = (rows, cols)
        if key not in self._plans:
            h = handle_t()
            stream = self._stream.cuda_stream
            rc = self._lib.pdwt_create_batched(None, 1, rows, cols, self.wname.encode("ASCII"), 1, 1, 1, 0, 0, 2,
                                               self.device.index, C.c_void_p(stream), C.byref(h))
            check(rc, "TiledWavelets plan", self._lib)
            info = PdwtInfo()
            check(self._lib.pdwt_get_info(h, C.byref(info), None, None, None, None), lib=self._lib)
            if info.nlevels != 1:
                raise ValueError("TiledWavelets: %d x %d is too small for one level of %s" % (rows, cols, self.wname))
            self._plans[key] = h
        return self._plans[key]

    def _view(self, ptr, shape):
        return self._torch.as_tensor(_DeviceView(ptr, shape), device=self.device)

    # ---- ring exchange: returns (rows from the previous rank's bottom, rows from the next rank's top)
    def _exchange(self, top, bottom):
        torch, dist = self._torch, self._dist
        if self.world == 1:
            return bottom.clone(), top.clone()  # the ring closes on itself: periodic image
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        if self._via_host:
            top_h, bot_h = top.cpu(), bottom.cpu()
            from_prev, from_next = torch.empty_like(bot_h), torch.empty_like(top_h)
            reqs = [dist.isend(bot_h, nxt, group=self.group, tag=1), dist.isend(top_h, prev, group=self.group, tag=2),
                    dist.irecv(from_prev, prev, group=self.group, tag=1),
                    dist.irecv(from_next, nxt, group=self.group, tag=2)]
            for r in reqs:
                r.wait()
            return from_prev.to(self.device), from_next.to(self.device)
        from_prev, from_next = torch.empty_like(bottom), torch.empty_like(top)
        # one grouped launch (ncclGroupStart/End): with two ranks both neighbours are the same peer, and
        # the k-th send to a peer matches its k-th receive, hence this order
        ops = [dist.P2POp(dist.isend, bottom.contiguous(), nxt, self.group),
               dist.P2POp(dist.isend, top.contiguous(), prev, self.group),
               dist.P2POp(dist.irecv, from_prev, prev, self.group),
               dist.P2POp(dist.irecv, from_next, nxt, self.group)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        return from_prev, from_next

    def _on_stream(self):
        """Context: torch work goes to the plans' stream, ordered after / before the caller's stream."""
        torch = self._torch
        outer = torch.cuda.current_stream(self.device)
        side = self._stream

        class _Ctx(object):
            def __enter__(ctx):
                side.wait_stream(outer)
                ctx.inner = torch.cuda.stream(side)
                ctx.inner.__enter__()

            def __exit__(ctx, *exc):
                ctx.inner.__exit__(*exc)
                outer.wait_stream(side)
                return False

        return _Ctx()

    def forward(self, slab=None):
        with self._on_stream():
            return self._forward(slab)

    def inverse(self):
        with self._on_stream():
            return self._inverse()

    def _forward(self, slab=None):
        torch = self._torch
        if slab is not None:
            self.slab.copy_(torch.as_tensor(slab, dtype=torch.float32, device=self.device))
        cur, hp = self.slab, self._hp
        bands = [None]
        for _ in range(self.levels):
            m, w = int(cur.shape[0]), int(cur.shape[1])
            h = self._plan(m + 2 * hp, w)
            img = self._view(self._lib.pdwt_image_ptr(h), (m + 2 * hp, w))
            if hp:
                from_prev, from_next = self._exchange(cur[:hp], cur[m - hp:])
                img[:hp].copy_(from_prev)
                img[hp + m:].copy_(from_next)
            img[hp:hp + m].copy_(cur)
            check(self._lib.pdwt_forward(h), "TiledWavelets.forward", self._lib)
            rows2, w2 = (m + 2 * hp) // 2, w // 2
            out = [self._view(self._lib.pdwt_coeff_ptr(h, k), (rows2, w2))[hp // 2:hp // 2 + m // 2].clone()
                   for k in range(4)]
            bands.append((out[1], out[2], out[3]))
            cur = out[0]
        bands[0] = cur
        self._bands = bands
        return self

    def _inverse(self):
        if self._bands is None:
            raise RuntimeError("TiledWavelets.inverse: call forward() first")
        torch, hq = self._torch, self._hq
        cur = self._bands[0]
        for lvl in range(self.levels, 0, -1):
            H, V, D = self._bands[lvl]
            m2, w2 = int(cur.shape[0]), int(cur.shape[1])
            h = self._plan(2 * (m2 + 2 * hq), 2 * w2)
            ext = [cur, H, V, D]
            if hq:
                stack = torch.stack(ext)                                     # (4, m2, w2)
                from_prev, from_next = self._exchange(stack[:, :hq].contiguous(), stack[:, m2 - hq:].contiguous())
                ext = [torch.cat([from_prev[k], ext[k], from_next[k]]) for k in range(4)]
            # band 0 through set_coeff (it also makes the plan's coefficients current again after the
            # previous inverse), the details straight into the plan's buffers
            check(self._lib.pdwt_set_coeff(h, C.c_void_p(ext[0].contiguous().data_ptr()), 0, 1), lib=self._lib)
            for k in (1, 2, 3):
                self._view(self._lib.pdwt_coeff_ptr(h, k), (m2 + 2 * hq, w2)).copy_(ext[k])
            check(self._lib.pdwt_inverse(h), "TiledWavelets.inverse", self._lib)
            img = self._view(self._lib.pdwt_image_ptr(h), (2 * (m2 + 2 * hq), 2 * w2))
            cur = img[2 * hq:2 * hq + 2 * m2].clone()
        self.slab = cur
        return self

    # ---- results (this rank's slabs)
    @property
    def image(self):
        self._torch.cuda.synchronize(self.device)
        return self.slab.cpu().numpy()

    @property
    def coeffs(self):
        """[A, [H1, V1, D1], [H2, V2, D2], ...] like Wavelets.coeffs, each the row slab of this rank."""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.coeffs: call forward() first")
        self._torch.cuda.synchronize(self.device)
        return [self._bands[0].cpu().numpy()] + [[b.cpu().numpy() for b in lvl] for lvl in self._bands[1:]]

    def cleanup(self):
        for h in self._plans.values():
            self._lib.pdwt_destroy(h)
        self._plans = {}

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
