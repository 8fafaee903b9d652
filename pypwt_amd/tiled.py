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

After some levels a slab is thinner than the halo (or no longer divisible by two): the remaining
approximation band -- by then 4^-t of the image -- is GATHERED on rank 0 (one all-gather: the only
collective of the path), rank 0 finishes the transform with an ordinary single-GPU plan, and the inverse
hands the slabs back with one broadcast ("after ~log2(G) levels ... gather the remaining A
band onto one GPU").  `tiled_levels` says how many levels ran as slabs.

The undecimated transform (do_swt=1) needs no per-level exchange: an output row of level l depends on the image
rows within hlen (2^l - 1) of it, so ONE exchange of hs = hlen (2^levels - 1) image rows per side lets every rank run
the whole multi-level SWT plan on its extended slab and keep the interior rows of every band (the inverse: one
exchange of the same halo of all 3 levels + 1 bands, stacked into one message per neighbour).  SURVEY.md 8e's
"halo x 2^(l-1)" summed over the levels.

Restrictions: separable 2D transforms, float32.  DWT: columns divisible by 2^levels, rows per rank divisible by
2^tiled_levels with tiled_levels >= 1.  SWT: rows per rank >= hlen (2^levels - 1).

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
    def __init__(self, slab, wname, levels, group=None, do_swt=0):
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
        self.do_swt = int(bool(do_swt))
        self._deep = None    # rank 0: the single-GPU plan of the gathered approximation
        self._plans = {}
        self._bands = None   # after forward(): [A_L, (H1,V1,D1), ...] torch slabs
        # every plan runs on ONE side stream that torch also uses for its copies (a NULL stream handle would
        # mean "private stream" to pdwt_create_batched, unordered with torch's default stream)
        self._stream = torch.cuda.Stream(device=self.device)
        if self.do_swt:
            if self.levels < 1:
                raise ValueError("TiledWavelets: levels must be >= 1")
            self._hs = self.hlen * ((1 << self.levels) - 1)   # rows of each neighbour every band of this slab depends on
            if self._hs > self.n:
                raise ValueError("TiledWavelets: a slab of %d rows is thinner than the halo of a %d-level SWT with %s (%d rows)"
                                 % (self.n, self.levels, self.wname, self._hs))
            self.tiled_levels, self.deep_levels = self.levels, 0
            return
        if self.levels < 1 or self.Nc % (1 << self.levels):
            raise ValueError("TiledWavelets: columns (%d) must be divisible by 2^levels" % self.Nc)
        c = self.hlen // 2 - 1
        self._hp = c + (c & 1)                        # analysis halo rows (even)
        H2 = self.hlen // 2
        C2, S = H2 // 2, (0 if (H2 & 1) else 1)
        self._hq = max(C2, H2 - 1 - C2 + S)           # synthesis halo rows (coefficient rows)
        # levels that run as slabs: the slab entering the level is even, at least as tall as the analysis halo,
        # and its half at least as tall as the synthesis halo; the rest runs on rank 0 after the gather
        t, m = 0, self.n
        while t < self.levels and m % 2 == 0 and m >= self._hp and (m >> 1) >= self._hq and m >= 2:
            t, m = t + 1, m >> 1
        if t < 1:
            raise ValueError("TiledWavelets: a slab of %d rows is too thin (or odd) for one level of %s (halo %d)"
                             % (self.n, self.wname, max(self._hp, 2 * self._hq)))
        self.tiled_levels, self.deep_levels = t, self.levels - t

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

    # ---- undecimated transform: the whole multi-level plan on the slab extended by hs rows per side
    def _swt_plan(self):
        key = ("swt", self.n + 2 * self._hs, self.Nc)
        if key not in self._plans:
            h = handle_t()
            rc = self._lib.pdwt_create_batched(None, 1, key[1], key[2], self.wname.encode("ASCII"), self.levels, 1, 1, 0,
                                               1, 2, self.device.index, C.c_void_p(self._stream.cuda_stream), C.byref(h))
            check(rc, "TiledWavelets SWT plan", self._lib)
            info = PdwtInfo()
            check(self._lib.pdwt_get_info(h, C.byref(info), None, None, None, None), lib=self._lib)
            if info.nlevels != self.levels:
                self._lib.pdwt_destroy(h)
                raise ValueError("TiledWavelets: %d SWT levels requested, the slab allows only %d" % (self.levels, info.nlevels))
            self._plans[key] = h
        return self._plans[key]

    def _forward_swt(self):
        m, w, hs = self.n, self.Nc, self._hs
        h = self._swt_plan()
        img = self._view(self._lib.pdwt_image_ptr(h), (m + 2 * hs, w))
        from_prev, from_next = self._exchange(self.slab[:hs], self.slab[m - hs:])
        img[:hs].copy_(from_prev)
        img[hs + m:].copy_(from_next)
        img[hs:hs + m].copy_(self.slab)
        check(self._lib.pdwt_forward(h), "TiledWavelets.forward (SWT)", self._lib)
        flat = [self._view(self._lib.pdwt_coeff_ptr(h, k), (m + 2 * hs, w))[hs:hs + m].clone()
                for k in range(3 * self.levels + 1)]
        self._bands = [flat[0]] + [tuple(flat[1 + 3 * l:4 + 3 * l]) for l in range(self.levels)]
        return self

    def _inverse_swt(self):
        torch = self._torch
        m, w, hs = self.n, self.Nc, self._hs
        h = self._swt_plan()
        flat = [self._bands[0]] + [b for lvl in self._bands[1:] for b in lvl]
        stack = torch.stack(flat)                                            # (3 levels + 1, m, w)
        from_prev, from_next = self._exchange(stack[:, :hs].contiguous(), stack[:, m - hs:].contiguous())
        ext0 = torch.cat([from_prev[0], flat[0], from_next[0]]).contiguous()
        # band 0 through set_coeff (it makes the plan's coefficients current), the details into the plan's buffers
        check(self._lib.pdwt_set_coeff(h, C.c_void_p(ext0.data_ptr()), 0, 1), lib=self._lib)
        for k in range(1, len(flat)):
            dst = self._view(self._lib.pdwt_coeff_ptr(h, k), (m + 2 * hs, w))
            dst[:hs].copy_(from_prev[k])
            dst[hs:hs + m].copy_(flat[k])
            dst[hs + m:].copy_(from_next[k])
        check(self._lib.pdwt_inverse(h), "TiledWavelets.inverse (SWT)", self._lib)
        self.slab = self._view(self._lib.pdwt_image_ptr(h), (m + 2 * hs, w))[hs:hs + m].clone()
        return self

    def _forward(self, slab=None):
        torch = self._torch
        if slab is not None:
            self.slab.copy_(torch.as_tensor(slab, dtype=torch.float32, device=self.device))
        if self.do_swt:
            return self._forward_swt()
        cur, hp = self.slab, self._hp
        bands = [None]
        for _ in range(self.tiled_levels):
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
        if self.deep_levels:
            bands[0] = None
            self._a_slab_shape = tuple(cur.shape)
            full = self._gather_rows(cur)            # every rank receives it (all-gather); rank 0 uses it
            if self.rank == 0:
                h = self._deep_plan(int(full.shape[0]), int(full.shape[1]))
                check(self._lib.pdwt_set_image(h, C.c_void_p(full.contiguous().data_ptr()), 1), lib=self._lib)
                check(self._lib.pdwt_forward(h), "TiledWavelets.forward (gathered levels)", self._lib)
                rows, cols = C.c_int(), C.c_int()
                deep = []
                for num in range(3 * self.deep_levels + 1):
                    self._lib.pdwt_coeff_count(h, num, C.byref(rows), C.byref(cols))
                    deep.append(self._view(self._lib.pdwt_coeff_ptr(h, num), (rows.value, cols.value)).clone())
                bands[0] = deep[0]
                for l in range(self.deep_levels):
                    bands.append(tuple(deep[1 + 3 * l:4 + 3 * l]))
            else:
                bands += [None] * self.deep_levels
        self._bands = bands
        return self

    def _deep_plan(self, rows, cols):
        if self._deep is None:
            h = handle_t()
            rc = self._lib.pdwt_create_batched(None, 1, rows, cols, self.wname.encode("ASCII"), self.deep_levels, 1, 1, 0,
                                               0, 2, self.device.index, C.c_void_p(self._stream.cuda_stream), C.byref(h))
            check(rc, "TiledWavelets gathered plan", self._lib)
            info = PdwtInfo()
            check(self._lib.pdwt_get_info(h, C.byref(info), None, None, None, None), lib=self._lib)
            if info.nlevels != self.deep_levels:
                self._lib.pdwt_destroy(h)
                raise ValueError("TiledWavelets: %d levels requested, the image allows only %d"
                                 % (self.levels, self.tiled_levels + info.nlevels))
            self._deep = h
        return self._deep

    # ---- the ONE collective of the path: all ranks' slabs stacked in rank order
    def _gather_rows(self, slab):
        torch, dist = self._torch, self._dist
        if self.world == 1:
            return slab
        if self._via_host:
            parts = [torch.empty(slab.shape, dtype=slab.dtype) for _ in range(self.world)]
            dist.all_gather(parts, slab.cpu().contiguous(), group=self.group)
            return torch.cat(parts).to(self.device)
        parts = [torch.empty_like(slab) for _ in range(self.world)]
        dist.all_gather(parts, slab.contiguous(), group=self.group)
        return torch.cat(parts)

    def _scatter_rows(self, full, slab_shape):
        """rank 0 holds `full`; every rank gets its slab (one broadcast, each rank slices)."""
        torch, dist = self._torch, self._dist
        if self.world == 1:
            return full
        rows = slab_shape[0] * self.world
        if self._via_host:
            buf = full.cpu().contiguous() if self.rank == 0 else torch.empty((rows, slab_shape[1]), dtype=torch.float32)
            dist.broadcast(buf, 0, group=self.group)
            return buf[self.rank * slab_shape[0]:(self.rank + 1) * slab_shape[0]].to(self.device)
        buf = full.contiguous() if self.rank == 0 else torch.empty((rows, slab_shape[1]), dtype=torch.float32,
                                                                    device=self.device)
        dist.broadcast(buf, 0, group=self.group)
        return buf[self.rank * slab_shape[0]:(self.rank + 1) * slab_shape[0]].clone()

    def _inverse(self):
        if self._bands is None:
            raise RuntimeError("TiledWavelets.inverse: call forward() first")
        if self.do_swt:
            return self._inverse_swt()
        torch, hq = self._torch, self._hq
        cur = self._bands[0]
        if self.deep_levels:
            full = None
            if self.rank == 0:  # undo the gathered levels, then hand the slabs of A_t back
                h = self._deep
                check(self._lib.pdwt_set_coeff(h, C.c_void_p(self._bands[0].contiguous().data_ptr()), 0, 1), lib=self._lib)
                for l in range(self.deep_levels):
                    for k in range(3):
                        b = self._bands[self.tiled_levels + 1 + l][k]
                        self._view(self._lib.pdwt_coeff_ptr(h, 1 + 3 * l + k), tuple(b.shape)).copy_(b)
                check(self._lib.pdwt_inverse(h), "TiledWavelets.inverse (gathered levels)", self._lib)
                full = self._view(self._lib.pdwt_image_ptr(h), (self._a_slab_shape[0] * self.world, self._a_slab_shape[1]))
            cur = self._scatter_rows(full, self._a_slab_shape)
        for lvl in range(self.tiled_levels, 0, -1):
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
        """[A, [H1, V1, D1], [H2, V2, D2], ...] like Wavelets.coeffs.  Levels 1 .. tiled_levels: the row slab of this
        rank.  Gathered levels (and A): the WHOLE band on rank 0, None on the other ranks."""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.coeffs: call forward() first")
        self._torch.cuda.synchronize(self.device)
        A = None if self._bands[0] is None else self._bands[0].cpu().numpy()
        return [A] + [None if lvl is None else [b.cpu().numpy() for b in lvl] for lvl in self._bands[1:]]

    def cleanup(self):
        for h in self._plans.values():
            self._lib.pdwt_destroy(h)
        self._plans = {}
        if getattr(self, "_deep", None):
            self._lib.pdwt_destroy(self._deep)
            self._deep = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
