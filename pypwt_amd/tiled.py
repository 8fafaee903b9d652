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


class _LevelPlan(object):
    """One single-GPU plan on an extended slab, with cached zero-copy views of its device buffers."""

    def __init__(self, owner, rows, cols, levels, do_swt):
        lib = owner._lib
        h = handle_t()
        rc = lib.pdwt_create_batched(None, 1, rows, cols, owner.wname.encode("ASCII"), levels, 1, 1, 0, do_swt, 2,
                                     owner.device.index, C.c_void_p(owner._stream.cuda_stream), C.byref(h))
        check(rc, "TiledWavelets plan", lib)
        info = PdwtInfo()
        check(lib.pdwt_get_info(h, C.byref(info), None, None, None, None), lib=lib)
        self.h, self.levels, self.lib = h, int(info.nlevels), lib
        self.img = owner._view(lib.pdwt_image_ptr(h), (rows, cols))
        # the coefficient bands lie back to back in one region (pdwt_coeff_region): one flat view, one view per band, and --
        # when all bands have one shape (a single decimated level, every undecimated plan) -- one (band, row, column) view
        # whose row ranges are the halos of ALL bands: one message per neighbour instead of one per band
        nb = 3 * self.levels + 1
        offs = (C.c_longlong * nb)()
        total = int(lib.pdwt_coeff_region(h, offs, nb))
        region = owner._view(lib.pdwt_coeff_ptr(h, 0), (total,))
        self.co, shapes = [], []
        r, c = C.c_int(), C.c_int()
        for num in range(nb):
            lib.pdwt_coeff_count(h, num, C.byref(r), C.byref(c))
            shapes.append((r.value, c.value))
            self.co.append(region[offs[num]:offs[num] + r.value * c.value].view(r.value, c.value))
        step = int(offs[1] - offs[0])
        same = all(sh == shapes[0] for sh in shapes) and all(int(offs[k]) == k * step for k in range(nb))
        self.stack = region.as_strided((nb,) + shapes[0], (step, shapes[0][1], 1)) if same else None

    @staticmethod
    def halo_pieces(t, H, h, m):
        """rows [H, H + m) of `t` (rows on its last-but-one axis) are this rank's, the h rows on either side of them the
        halo the level needs (the margin beyond, H - h rows per side, is never read for a row that is kept); the four row
        ranges of an exchange: (top rows, bottom rows, halo above, halo below)"""
        return (t[..., H:H + h, :], t[..., H + m - h:H + m, :], t[..., H - h:H, :], t[..., H + m:H + m + h, :])

    def destroy(self):
        if self.h is not None:
            self.lib.pdwt_destroy(self.h)
            self.h = None


class TiledWavelets(object):
    """Data layout: the slab and every band slab live INSIDE the buffers of the single-GPU plans that work on them
    (the interior rows of an extended slab); the halo rows around them are received straight into the same buffers.
    A level costs its kernel and one halo exchange -- no staging tensors and (round 4) no copy of the approximation between
    levels: the image of level l + 1's plan IS band 0 of level l's plan (pdwt_bind_image).  For that the margins shrink
    geometrically: level l's slab is extended by H_l = hp 2^(t - l) rows per side (t = tiled levels), its bands by H_l / 2 =
    H_(l+1), so the geometries chain; only the hp (hq) rows next to the interior are exchanged and read, the rest of the
    margin is room.  `coeffs` / `image` copy to the host when asked.

    Aliasing: `slab` and the tensors of `device_coeffs` are zero-copy VIEWS of plan buffers.  They are overwritten by
    the next forward() / inverse() (clone what must survive), `slab` is None after cleanup(), and editing the
    coefficient views is the intended way to threshold between forward() and inverse().  inverse() runs once per
    forward(): a second call warns and does nothing (the reference's W_INVERSE state) unless mark_coeffs_current()
    has re-armed it after an in-place edit."""

    def __init__(self, slab, wname, levels, group=None, do_swt=0, loopback=False, comm=None):
        """loopback: with ONE rank, still send the halos / gather / broadcast through the process group (the rank
        is its own neighbour) instead of copying them -- the way to run the RCCL transport on a one-GPU box.
        comm: a pypwt_amd.comm.Communicator -- the halos, the gather and the broadcast then go through the library's own
        RCCL calls on the plans' stream (pdwt_comm_exchange: one grouped send / receive per level, a few microseconds of host
        time instead of the ~100 us of torch.distributed.batch_isend_irecv); rank and world size are the communicator's
        and no torch process group is needed.  A communicator of one rank is its own neighbour (loopback)."""
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
        self._comm = comm
        if comm is not None:
            self.world, self.rank = comm.size, comm.rank
            self._via_host = False
            self._loopback = self.world == 1
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self._via_host = dist.is_initialized() and dist.get_backend(group) != "nccl"  # gloo: stage halos on the host
            self._loopback = bool(loopback) and dist.is_initialized() and self.world == 1
        self.device = torch.device("cuda", torch.cuda.current_device())
        src = torch.as_tensor(np.ascontiguousarray(slab, dtype=np.float32) if isinstance(slab, np.ndarray) else slab)
        if src.dim() != 2:
            raise ValueError("TiledWavelets: the slab must be a 2D array (rows of this rank x all columns)")
        self.n, self.Nc = int(src.shape[0]), int(src.shape[1])
        self.wname = str(wname)
        buf = (C.c_float * 160)()
        hlen = self._lib.pdwt_wavelet_filters(self.wname.encode("ASCII"), buf, 160)
        check(hlen, "TiledWavelets()", self._lib)
        self.hlen = int(hlen)
        self.levels = int(levels)
        self.do_swt = int(bool(do_swt))
        self._deep = None    # rank 0: the single-GPU plan of the gathered approximation
        self._plans = {}
        self._piece_cache = {}
        self._bands = None   # after forward(): [A_L, (H1,V1,D1), ...] views of the plans' buffers
        self._in_coeff_domain = False
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
            P = self._swt_plan()
            self.slab = P.img[self._hs:self._hs + self.n]
        else:
            if self.levels < 1 or self.Nc % (1 << self.levels):
                raise ValueError("TiledWavelets: columns (%d) must be divisible by 2^levels" % self.Nc)
            c = self.hlen // 2 - 1
            self._hp = c + (c & 1)                        # analysis halo rows (even)
            H2 = self.hlen // 2
            C2, S = H2 // 2, (0 if (H2 & 1) else 1)
            self._hq = max(C2, H2 - 1 - C2 + S)           # synthesis halo rows (coefficient rows); always hp / 2
            assert 2 * self._hq == self._hp
            # levels that run as slabs: the slab entering the level is even, at least as tall as the analysis halo,
            # and its half at least as tall as the synthesis halo; the rest runs on rank 0 after the gather
            t, m = 0, self.n
            while t < self.levels and m % 2 == 0 and m >= self._hp and (m >> 1) >= self._hq and m >= 2:
                t, m = t + 1, m >> 1
            if t < 1:
                raise ValueError("TiledWavelets: a slab of %d rows is too thin (or odd) for one level of %s (halo %d)"
                                 % (self.n, self.wname, max(self._hp, 2 * self._hq)))
            self.tiled_levels, self.deep_levels = t, self.levels - t
            P = self._level(1)
            H1 = self._margin(1)
            self.slab = P.img[H1:H1 + self.n]   # the slab lives in the interior of level 1's plan
        with self._on_stream():
            self.slab.copy_(src.to(self.device, dtype=torch.float32, non_blocking=False))

    # ---- plans, one per level.  Level l (1-based) of the decimated transform works on the slab of n / 2^(l-1) rows extended by
    # _margin(l) rows per side; its four outputs have half the rows, margins included -- and since the synthesis halo is half
    # the analysis halo the SAME plan undoes the level: one plan per level holds the slab's interior and the halos of both
    # directions.
    def _margin(self, l):
        """rows by which level l's slab is extended per side: hp 2^(t - l); its bands are extended by half of that"""
        return self._hp << (self.tiled_levels - l)

    def _level(self, l):
        """the one-level plan of level l (1-based): the slab of n / 2^(l-1) rows extended by _margin(l) rows per side.  From
        level 2 on its image is bound to band 0 of the level above (same geometry by construction): no copy in either
        direction."""
        if l not in self._plans:
            rows, cols = (self.n >> (l - 1)) + 2 * self._margin(l), self.Nc >> (l - 1)
            P = _LevelPlan(self, rows, cols, 1, 0)
            if P.levels != 1:
                P.destroy()
                raise ValueError("TiledWavelets: %d x %d is too small for one level of %s" % (rows, cols, self.wname))
            if l > 1:
                up = self._level(l - 1)
                assert tuple(up.co[0].shape) == (rows, cols), (tuple(up.co[0].shape), rows, cols)
                check(self._lib.pdwt_bind_image(P.h, C.c_void_p(up.co[0].data_ptr())), "TiledWavelets plan", self._lib)
                P.img = up.co[0]
            self._plans[l] = P
        return self._plans[l]

    def _pieces(self, P, what, H, h, m):
        """cached halo row ranges of a plan's image ("img") or band stack ("stack"): interior rows [H, H + m), halo h"""
        key = (id(P), what)
        if key not in self._piece_cache:
            self._piece_cache[key] = _LevelPlan.halo_pieces(getattr(P, what), H, h, m)
        return self._piece_cache[key]

    def _view(self, ptr, shape):
        return self._torch.as_tensor(_DeviceView(ptr, shape), device=self.device)

    # ---- ring exchange, in place.  pieces = [(top rows, bottom rows, halo above, halo below), ...]: this rank's top
    # rows go to the previous rank's "halo below", its bottom rows to the next rank's "halo above".
    def _exchange_into(self, pieces):
        torch, dist = self._torch, self._dist
        if self.world == 1 and not self._loopback:
            for top, bottom, above, below in pieces:  # the ring closes on itself: periodic image
                above.copy_(bottom)
                below.copy_(top)
            return
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        if self._comm is not None:
            # the library's own RCCL calls on the plans' stream.  Every message is a range of whole rows of ONE plane of a plan
            # buffer (contiguous): the halos of a band stack go as one message per band inside the same group -- sent from
            # and received into the buffers themselves, nothing packed, nothing copied.
            def msgs(t):
                if t.is_contiguous():
                    return [(t.data_ptr(), t.numel())]
                return [(t[k].data_ptr(), t[k].numel()) for k in range(t.shape[0])]  # (band, rows, columns): rows are whole
            sends = [(p_, n_, nxt) for pc in pieces for p_, n_ in msgs(pc[1])] + [(p_, n_, prev) for pc in pieces for p_, n_ in msgs(pc[0])]
            recvs = [(p_, n_, prev) for pc in pieces for p_, n_ in msgs(pc[2])] + [(p_, n_, nxt) for pc in pieces for p_, n_ in msgs(pc[3])]
            self._comm.exchange(sends, recvs, stream=torch.cuda.current_stream(self.device).cuda_stream)
            return
        if self._via_host:
            tops = torch.cat([p[0].reshape(-1) for p in pieces]).cpu()
            bots = torch.cat([p[1].reshape(-1) for p in pieces]).cpu()
            from_prev, from_next = torch.empty_like(bots), torch.empty_like(tops)
            reqs = [dist.isend(bots, nxt, group=self.group, tag=1), dist.isend(tops, prev, group=self.group, tag=2),
                    dist.irecv(from_prev, prev, group=self.group, tag=1),
                    dist.irecv(from_next, nxt, group=self.group, tag=2)]
            for r in reqs:
                r.wait()
            o = 0
            for top, bottom, above, below in pieces:
                k = above.numel()
                above.copy_(from_prev[o:o + k].view(above.shape))
                below.copy_(from_next[o:o + k].view(below.shape))
                o += k
            return
        # one grouped launch (ncclGroupStart/End): with two ranks both neighbours are the same peer, and
        # the k-th send to a peer matches its k-th receive, hence this order.  Every piece is a range of whole
        # rows of a plan buffer (contiguous): sent from and received into the buffers themselves.
        # The halos of a band STACK are strided: packed into / unpacked from one message by one copy each.
        land = lambda t: t if t.is_contiguous() else torch.empty(t.shape, dtype=t.dtype, device=t.device)
        above, below = [land(p[2]) for p in pieces], [land(p[3]) for p in pieces]
        ops = [dist.P2POp(dist.isend, p[1].contiguous(), nxt, self.group) for p in pieces]
        ops += [dist.P2POp(dist.isend, p[0].contiguous(), prev, self.group) for p in pieces]
        ops += [dist.P2POp(dist.irecv, t, prev, self.group) for t in above]
        ops += [dist.P2POp(dist.irecv, t, nxt, self.group) for t in below]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for p, a, b in zip(pieces, above, below):
            if a is not p[2]:
                p[2].copy_(a)
            if b is not p[3]:
                p[3].copy_(b)

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
            P = _LevelPlan(self, key[1], key[2], self.levels, 1)
            if P.levels != self.levels:
                P.destroy()
                raise ValueError("TiledWavelets: %d SWT levels requested, the slab allows only %d" % (self.levels, P.levels))
            self._plans[key] = P
        return self._plans[key]

    def _forward_swt(self):
        m, hs = self.n, self._hs
        P = self._swt_plan()
        self._exchange_into([self._pieces(P, "img", hs, hs, m)])
        check(self._lib.pdwt_forward(P.h), "TiledWavelets.forward (SWT)", self._lib)
        flat = [b[hs:hs + m] for b in P.co]
        self._bands = [flat[0]] + [tuple(flat[1 + 3 * l:4 + 3 * l]) for l in range(self.levels)]
        return self

    def _inverse_swt(self):
        m, hs = self.n, self._hs
        P = self._swt_plan()
        # the halo rows of all 3 levels + 1 bands, received into the bands' own buffers: one message per neighbour
        self._exchange_into([self._pieces(P, "stack", hs, hs, m)])
        check(self._lib.pdwt_inverse(P.h), "TiledWavelets.inverse (SWT)", self._lib)
        return self

    def _forward(self, slab=None):
        torch = self._torch
        if slab is not None:
            self.slab.copy_(torch.as_tensor(slab).to(self.device, dtype=torch.float32))
        if self.do_swt:
            self._forward_swt()
            self._in_coeff_domain = True
            return self
        hp, hq = self._hp, self._hq
        bands = [None]
        P = None
        for l in range(1, self.tiled_levels + 1):
            m, H = self.n >> (l - 1), self._margin(l)
            P = self._level(l)   # its image is band 0 of the level above: A is already where it is needed
            if hp:
                self._exchange_into([self._pieces(P, "img", H, hp, m)])
            check(self._lib.pdwt_forward(P.h), "TiledWavelets.forward", self._lib)
            bands.append(tuple(P.co[k][H // 2:H // 2 + m // 2] for k in (1, 2, 3)))
        cur = P.co[0][hq:hq + (self.n >> self.tiled_levels)]   # the last level's band margin is hp / 2 = hq
        bands[0] = cur
        if self.deep_levels:
            bands[0] = None
            self._a_slab_shape = tuple(cur.shape)
            full = self._gather_rows(cur)            # every rank receives it (all-gather); rank 0 uses it
            if self.rank == 0:
                D = self._deep_plan(int(full.shape[0]), int(full.shape[1]))
                D.img.copy_(full)
                check(self._lib.pdwt_set_image(D.h, C.c_void_p(D.img.data_ptr()), 1), lib=self._lib)  # marks the image current
                check(self._lib.pdwt_forward(D.h), "TiledWavelets.forward (gathered levels)", self._lib)
                bands[0] = D.co[0]
                for l in range(self.deep_levels):
                    bands.append(tuple(D.co[1 + 3 * l:4 + 3 * l]))
            else:
                bands += [None] * self.deep_levels
        self._bands = bands
        self._in_coeff_domain = True
        return self

    def _deep_plan(self, rows, cols):
        if self._deep is None:
            D = _LevelPlan(self, rows, cols, self.deep_levels, 0)
            if D.levels != self.deep_levels:
                D.destroy()
                raise ValueError("TiledWavelets: %d levels requested, the image allows only %d"
                                 % (self.levels, self.tiled_levels + D.levels))
            self._deep = D
        return self._deep

    # ---- the ONE collective of the path: all ranks' slabs stacked in rank order
    def _gather_rows(self, slab):
        torch, dist = self._torch, self._dist
        if self.world == 1 and not self._loopback:
            return slab
        if self._comm is not None:
            src = slab.contiguous()
            full = torch.empty((self.world * slab.shape[0],) + tuple(slab.shape[1:]), dtype=slab.dtype, device=self.device)
            self._comm.all_gather(src.data_ptr(), full.data_ptr(), src.numel(), stream=torch.cuda.current_stream(self.device).cuda_stream)
            return full
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
        if self.world == 1 and not self._loopback:
            return full
        rows = slab_shape[0] * self.world
        if self._comm is not None:
            buf = full.contiguous() if self.rank == 0 else torch.empty((rows, slab_shape[1]), dtype=torch.float32, device=self.device)
            self._comm.broadcast(buf.data_ptr(), buf.numel(), 0, stream=torch.cuda.current_stream(self.device).cuda_stream)
            return buf[self.rank * slab_shape[0]:(self.rank + 1) * slab_shape[0]]
        if self._via_host:
            buf = full.cpu().contiguous() if self.rank == 0 else torch.empty((rows, slab_shape[1]), dtype=torch.float32)
            dist.broadcast(buf, 0, group=self.group)
            return buf[self.rank * slab_shape[0]:(self.rank + 1) * slab_shape[0]].to(self.device)
        buf = full.contiguous() if self.rank == 0 else torch.empty((rows, slab_shape[1]), dtype=torch.float32,
                                                                    device=self.device)
        dist.broadcast(buf, 0, group=self.group)
        return buf[self.rank * slab_shape[0]:(self.rank + 1) * slab_shape[0]]

    def _inverse(self):
        if self._bands is None:
            raise RuntimeError("TiledWavelets.inverse: call forward() first")
        if not self._in_coeff_domain:
            # the reference's W_INVERSE state: a second inverse does nothing -- but says so (wt.cu:272-275)
            import warnings
            warnings.warn("TiledWavelets.inverse() has already been run: the image is current and nothing was done.  "
                          "After editing device_coeffs in place call mark_coeffs_current() to invert them.", RuntimeWarning)
            return self
        if self.do_swt:
            self._inverse_swt()
            self._in_coeff_domain = False
            return self
        hp, hq = self._hp, self._hq
        t = self.tiled_levels
        P = self._level(t)
        m2 = self.n >> t
        if self.deep_levels:
            full = None
            if self.rank == 0:  # undo the gathered levels, then hand the slabs of A_t back
                D = self._deep
                check(self._lib.pdwt_inverse(D.h), "TiledWavelets.inverse (gathered levels)", self._lib)
                full = D.img
            P.co[0][hq:hq + m2].copy_(self._scatter_rows(full, self._a_slab_shape))
        for l in range(t, 0, -1):
            m2 = self.n >> l
            P = self._level(l)
            if hq:
                self._exchange_into([self._pieces(P, "stack", self._margin(l) // 2, hq, m2)])
            check(self._lib.pdwt_inverse(P.h), "TiledWavelets.inverse", self._lib)   # writes A of the level above in place
        self._in_coeff_domain = False
        return self

    def mark_coeffs_current(self):
        """After inverse(): declare the coefficient buffers current again (the caller has edited `device_coeffs` in
        place), so that the next inverse() runs instead of being refused.  Nothing is copied: every plan is told that
        its band 0 was written in place (pdwt_set_coeff with the plan's own pointer)."""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.mark_coeffs_current: call forward() first")
        plans = list(self._plans.values()) + ([self._deep] if getattr(self, "_deep", None) else [])
        for P in plans:
            if P.h is not None:
                check(self._lib.pdwt_set_coeff(P.h, C.c_void_p(self._lib.pdwt_coeff_ptr(P.h, 0)), 0, 1),
                      "TiledWavelets.mark_coeffs_current", self._lib)
        self._in_coeff_domain = True
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

    @property
    def device_coeffs(self):
        """the same structure as zero-copy torch views of the plans' device buffers (valid until the next forward /
        inverse; writing to them -- thresholding, say -- before inverse() is the intended use)"""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.device_coeffs: call forward() first")
        return [self._bands[0]] + [None if lvl is None else list(lvl) for lvl in self._bands[1:]]

    def cleanup(self):
        self._bands = None
        self.slab = None
        self._piece_cache = {}
        for P in self._plans.values():
            P.destroy()
        self._plans = {}
        if getattr(self, "_deep", None):
            self._deep.destroy()
            self._deep = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
