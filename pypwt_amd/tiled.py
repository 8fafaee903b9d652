"""TiledWavelets: ONE 2D image split in row slabs over the GPUs of a node, one rank (process) per GPU.

The batched case (independent images, one plan per GPU) needs no communication and is what bench.py
measures.  This module is the other multi-GPU case of the north star: a single image too large for one
GPU.  Rank r of G owns rows [r n, (r+1) n) of a (G n) x Nc image and the matching row slab of every
sub-band.  The row pass of a level is local; the column filters reach `hlen/2 - 1` rows into the
neighbouring slabs (analysis) and at most `hlen/4 + 1` coefficient rows (synthesis), so each GROUP of levels does
one ring halo exchange with the two neighbours -- point-to-point send/recv (RCCL over xGMI: neighbour traffic
only, no collective) -- and then runs the ordinary single-GPU kernels on the slab extended by the halo rows,
keeping the interior of the result: the periodic wrap of the kernels only touches rows that are discarded.
The ring is periodic, like the transform (reference semantics pdwt/src/separable.cu:114-121).

Round 5: no torch.  The slabs, bands and halos are row ranges of the plans' own device buffers (`DeviceRows`: pointer + shape,
`__cuda_array_interface__` for whoever wants to wrap them); copies go through `pdwt_copy` on the plans' stream; the transport is
`pypwt_amd.comm.Communicator` (the library's RCCL calls, one grouped send / receive per exchange, enqueued on the same stream),
or -- for ranks that SHARE a GPU, which RCCL refuses: the tests on a one-GPU box -- `pypwt_amd.comm.HostRing` (TCP, staged on the
host); one rank without either closes the ring on itself with device-to-device copies.  And the last K slab levels run as ONE
K-level plan behind ONE exchange per direction (the library fuses the small levels inside it as it does in any plan): a group of K
levels needs hp (2^K - 1) valid rows beyond the slab, and its inverse q_j coefficient rows of its j-th level, q_1 = hq, q_(j+1) =
hq + ceil(q_j / 2) < 2 hq.  A thick slab (16384 rows, db4: K = 4, 96 rows of margin) does its whole transform behind two exchanges;
a thin one falls back to smaller groups, down to one level each.

After some levels a slab is thinner than the halo (or no longer divisible by two): the remaining
approximation band -- by then 4^-t of the image -- is GATHERED (one all-gather: the only
collective of the path), rank 0 finishes the transform with an ordinary single-GPU plan, and the inverse
hands the slabs back with one broadcast ("after ~log2(G) levels ... gather the remaining A
band onto one GPU").  `tiled_levels` says how many levels ran as slabs.

The undecimated transform (do_swt=1) needs no per-level exchange: an output row of level l depends on the image
rows within hlen (2^l - 1) of it, so ONE exchange of hs = hlen (2^levels - 1) image rows per side lets every rank run
the whole multi-level SWT plan on its extended slab and keep the interior rows of every band (the inverse: one
exchange of the same halo of all 3 levels + 1 bands).  SURVEY.md 8e's "halo x 2^(l-1)" summed over the levels.

Restrictions: separable 2D transforms, float32.  DWT: columns divisible by 2^levels, rows per rank divisible by
2^tiled_levels with tiled_levels >= 1.  SWT: rows per rank >= hlen (2^levels - 1).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import PdwtInfo, check, handle_t


class DeviceRows(object):
    """Rows [r0, r0 + rows) of a (total rows) x cols plane of a plan's device buffer: a contiguous, borrowed range.
    `get()` / `set(a)` copy to / from the host on the plans' stream; `__cuda_array_interface__` (version 3, with that stream)
    lets torch / cupy wrap it without a copy.  Valid until the owner's cleanup()."""

    def __init__(self, owner, ptr, rows, cols):
        self._owner = owner
        self.ptr, self.shape = int(ptr), (int(rows), int(cols))
        self.dtype = np.dtype(np.float32)

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": "<f4", "data": (self.ptr, False), "version": 3, "strides": None,
                "stream": self._owner._stream or None}

    @property
    def count(self):
        return self.shape[0] * self.shape[1]

    def rows(self, a, b):
        assert 0 <= a <= b <= self.shape[0], (a, b, self.shape)
        return DeviceRows(self._owner, self.ptr + 4 * a * self.shape[1], b - a, self.shape[1])

    def get(self):
        out = np.empty(self.shape, dtype=np.float32)
        self._owner._copy(out.ctypes.data, self.ptr, self.count, 2)
        return out

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != self.shape:
            raise ValueError("DeviceRows.set: shape %s, expected %s" % (a.shape, self.shape))
        self._owner._copy(self.ptr, a.ctypes.data, self.count, 1)


class _Plan(object):
    """One single-GPU plan on an extended slab: its image and its bands as DeviceRows of the plan's own buffers."""

    def __init__(self, owner, rows, cols, levels, do_swt):
        lib = owner._lib
        h = handle_t()
        rc = lib.pdwt_create_batched(None, 1, rows, cols, owner.wname.encode("ASCII"), levels, 1, 1, 0, do_swt, 2,
                                     owner.device, C.c_void_p(owner._stream), C.byref(h))
        check(rc, "TiledWavelets plan", lib)
        if not owner._stream:  # the first plan's private stream serves every later plan, every copy and the transport
            owner._stream = int(lib.pdwt_get_stream(h) or 0)
            owner._first = h
        info = PdwtInfo()
        check(lib.pdwt_get_info(h, C.byref(info), None, None, None, None), lib=lib)
        self.h, self.levels, self.lib = h, int(info.nlevels), lib
        self.img = DeviceRows(owner, lib.pdwt_image_ptr(h), rows, cols)
        self.co = []
        r, c = C.c_int(), C.c_int()
        for num in range(3 * self.levels + 1):
            lib.pdwt_coeff_count(h, num, C.byref(r), C.byref(c))
            self.co.append(DeviceRows(owner, lib.pdwt_coeff_ptr(h, num), r.value, c.value))

    def destroy(self):
        if self.h is not None:
            self.lib.pdwt_destroy(self.h)
            self.h = None


def _halo(t, H, h, m):
    """rows [H, H + m) of plane `t` are this rank's, the h rows on either side of them the halo a level group needs (the margin
    beyond is never read for a row that is kept): (top rows, bottom rows, halo above, halo below)"""
    return (t.rows(H, H + h), t.rows(H + m - h, H + m), t.rows(H - h, H), t.rows(H + m, H + m + h))


class TiledWavelets(object):
    """Data layout: the slab and every band slab live INSIDE the buffers of the single-GPU plans that work on them
    (the interior rows of an extended slab); the halo rows around them are received straight into the same buffers.
    A level group costs its kernels and one halo exchange -- no staging buffers and no copy of the approximation between
    groups: the image of the next group's plan IS band 0 of this group's plan (pdwt_bind_image).  For that the margins shrink
    geometrically: a group of K levels whose slab is extended by M rows per side leaves band 0 extended by M / 2^K, the next
    group's margin; only the rows next to the interior are exchanged and read, the rest of the margin is room.
    `coeffs` / `image` copy to the host when asked.

    Aliasing: `slab` and the entries of `device_coeffs` are zero-copy VIEWS (DeviceRows) of plan buffers.  They are overwritten
    by the next forward() / inverse() (copy what must survive), `slab` is None after cleanup(), and editing the coefficient
    views (`set`, or a wrapper around their __cuda_array_interface__) is the intended way to change coefficients between
    forward() and inverse().  inverse() runs once per forward(): a second call warns and does nothing (the reference's
    W_INVERSE state) unless mark_coeffs_current() has re-armed it after an in-place edit."""

    def __init__(self, slab, wname, levels, do_swt=0, comm=None, ring=None, device=None, fuse_last=None):
        """comm: a pypwt_amd.comm.Communicator -- halos, gather and broadcast are the library's RCCL calls on the plans' stream
        (a communicator of one rank is its own neighbour: the transport on a one-GPU box).  ring: a pypwt_amd.comm.HostRing --
        the same messages over TCP, staged on the host (ranks sharing one GPU).  Neither: ONE rank, the ring closes on itself.
        device: HIP device index (default LOCAL_RANK, else 0).  fuse_last: the most levels in the last slab group (None: as many
        as the slab's thickness allows; 1 = every level its own plan and exchange)."""
        self._lib = _lib.load()
        self._comm, self._ring = comm, ring
        if comm is not None and ring is not None:
            raise ValueError("TiledWavelets: give a Communicator or a HostRing, not both")
        t = comm if comm is not None else ring
        self.world, self.rank = (t.size, t.rank) if t is not None else (1, 0)
        self.device = int(device) if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
        if self._lib.pdwt_device_count() <= 0:
            raise RuntimeError("TiledWavelets needs a HIP device (there is no CPU implementation)")
        self.device %= max(1, self._lib.pdwt_device_count())
        src_dev = _device_source(slab)
        if src_dev is None:
            slab = np.ascontiguousarray(slab, dtype=np.float32)
            shape = slab.shape
        else:
            shape = src_dev[1]
        if len(shape) != 2:
            raise ValueError("TiledWavelets: the slab must be a 2D array (rows of this rank x all columns)")
        self.n, self.Nc = int(shape[0]), int(shape[1])
        self.wname = str(wname)
        buf = (C.c_float * 160)()
        hlen = self._lib.pdwt_wavelet_filters(self.wname.encode("ASCII"), buf, 160)
        check(hlen, "TiledWavelets()", self._lib)
        self.hlen = int(hlen)
        self.levels = int(levels)
        self.do_swt = int(bool(do_swt))
        self._stream, self._first = 0, None   # set by the first plan
        self._deep = None    # the single-GPU plan of the gathered approximation (every rank has one: it is the gather buffer)
        self._plans = {}
        self._prepared = {}
        self._bands = None   # after forward(): [A_L, (H1,V1,D1), ...] DeviceRows of the plans' buffers
        self._in_coeff_domain = False
        if self.do_swt:
            if self.levels < 1:
                raise ValueError("TiledWavelets: levels must be >= 1")
            self._hs = self.hlen * ((1 << self.levels) - 1)   # rows of each neighbour every band of this slab depends on
            if self._hs > self.n:
                raise ValueError("TiledWavelets: a slab of %d rows is thinner than the halo of a %d-level SWT with %s (%d rows)"
                                 % (self.n, self.levels, self.wname, self._hs))
            self.tiled_levels, self.deep_levels = self.levels, 0
            self.groups = [(1, self.levels)]
            P = self._swt_plan()
            self.slab = P.img.rows(self._hs, self._hs + self.n)
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
            # level groups (first level, levels in the group): one level each, then the last K levels as ONE plan behind ONE exchange
            # per direction.  K: the most levels whose halo the slab entering the group can give (hp (2^K - 1) rows; its 2^-K-th
            # part the widest coefficient halo of the inverse) at a margin that costs at most ~6 % more rows (2 hp 2^K / rows)
            kmax = t if fuse_last is None else max(1, min(t, int(fuse_last)))
            K = 1
            for k in range(2, kmax + 1):
                m_in = self.n >> (t - k)
                q = self._hq
                for _ in range(k - 1):
                    q = self._hq + (q + 1) // 2
                if m_in >= self._hp * ((1 << k) - 1) and (m_in >> k) >= q and 2 * (self._hp << k) <= max(0.0625 * m_in, 2 * self._hp * 4):
                    K = k
            self.groups = [(l, 1) for l in range(1, t - K + 1)] + [(t - K + 1, K)]
            # margins, from the last group upwards: a group of K levels needs hp (2^K - 1) valid rows beyond the slab in a
            # margin that 2^K divides; the group above has 2^(its K) times the margin of the group below
            self._M = {}
            M = (self._hp << K) if K >= 2 else self._hp
            for first, k in reversed(self.groups):
                self._M[first] = M
                if first > 1:
                    prev_k = [kk for ff, kk in self.groups if ff + kk == first][0]
                    M <<= prev_k
            P = self._group_plan(1)
            H1 = self._M[1]
            self.slab = P.img.rows(H1, H1 + self.n)   # the slab lives in the interior of the first group's plan
        if src_dev is None:
            self.slab.set(slab)
        else:
            self._lib.pdwt_sync_producer(self.device, None, 1)
            self._copy(self.slab.ptr, src_dev[0], self.slab.count, 0)

    # ---- plumbing
    def _copy(self, dst, src, count, kind):
        check(self._lib.pdwt_copy(self._first, C.c_void_p(dst), C.c_void_p(src), int(count), int(kind)), "TiledWavelets copy", self._lib)

    def synchronize(self):
        if self._first is not None:
            check(self._lib.pdwt_synchronize(self._first), lib=self._lib)

    def _new_plan(self, rows, cols, levels, do_swt):
        P = _Plan(self, rows, cols, levels, do_swt)
        if P.levels != levels:
            P.destroy()
            raise ValueError("TiledWavelets: %d x %d is too small for %d level(s) of %s" % (rows, cols, levels, self.wname))
        return P

    # ---- plans, one per level group.  The group starting at level l (1-based) works on the slab of n / 2^(l-1) rows extended by
    # _M[l] rows per side; the bands of its j-th level are extended by _M[l] / 2^j -- and since the synthesis halo is half the
    # analysis halo the SAME plan undoes the group: one plan per group holds the slab's interior and the halos of both directions.
    def _group_plan(self, first):
        if first not in self._plans:
            k = dict(self.groups)[first]
            rows, cols = (self.n >> (first - 1)) + 2 * self._M[first], self.Nc >> (first - 1)
            P = self._new_plan(rows, cols, k, 0)
            if first > 1:
                up_first = [ff for ff, kk in self.groups if ff + kk == first][0]
                up = self._group_plan(up_first)
                assert up.co[0].shape == (rows, cols), (up.co[0].shape, rows, cols)
                check(self._lib.pdwt_bind_image(P.h, C.c_void_p(up.co[0].ptr)), "TiledWavelets plan", self._lib)
                P.img = up.co[0]
            self._plans[first] = P
        return self._plans[first]

    # ---- ring exchange, in place.  pieces = [(top rows, bottom rows, halo above, halo below), ...]: this rank's top
    # rows go to the previous rank's "halo below", its bottom rows to the next rank's "halo above".
    def _exchange_into(self, key, pieces):
        if self._comm is not None:
            # the library's own RCCL calls on the plans' stream.  Every message is a range of whole rows of ONE plane of a plan
            # buffer (contiguous), sent from and received into the buffers themselves: nothing packed, nothing copied.  The k-th
            # send to a peer matches its k-th receive (with two ranks both neighbours are the same peer), hence this order.
            if key not in self._prepared:
                prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
                sends = [(p[1].ptr, p[1].count, nxt) for p in pieces] + [(p[0].ptr, p[0].count, prev) for p in pieces]
                recvs = [(p[2].ptr, p[2].count, prev) for p in pieces] + [(p[3].ptr, p[3].count, nxt) for p in pieces]
                self._prepared[key] = self._comm.prepare(sends, recvs)
            self._comm.exchange_prepared(self._prepared[key], stream=self._stream)
            return
        if self._ring is not None and self.world > 1:
            tops = b"".join(p[0].get().tobytes() for p in pieces)
            bots = b"".join(p[1].get().tobytes() for p in pieces)
            from_prev, from_next = self._ring.sendrecv(tops, bots)   # the previous rank's bottom rows, the next rank's top rows
            o = 0
            for top, bottom, above, below in pieces:
                k = 4 * above.count
                above.set(np.frombuffer(from_prev, dtype=np.float32, count=above.count, offset=o).reshape(above.shape))
                below.set(np.frombuffer(from_next, dtype=np.float32, count=below.count, offset=o).reshape(below.shape))
                o += k
            return
        for top, bottom, above, below in pieces:  # one rank: the ring closes on itself (periodic image)
            self._copy(above.ptr, bottom.ptr, bottom.count, 0)
            self._copy(below.ptr, top.ptr, top.count, 0)

    def forward(self, slab=None):
        if slab is not None:
            dev = _device_source(slab)
            if dev is None:
                self.slab.set(slab)
            else:
                self._lib.pdwt_sync_producer(self.device, None, 1)
                self._copy(self.slab.ptr, dev[0], self.slab.count, 0)
        if self.do_swt:
            self._forward_swt()
        else:
            self._forward_dwt()
        self._in_coeff_domain = True
        return self

    # ---- undecimated transform: the whole multi-level plan on the slab extended by hs rows per side
    def _swt_plan(self):
        if "swt" not in self._plans:
            self._plans["swt"] = self._new_plan(self.n + 2 * self._hs, self.Nc, self.levels, 1)
        return self._plans["swt"]

    def _forward_swt(self):
        m, hs = self.n, self._hs
        P = self._swt_plan()
        self._exchange_into("swt-fwd", [_halo(P.img, hs, hs, m)])
        check(self._lib.pdwt_forward(P.h), "TiledWavelets.forward (SWT)", self._lib)
        flat = [b.rows(hs, hs + m) for b in P.co]
        self._bands = [flat[0]] + [tuple(flat[1 + 3 * l:4 + 3 * l]) for l in range(self.levels)]

    def _inverse_swt(self):
        m, hs = self.n, self._hs
        P = self._swt_plan()
        # the halo rows of all 3 levels + 1 bands, received into the bands' own buffers: one grouped exchange
        self._exchange_into("swt-inv", [_halo(b, hs, hs, m) for b in P.co])
        check(self._lib.pdwt_inverse(P.h), "TiledWavelets.inverse (SWT)", self._lib)

    # ---- decimated transform
    def _forward_dwt(self):
        hp, hq = self._hp, self._hq
        bands = [None] * (1 + self.tiled_levels)
        P = None
        for first, k in self.groups:
            m, M = self.n >> (first - 1), self._M[first]
            P = self._group_plan(first)   # its image is band 0 of the group above: A is already where it is needed
            h = hp * ((1 << k) - 1)
            if h:
                self._exchange_into(("fwd", first), [_halo(P.img, M, h, m)])
            check(self._lib.pdwt_forward(P.h), "TiledWavelets.forward", self._lib)
            for j in range(1, k + 1):   # the group's j-th level: pdwt numbering puts the finest level's details first
                Mj, mj = M >> j, m >> j
                bands[first + j - 1] = tuple(P.co[3 * (j - 1) + b].rows(Mj, Mj + mj) for b in (1, 2, 3))
        first, k = self.groups[-1]
        Mk = self._M[first] >> k
        cur = P.co[0].rows(Mk, Mk + (self.n >> self.tiled_levels))
        bands[0] = cur
        if self.deep_levels:
            bands[0] = None
            D = self._deep_plan(self.world * cur.shape[0], cur.shape[1])
            self._gather_rows(cur, D.img)            # every rank receives it (all-gather); rank 0 uses it
            if self.rank == 0:
                check(self._lib.pdwt_set_image(D.h, C.c_void_p(D.img.ptr), 1), lib=self._lib)  # in place: marks the image current
                check(self._lib.pdwt_forward(D.h), "TiledWavelets.forward (gathered levels)", self._lib)
                bands[0] = D.co[0]
                for l in range(self.deep_levels):
                    bands.append(tuple(D.co[1 + 3 * l:4 + 3 * l]))
            else:
                bands += [None] * self.deep_levels
        self._bands = bands

    def _deep_plan(self, rows, cols):
        if self._deep is None:
            try:
                self._deep = self._new_plan(rows, cols, self.deep_levels, 0)
            except ValueError:
                raise ValueError("TiledWavelets: %d levels requested, the image allows fewer (%d ran as slabs)" % (self.levels, self.tiled_levels))
        return self._deep

    # ---- the ONE collective of the path: all ranks' slabs stacked in rank order
    def _gather_rows(self, slab, full):
        if self._comm is not None:
            self._comm.all_gather(slab.ptr, full.ptr, slab.count, stream=self._stream)
        elif self._ring is not None and self.world > 1:
            parts = self._ring.all_gather(slab.get().tobytes())
            full.set(np.frombuffer(b"".join(parts), dtype=np.float32).reshape(full.shape))
        else:
            self._copy(full.ptr, slab.ptr, slab.count, 0)

    def _scatter_rows(self, full, slab):
        """rank 0 holds `full` (the deep plan's image: the same buffer on every rank); every rank gets its slab (one broadcast,
        each rank keeps its rows)"""
        if self._comm is not None:
            self._comm.broadcast(full.ptr, full.count, 0, stream=self._stream)
        elif self._ring is not None and self.world > 1:
            data = self._ring.broadcast(full.get().tobytes() if self.rank == 0 else None, 0)
            if self.rank != 0:
                full.set(np.frombuffer(data, dtype=np.float32).reshape(full.shape))
        mine = full.rows(self.rank * slab.shape[0], (self.rank + 1) * slab.shape[0])
        self._copy(slab.ptr, mine.ptr, mine.count, 0)

    def inverse(self):
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
        hq = self._hq
        t = self.tiled_levels
        first, k = self.groups[-1]
        P = self._group_plan(first)
        if self.deep_levels:
            D = self._deep
            if self.rank == 0:  # undo the gathered levels, then hand the slabs of A_t back
                check(self._lib.pdwt_inverse(D.h), "TiledWavelets.inverse (gathered levels)", self._lib)
            Mk = self._M[first] >> k
            self._scatter_rows(D.img, P.co[0].rows(Mk, Mk + (self.n >> t)))
        for first, k in reversed(self.groups):
            m, M = self.n >> (first - 1), self._M[first]
            P = self._group_plan(first)
            if hq:
                # coefficient rows the group's inverse reads beyond the slab: hq of its first level's details; to rebuild the
                # first level's approximation that far out, hq + ceil(hq / 2) of the second level's bands (A included)
                pieces, q = [], hq
                for j in range(1, k + 1):
                    Mj, mj = M >> j, m >> j
                    nums = [3 * (j - 1) + b for b in (1, 2, 3)] + ([0] if j == k else [])
                    pieces += [_halo(P.co[num], Mj, q, mj) for num in nums]
                    q = hq + (q + 1) // 2
                self._exchange_into(("inv", first), pieces)
            check(self._lib.pdwt_inverse(P.h), "TiledWavelets.inverse", self._lib)   # writes A of the group above in place
        self._in_coeff_domain = False
        return self

    def mark_coeffs_current(self):
        """After inverse(): declare the coefficient buffers current again (the caller has edited `device_coeffs` in
        place), so that the next inverse() runs instead of being refused.  Nothing is copied: every plan is told that
        its band 0 was written in place (pdwt_set_coeff with the plan's own pointer)."""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.mark_coeffs_current: call forward() first")
        plans = list(self._plans.values()) + ([self._deep] if self._deep is not None else [])
        for P in plans:
            if P.h is not None:
                check(self._lib.pdwt_set_coeff(P.h, C.c_void_p(self._lib.pdwt_coeff_ptr(P.h, 0)), 0, 1),
                      "TiledWavelets.mark_coeffs_current", self._lib)
        self._in_coeff_domain = True
        return self

    # ---- results (this rank's slabs)
    @property
    def image(self):
        return self.slab.get()

    @property
    def coeffs(self):
        """[A, [H1, V1, D1], [H2, V2, D2], ...] like Wavelets.coeffs.  Levels 1 .. tiled_levels: the row slab of this
        rank.  Gathered levels (and A): the WHOLE band on rank 0, None on the other ranks."""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.coeffs: call forward() first")
        A = None if self._bands[0] is None else self._bands[0].get()
        return [A] + [None if lvl is None else [b.get() for b in lvl] for lvl in self._bands[1:]]

    @property
    def device_coeffs(self):
        """the same structure as zero-copy DeviceRows views of the plans' device buffers (valid until the next forward /
        inverse; writing to them -- thresholding, say -- before inverse() is the intended use)"""
        if self._bands is None:
            raise RuntimeError("TiledWavelets.device_coeffs: call forward() first")
        return [self._bands[0]] + [None if lvl is None else list(lvl) for lvl in self._bands[1:]]

    def cleanup(self):
        self._bands = None
        self.slab = None
        self._prepared = {}
        first = self._first
        later = [P for P in list(self._plans.values()) + ([self._deep] if self._deep is not None else []) if P.h is not None]
        if later:
            self.synchronize()
        for P in later:   # the plan whose stream the others borrowed goes last
            if first is None or P.h.value != first.value:
                P.destroy()
        for P in later:
            P.destroy()
        self._plans = {}
        self._deep = None
        self._first = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def _device_source(a):
    """(pointer, shape) of a float32 C-contiguous device array (``__cuda_array_interface__``), else None"""
    cai = getattr(a, "__cuda_array_interface__", None)
    if cai is None:
        return None
    if np.dtype(cai["typestr"]) != np.float32 or cai.get("strides"):
        raise ValueError("TiledWavelets: a device slab must be C-contiguous float32")
    return int(cai["data"][0]), tuple(int(x) for x in cai["shape"])
