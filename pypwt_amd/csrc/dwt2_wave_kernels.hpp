// dwt2_wave_kernels.hpp -- register-resident 2D DWT level kernels for gfx950: ONE WAVEFRONT PER TILE,
// no LDS, no workgroup barrier.
//
// Replaces, for short even filters (hlen <= 10) on 16-B aligned rows, the LDS-tiled level kernels of
// dwt2_fast_kernels.hpp (reference: w_kern_forward_pass1/2, pdwt/src/separable.cu:91-176, and
// w_kern_inverse_pass1/2, :246-328).  What rocprofv3 said about the LDS tiles (profiles/r01h_*): as many
// scalar as vector instructions, 2.4x the vector instructions the arithmetic needs, 41 % of the wave
// cycles stalled on instruction issue, and 4.5 us for a level of 1 MB -- three barrier-separated phases
// per tile are a latency chain that small levels cannot hide.  Here a wavefront walks DOWN a strip of
// 256 image columns (forward; 128 coefficient columns in the inverse):
//   * every lane owns 4 adjacent columns: one coalesced 16-B load per image row (1 KiB per wavefront),
//     issued NR-1 rows ahead into a register ring, so the loads of the next rows are in flight while the
//     current row is filtered;
//   * the ROW filter needs hlen/2-1 samples of each neighbouring lane: wave-level shifts by one lane
//     (DPP wave_shr:1 / wave_shl:1, a register-to-register move), the two edge lanes take theirs from
//     a second, two-address load of the strip's halo columns (periodic wrap = address arithmetic);
//   * the COLUMN filter is a running sum: a filtered row is added, with the matching tap, to the hlen/2
//     output rows it contributes to (packed v_pk_fma_f32 on interleaved (L,H) pairs, taps in SGPRs); a
//     finished output row is stored and its accumulators restart.  The hlen-2 rows two vertically
//     adjacent segments share are re-read by both (L2 hits: segments of one XCD are vertical neighbours).
// Nothing is staged in LDS and no wavefront ever waits for another one.  In the steady state every
// vector-memory instruction is unconditional, so hipcc's s_waitcnt vmcnt(N) are exact counts and the
// ring really stays NR-1 rows deep (a load or store under a branch makes the counts collapse to 0).
//
// CPU emulation (tests/cpu_emu): a wavefront is a loop over 64 lanes per phase; a phase ends where
// lanes read each other's registers (WaveReg::from_prev / from_next).
#pragma once

#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

// A per-lane register array of N values of type T, with access to the neighbouring lanes' copies.
#ifdef PDWT_CPU_EMU
#define PDWT_WAVE_LANES(lane) for (int lane = 0; lane < 64; ++lane)
template <class T, int N>
struct WaveReg {
    T v[64][N];
    T* mine(int lane) { return v[lane]; }
    // element idx of lane-1 / lane+1; lane 0 / lane 63 get `fill`
    T from_prev(int idx, int lane, T fill) const { return lane > 0 ? v[lane - 1][idx] : fill; }
    T from_next(int idx, int lane, T fill) const { return lane < 63 ? v[lane + 1][idx] : fill; }
    T from_prev2(int idx, int lane, T fill) const { return lane > 1 ? v[lane - 2][idx] : fill; }
    T from_next2(int idx, int lane, T fill) const { return lane < 62 ? v[lane + 2][idx] : fill; }
};
#else
#define PDWT_WAVE_LANES(lane) for (int lane = threadIdx.x & 63, pdwt_once_ = 1; pdwt_once_; pdwt_once_ = 0)
// DPP wave shifts by one lane (GFX9 family): lanes without a source lane keep `fill`
static __device__ __forceinline__ float dpp_from_prev(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
static __device__ __forceinline__ float dpp_from_next(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
// fp64 values move as two 32-bit halves
static __device__ __forceinline__ double dpp_from_prev(double v, double fill) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, fill);
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)a, 0x138, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(a >> 32), 0x138, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
static __device__ __forceinline__ double dpp_from_next(double v, double fill) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, fill);
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)a, 0x130, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(a >> 32), 0x130, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <class T, int N>
struct WaveReg {
    T v[N];
    __device__ __forceinline__ T* mine(int) { return v; }
    __device__ __forceinline__ T from_prev(int idx, int, T fill) const { return dpp_from_prev(v[idx], fill); }
    __device__ __forceinline__ T from_next(int idx, int, T fill) const { return dpp_from_next(v[idx], fill); }
    // two lanes away: the shift applied twice
    __device__ __forceinline__ T from_prev2(int idx, int, T fill) const {
        return dpp_from_prev(dpp_from_prev(v[idx], fill), fill);
    }
    __device__ __forceinline__ T from_next2(int idx, int, T fill) const {
        return dpp_from_next(dpp_from_next(v[idx], fill), fill);
    }
};
#endif

// Keeps hipcc's scheduler from hoisting the loads of all rows of an unrolled group to its top (which
// makes every ring slot live at once: 170-200 VGPRs instead of ~100): nothing moves across it.
#ifdef PDWT_CPU_EMU
#define PDWT_ROW_FENCE() ((void)0)
#else
#define PDWT_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

constexpr int kWaveMaxHlen = 10;  // one neighbouring lane on each side covers the row filter's support

constexpr unsigned kRealBytes = (unsigned)sizeof(real_t);

// 16-B / 8-B accesses (fp64: 32-B / 16-B) at (uniform base) + (per-lane unsigned byte offset): the form hipcc turns into
// global_load/store ... v_offset, s[base:base+1] (no 64-bit vector address arithmetic)
PDWT_DEVICE v4f wave_ld16(const real_t* base, unsigned byte_off) {
    return *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + byte_off);
}
PDWT_DEVICE void wave_st8(real_t* base, unsigned byte_off, real_t x, real_t y) {
    real2_t w;  // the HIP struct type, not the ext vector: its stores do not alias the row loads for the scheduler
    w.x = x;
    w.y = y;
    *reinterpret_cast<real2_t*>(reinterpret_cast<char*>(base) + byte_off) = w;
}

// ------------------------------------------------------------------------------------------------
// forward level: in (Nr, Nc) -> A, H, V, D (Nr2, Nc2);  Nc % 4 == 0, 16-B aligned rows
// ------------------------------------------------------------------------------------------------
struct FwdWaveArgs {
    const real_t* in;
    real_t *A, *H, *V, *D;
    int Nr, Nc, Nr2, Nc2;
    long long in_bstride, out_bstride;
    int strips;    // ceil(Nc / 256)
    int segs;      // ceil(Nr2 / seg_out)
    int seg_out;   // output rows per wavefront
    FilterBankI fb;  // (dec_lo, dec_hi)
};

template <int HLEN>
struct FwdWaveGeom {
    static constexpr int C = HLEN / 2 - 1;             // analysis centre = samples needed from each neighbour lane
    static constexpr int NS = HLEN / 2;                // output rows a filtered row contributes to
    // rows per unrolled group (the slot pattern's period) and ring slots (NR - 1 rows of loads in flight)
    static constexpr int GR = HLEN < 4 ? 4 : (HLEN == 6 ? 12 : HLEN);
#ifndef PDWT_FWD_RING_H8
#define PDWT_FWD_RING_H8 4  // ring slots of the hlen-8 forward kernel (wbench sweeps it)
#endif
    static constexpr int NR = HLEN == 10 ? 5 : (HLEN == 8 ? PDWT_FWD_RING_H8 : 4);
    static_assert(HLEN >= 2 && (HLEN & 1) == 0 && HLEN <= kWaveMaxHlen, "short even filters only");
    static_assert((GR / 2) % NS == 0 && GR % NR == 0, "slot / ring indices are static inside a group");
};

// the rows a wavefront reads, in order: image row of position n of its walk (periodic over the evenly
// extended height, the extension row repeats the last one: pdwt/src/separable.cu:114-121)
struct RowWalk {
    int yp, np, last, left;  // position in [0, np), period, Nr - 1, rows still to advance over
    PDWT_DEVICE void start(int y, int Nr, int nrows) {
        np = Nr + (Nr & 1);
        yp = true_mod(y, np);
        last = Nr - 1;
        left = nrows - 1;
    }
    // plain periodic walk over n rows (coefficient rows of the inverse)
    PDWT_DEVICE void start_periodic(int y, int n, int nrows) {
        np = n;
        yp = true_mod(y, n);
        last = n - 1;
        left = nrows - 1;
    }
    // image row to load now; then step (stays on the last needed row once the walk is over, so the
    // loads past the end re-read a cached row instead of being branched around)
    PDWT_DEVICE int next() {
        const int sy = yp < last ? yp : last;
        if (left > 0) {
            --left;
            ++yp;
            if (yp == np) yp = 0;
        }
        return sy;
    }
};

template <int HLEN>
struct FwdWaveState {
    using G = FwdWaveGeom<HLEN>;
    WaveReg<real_t, 4 * G::NR> ring;   // [slot][4]: own columns of the rows in flight
    WaveReg<real_t, 4 * G::NR> hring;  // halo columns (lanes 0..31: left of the strip, 32..63: right)
    WaveReg<v2f, 4 * G::NS> acc;     // per slot: (A,V) col 0, (A,V) col 1, (H,D) col 0, (H,D) col 1
    WaveReg<unsigned, 3> off;        // byte offsets: own float4, halo float4, own output pair
};

// load the next row of the walk into ring slot `slot`
template <int HLEN>
PDWT_DEVICE void fwd_wave_load(FwdWaveState<HLEN>& st, int slot, const real_t* in, int Nc, RowWalk& walk) {
    const real_t* row = in + (long long)walk.next() * Nc;
    PDWT_WAVE_LANES(lane) {
        const v4f x = wave_ld16(row, st.off.mine(lane)[0]);
        real_t* r = st.ring.mine(lane) + 4 * slot;
        r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w;
        if (FwdWaveGeom<HLEN>::C > 0) {
            const v4f h = wave_ld16(row, st.off.mine(lane)[1]);
            real_t* q = st.hring.mine(lane) + 4 * slot;
            q[0] = h.x; q[1] = h.y; q[2] = h.z; q[3] = h.w;
        }
    }
}

// Row `r` of the wavefront's walk (r = 0 is image row 2 oy0 - C) sits in ring slot SLOT: filter it along x,
// add it to the running column sums.  R1 = r & 1, IH = (r >> 1) mod NS (both static).  WARM: one of the
// first HLEN - 2 rows (contributes only to output rows >= 0; dmax = r >> 1).
template <int HLEN, int SLOT, int R1, int IH, bool WARM, int DMAX>
PDWT_DEVICE void fwd_wave_row(FwdWaveState<HLEN>& st, const FilterBankI& fb) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int C = G::C, NS = G::NS;
    PDWT_WAVE_LANES(lane) {
        real_t v[HLEN + 2];
        if (C > 0) {
            const real_t* hv = st.hring.mine(lane) + 4 * SLOT;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                v[t] = st.ring.from_prev(4 * SLOT + 4 - C + t, lane, hv[4 - C + t]);
                v[C + 4 + t] = st.ring.from_next(4 * SLOT + t, lane, hv[t]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[C + q] = st.ring.mine(lane)[4 * SLOT + q];
        v2f lh0 = mk2(0.f, 0.f), lh1 = mk2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const v2f tap = fb.t[HLEN - 1 - j];
            lh0 = fma2(bc(v[j]), tap, lh0);
            lh1 = fma2(bc(v[2 + j]), tap, lh1);
        }
        // filtered row r is tap j = (r & 1) + 2 d of the output rows i = (r >> 1) - d, d = 0 .. NS-1
        v2f* a = st.acc.mine(lane);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            if (WARM && d > DMAX) continue;  // output rows before the segment: another wavefront's
            constexpr int kNS = NS;
            const int slot = ((IH - d) % kNS + kNS) % kNS;
            const int j = R1 + 2 * d;
            const v2f tap = fb.t[HLEN - 1 - j];
            v2f* s = a + 4 * slot;
            const v2f z = mk2(0.f, 0.f);
            s[0] = fma2(lh0, bc(tap.x), j == 0 ? z : s[0]);
            s[1] = fma2(lh1, bc(tap.x), j == 0 ? z : s[1]);
            s[2] = fma2(lh0, bc(tap.y), j == 0 ? z : s[2]);
            s[3] = fma2(lh1, bc(tap.y), j == 0 ? z : s[3]);
        }
    }
}

// store the finished output row held in accumulator slot SLOT_OUT to row pointers (A,H,V,D) + rowoff
template <int HLEN, int SLOT_OUT, bool GUARD>
PDWT_DEVICE void fwd_wave_store(FwdWaveState<HLEN>& st, const FwdWaveArgs& a, long long rowoff, bool row_ok,
                                int x0) {
    PDWT_WAVE_LANES(lane) {
        const v2f* s = st.acc.mine(lane) + 4 * SLOT_OUT;
        const unsigned o = st.off.mine(lane)[2];
        if (!GUARD || (row_ok && (x0 >> 1) + 2 * lane < a.Nc2)) {
            wave_st8(a.A + rowoff, o, s[0].x, s[1].x);
            wave_st8(a.V + rowoff, o, s[0].y, s[1].y);
            wave_st8(a.H + rowoff, o, s[2].x, s[3].x);
            wave_st8(a.D + rowoff, o, s[2].y, s[3].y);
        }
    }
}

// One group of GR rows of the steady state (after the HLEN - 2 warm-up rows): rows r = HLEN - 2 + GR it + g.
template <int HLEN, bool GUARD, int G0 = 0>
PDWT_DEVICE void fwd_wave_group(FwdWaveState<HLEN>& st, const FwdWaveArgs& a, const real_t* in, RowWalk& walk,
                                long long& rowoff, int& out_left, int x0) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int NS = G::NS, GR = G::GR, NR = G::NR;
    if constexpr (G0 < GR) {
        constexpr int r = HLEN - 2 + G0;  // + GR it: slot indices do not depend on it
        constexpr int slot = r % NR;
        // the ring slot of row r - 1 is free again: request row r - 1 + NR
        fwd_wave_load<HLEN>(st, (r + NR - 1) % NR, in, a.Nc, walk);
        fwd_wave_row<HLEN, slot, (r & 1), ((r >> 1) % NS), false, 0>(st, a.fb);
        if constexpr (G0 & 1) {  // output row (r >> 1) - NS + 1 is complete
            constexpr int so = ((((r >> 1) - (NS - 1)) % NS) + NS) % NS;
            fwd_wave_store<HLEN, so, GUARD>(st, a, rowoff, out_left > 0, x0);
            rowoff += a.Nc2;
            --out_left;
        }
        PDWT_ROW_FENCE();
        fwd_wave_group<HLEN, GUARD, G0 + 1>(st, a, in, walk, rowoff, out_left, x0);
    }
}

template <int HLEN, int R = 0>
PDWT_DEVICE void fwd_wave_warmup(FwdWaveState<HLEN>& st, const FwdWaveArgs& a, const real_t* in, RowWalk& walk) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int NS = G::NS, NR = G::NR;
    if constexpr (R < HLEN - 2) {
        fwd_wave_load<HLEN>(st, (R + NR - 1) % NR, in, a.Nc, walk);
        fwd_wave_row<HLEN, R % NR, (R & 1), ((R >> 1) % NS), true, (R >> 1)>(st, a.fb);
        PDWT_ROW_FENCE();
        fwd_wave_warmup<HLEN, R + 1>(st, a, in, walk);
    }
}

// One wavefront: strip `strip` (image columns [256 strip, 256 strip + 256)), output rows
// [seg * seg_out, (seg + 1) * seg_out) of image bz.
// GUARD = false: the host guarantees whole strips (Nc % 256 == 0) and whole groups (seg_out and Nr2 multiples
// of GR / 2), so no store carries a predicate.
template <int HLEN, bool GUARD>
PDWT_DEVICE void dwt2_fwd_wave(const FwdWaveArgs& a, int strip, int seg, int bz) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int C = G::C, GR = G::GR, NR = G::NR;

    const int oy0 = seg * a.seg_out;
    int oy_end = oy0 + a.seg_out;
    if (oy_end > a.Nr2) oy_end = a.Nr2;
    int out_left = oy_end - oy0;
    if (out_left <= 0) return;
    const int nrows = 2 * out_left + HLEN - 2;  // image rows this wavefront filters
    const int x0 = strip * 256;
    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    long long rowoff = (long long)bz * a.out_bstride + (long long)oy0 * a.Nc2;

    FwdWaveState<HLEN> st;
    PDWT_WAVE_LANES(lane) {
        // lanes past the right image edge load the periodic continuation: their neighbours need it
        st.off.mine(lane)[0] = kRealBytes * (unsigned)wrap_periodic(x0 + 4 * lane, a.Nc);
        st.off.mine(lane)[1] = kRealBytes * (unsigned)wrap_periodic(lane < 32 ? x0 - 4 : x0 + 256, a.Nc);
        st.off.mine(lane)[2] = kRealBytes * (unsigned)((x0 >> 1) + 2 * lane);
    }
    RowWalk walk;
    walk.start(2 * oy0 - C, a.Nr, nrows);
    // rows 0 .. NR-2 in flight before the first one is used (row r lives in ring slot r % NR)
#pragma unroll
    for (int p = 0; p < NR - 1; ++p) fwd_wave_load<HLEN>(st, p, in, a.Nc, walk);
    fwd_wave_warmup<HLEN>(st, a, in, walk);

    const int ngroups = (out_left + GR / 2 - 1) / (GR / 2);
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) fwd_wave_group<HLEN, GUARD>(st, a, in, walk, rowoff, out_left, x0);
}

// ------------------------------------------------------------------------------------------------
// inverse level: A, H, V, D (Nrc, Ncc) -> out (Nr, Nc);  Nc == 2 Ncc, Ncc even, 16-B aligned output rows
//
// Synthesis along one axis (pdwt/src/separable.cu:246-328, restated as in dwt2_fast_kernels.hpp): with
// H2 = hlen/2, C = H2/2, S = 1 - (H2 & 1), coefficient "pair" K yields the two samples
//     out[2K + 2C - S]     = sum_j  lo[hlen-2-2j] a[K+j] + hi[hlen-2-2j] d[K+j]
//     out[2K + 2C - S + 1] = sum_j  lo[hlen-1-2j] a[K+j] + hi[hlen-1-2j] d[K+j]        j = 0 .. H2-1
// (indices periodic).  A wavefront walks DOWN a strip of 128 coefficient columns; a lane owns 2 of them
// and the 4 image columns below them.  Per coefficient row: ROW synthesis first, on the raw coefficients
// (so the samples of the neighbouring lanes and the strip's halo columns are plain loaded values), with
// the bands paired as (A,H), (V,D): u = (row-synthesised low-y band, high-y band) for 4 image columns;
// then the COLUMN synthesis as running sums over the H2 pairs of image rows the row contributes to,
// accumulators paired as (even row, odd row), taps paired as consecutive taps of one filter.  A finished
// pair of image rows leaves as two 16-B stores per lane (1 KiB contiguous per wavefront and row).
// ------------------------------------------------------------------------------------------------
struct InvWaveArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int strips;      // ceil(Ncc / 128)
    int segs;        // ceil(Nrc / seg_pairs)
    int seg_pairs;   // coefficient rows ("pairs" of image rows) per wavefront
    FilterBankI fb;  // (rec_lo[j], rec_hi[j])
    v2f pl[kWaveMaxHlen / 2], ph[kWaveMaxHlen / 2];  // (lo[h-2-2d], lo[h-1-2d]), (hi[h-2-2d], hi[h-1-2d])
};

template <int HLEN>
struct InvWaveGeom {
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int S = (H2 & 1) ? 0 : 1;
    static constexpr int NLEFT = C;                           // coefficient columns needed from the previous lane
    static constexpr int NRIGHT = S ? H2 - C : H2 - C - 1;    // ... and from the next lane
#ifndef PDWT_INV_RING_H8
#define PDWT_INV_RING_H8 4  // ring slots of the hlen-8 inverse kernel (wbench sweeps it)
#endif
    static constexpr int NR = (H2 == 3 || H2 == 5) ? H2 : (H2 == 4 ? PDWT_INV_RING_H8 : 4);  // ring slots
    static constexpr int GR = (H2 == 3 || H2 == 5) ? H2 : NR;  // rows per unrolled group; H2 | GR, NR | GR
    static_assert(HLEN >= 2 && (HLEN & 1) == 0 && HLEN <= kWaveMaxHlen, "short even filters only");
    static_assert(NLEFT <= 2 && NRIGHT <= 2 && GR % H2 == 0 && GR % NR == 0, "one neighbour lane per side");
};

PDWT_DEVICE v2f wave_ld8(const real_t* base, unsigned byte_off) {
    return *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(base) + byte_off);
}
PDWT_DEVICE void wave_st16(real_t* base, unsigned byte_off, real_t x, real_t y, real_t z, real_t w) {
    real4_t o;
    o.x = x; o.y = y; o.z = z; o.w = w;
    *reinterpret_cast<real4_t*>(reinterpret_cast<char*>(base) + byte_off) = o;
}

template <int HLEN>
struct InvWaveState {
    using G = InvWaveGeom<HLEN>;
    WaveReg<real_t, 8 * G::NR> ring;   // [slot][A0 A1 H0 H1 V0 V1 D0 D1]: the lane's two coefficient columns
    WaveReg<real_t, 8 * G::NR> hring;  // halo columns (lanes 0..31: left of the strip, 32..63: right)
    WaveReg<v2f, 4 * G::H2> acc;     // [slot][image column 0..3] = (even row, odd row)
    WaveReg<unsigned, 3> off;        // byte offsets: own float2, halo float2, own output float4
};

template <int HLEN>
PDWT_DEVICE void inv_wave_load(InvWaveState<HLEN>& st, int slot, const InvWaveArgs& a, long long boff, RowWalk& walk) {
    const long long ro = boff + (long long)walk.next() * a.Ncc;
    PDWT_WAVE_LANES(lane) {
        const unsigned o = st.off.mine(lane)[0], oh = st.off.mine(lane)[1];
        real_t* r = st.ring.mine(lane) + 8 * slot;
        const v2f xa = wave_ld8(a.A + ro, o), xh = wave_ld8(a.H + ro, o);
        const v2f xv = wave_ld8(a.V + ro, o), xd = wave_ld8(a.D + ro, o);
        r[0] = xa.x; r[1] = xa.y; r[2] = xh.x; r[3] = xh.y;
        r[4] = xv.x; r[5] = xv.y; r[6] = xd.x; r[7] = xd.y;
        if (InvWaveGeom<HLEN>::NLEFT + InvWaveGeom<HLEN>::NRIGHT > 0) {
            real_t* h = st.hring.mine(lane) + 8 * slot;
            const v2f ya = wave_ld8(a.A + ro, oh), yh = wave_ld8(a.H + ro, oh);
            const v2f yv = wave_ld8(a.V + ro, oh), yd = wave_ld8(a.D + ro, oh);
            h[0] = ya.x; h[1] = ya.y; h[2] = yh.x; h[3] = yh.y;
            h[4] = yv.x; h[5] = yv.y; h[6] = yd.x; h[7] = yd.y;
        }
    }
}

// Coefficient row n of the wavefront's walk sits in ring slot SLOT: row synthesis, then add it to the
// running column sums; N2 = n mod H2 (static).  WARM: one of the first H2 - 1 rows (DMAX = n).
template <int HLEN, int SLOT, int N2, bool WARM, int DMAX>
PDWT_DEVICE void inv_wave_row(InvWaveState<HLEN>& st, const InvWaveArgs& a) {
    using G = InvWaveGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, S = G::S, NL = G::NLEFT, NRT = G::NRIGHT;
    PDWT_WAVE_LANES(lane) {
        // (A,H) and (V,D) of the coefficient columns 2l-2 .. 2l+3 (index m); only m in [2-NL, 4+NRT) is used
        v2f pAH[6], pVD[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) pAH[m] = pVD[m] = mk2(0.f, 0.f);
        const real_t* c = st.ring.mine(lane) + 8 * SLOT;
        pAH[2] = mk2(c[0], c[2]); pAH[3] = mk2(c[1], c[3]);
        pVD[2] = mk2(c[4], c[6]); pVD[3] = mk2(c[5], c[7]);
        if (NL + NRT > 0) {
            const real_t* hv = st.hring.mine(lane) + 8 * SLOT;
            constexpr int B = 8 * SLOT;
#pragma unroll
            for (int t = 0; t < NL; ++t) {  // previous lane's column 2 - NL + t (its element e = 2 - NL + t)
                const int e = 2 - NL + t;
                pAH[e] = mk2(st.ring.from_prev(B + 0 + e, lane, hv[0 + e]), st.ring.from_prev(B + 2 + e, lane, hv[2 + e]));
                pVD[e] = mk2(st.ring.from_prev(B + 4 + e, lane, hv[4 + e]), st.ring.from_prev(B + 6 + e, lane, hv[6 + e]));
            }
#pragma unroll
            for (int t = 0; t < NRT; ++t) {  // next lane's column t
                pAH[4 + t] = mk2(st.ring.from_next(B + 0 + t, lane, hv[0 + t]), st.ring.from_next(B + 2 + t, lane, hv[2 + t]));
                pVD[4 + t] = mk2(st.ring.from_next(B + 4 + t, lane, hv[4 + t]), st.ring.from_next(B + 6 + t, lane, hv[6 + t]));
            }
        }
        // row synthesis: image columns q = 0..3 of this lane
        v2f u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool odd = q & 1;
            const bool use_te = S ? odd : !odd;                        // taps hlen-2-2j ("te") or hlen-1-2j ("to")
            const int m0 = (q >> 1) - C + 2 + ((S && odd) ? 1 : 0);
            v2f s = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const v2f tap = a.fb.t[use_te ? HLEN - 2 - 2 * j : HLEN - 1 - 2 * j];
                s = fma2(pAH[m0 + j], bc(tap.x), s);
                s = fma2(pVD[m0 + j], bc(tap.y), s);
            }
            u[q] = s;
        }
        // column synthesis: row n is element d of the pairs p = n - d
        v2f* acc = st.acc.mine(lane);
#pragma unroll
        for (int d = 0; d < H2; ++d) {
            if (WARM && d > DMAX) continue;  // pairs before the segment: another wavefront's
            constexpr int kH2 = H2;
            const int slot = ((N2 - d) % kH2 + kH2) % kH2;
            const v2f z = mk2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v2f t = fma2(bc(u[q].x), a.pl[d], d == 0 ? z : acc[4 * slot + q]);
                acc[4 * slot + q] = fma2(bc(u[q].y), a.ph[d], t);
            }
        }
    }
}

// store the finished pair of image rows held in accumulator slot SLOT_OUT
template <int HLEN, int SLOT_OUT, bool GUARD>
PDWT_DEVICE void inv_wave_store(InvWaveState<HLEN>& st, const InvWaveArgs& a, long long boff, int oy_e, int oy_o,
                                bool pair_ok, int kx0) {
    real_t* re = a.out + boff + (long long)oy_e * a.Nc;
    real_t* ro = a.out + boff + (long long)oy_o * a.Nc;
    PDWT_WAVE_LANES(lane) {
        const v2f* s = st.acc.mine(lane) + 4 * SLOT_OUT;
        const unsigned o = st.off.mine(lane)[2];
        const bool lane_ok = !GUARD || (pair_ok && 2 * kx0 + 4 * lane < a.Nc);
        if (lane_ok && (!GUARD || oy_e < a.Nr)) wave_st16(re, o, s[0].x, s[1].x, s[2].x, s[3].x);
        if (lane_ok && (!GUARD || oy_o < a.Nr)) wave_st16(ro, o, s[0].y, s[1].y, s[2].y, s[3].y);
    }
}

// image rows of the next pair: (oy_e, oy_o) advance by two, periodic over 2 Nrc
struct PairRows {
    int oy_e, period;
    PDWT_DEVICE void start(int K0, int C, int S, int Nrc) {
        period = 2 * Nrc;
        oy_e = true_mod(2 * K0 + 2 * C - S, period);
    }
    PDWT_DEVICE int even() const { return oy_e; }
    PDWT_DEVICE int odd() const { return oy_e + 1 == period ? 0 : oy_e + 1; }
    PDWT_DEVICE void step() {
        oy_e += 2;
        if (oy_e >= period) oy_e -= period;
    }
};

template <int HLEN, bool GUARD, int G0 = 0>
PDWT_DEVICE void inv_wave_group(InvWaveState<HLEN>& st, const InvWaveArgs& a, long long bin, long long bout,
                                RowWalk& walk, PairRows& rows, int& pairs_left, int kx0) {
    using G = InvWaveGeom<HLEN>;
    constexpr int H2 = G::H2, GR = G::GR, NR = G::NR;
    if constexpr (G0 < GR) {
        constexpr int n = H2 - 1 + G0;  // + GR it
        inv_wave_load<HLEN>(st, (n + NR - 1) % NR, a, bin, walk);
        inv_wave_row<HLEN, n % NR, n % H2, false, 0>(st, a);
        constexpr int so = (((n - (H2 - 1)) % H2) + H2) % H2;  // pair n - H2 + 1 is complete
        inv_wave_store<HLEN, so, GUARD>(st, a, bout, rows.even(), rows.odd(), pairs_left > 0, kx0);
        rows.step();
        --pairs_left;
        PDWT_ROW_FENCE();
        inv_wave_group<HLEN, GUARD, G0 + 1>(st, a, bin, bout, walk, rows, pairs_left, kx0);
    }
}

template <int HLEN, int N = 0>
PDWT_DEVICE void inv_wave_warmup(InvWaveState<HLEN>& st, const InvWaveArgs& a, long long bin, RowWalk& walk) {
    using G = InvWaveGeom<HLEN>;
    constexpr int H2 = G::H2, NR = G::NR;
    if constexpr (N < H2 - 1) {
        inv_wave_load<HLEN>(st, (N + NR - 1) % NR, a, bin, walk);
        inv_wave_row<HLEN, N % NR, N % H2, true, N>(st, a);
        PDWT_ROW_FENCE();
        inv_wave_warmup<HLEN, N + 1>(st, a, bin, walk);
    }
}

// One wavefront: strip `strip` (coefficient columns [128 strip, 128 strip + 128)), pairs
// [seg * seg_pairs, (seg + 1) * seg_pairs) of image bz.
// GUARD = false: the host guarantees whole strips (Ncc % 128 == 0), whole groups (seg_pairs and Nrc
// multiples of GR) and Nr == 2 Nrc.
template <int HLEN, bool GUARD>
PDWT_DEVICE void dwt2_inv_wave(const InvWaveArgs& a, int strip, int seg, int bz) {
    using G = InvWaveGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, S = G::S, GR = G::GR, NR = G::NR;

    const int K0 = seg * a.seg_pairs;
    int K_end = K0 + a.seg_pairs;
    if (K_end > a.Nrc) K_end = a.Nrc;
    int pairs_left = K_end - K0;
    if (pairs_left <= 0) return;
    const int nrows = pairs_left + H2 - 1;  // coefficient rows this wavefront reads
    const int kx0 = strip * 128;
    const long long bin = (long long)bz * a.in_bstride, bout = (long long)bz * a.out_bstride;

    InvWaveState<HLEN> st;
    PDWT_WAVE_LANES(lane) {
        st.off.mine(lane)[0] = kRealBytes * (unsigned)wrap_periodic(kx0 + 2 * lane, a.Ncc);
        st.off.mine(lane)[1] = kRealBytes * (unsigned)wrap_periodic(lane < 32 ? kx0 - 2 : kx0 + 128, a.Ncc);
        st.off.mine(lane)[2] = kRealBytes * (unsigned)(2 * kx0 + 4 * lane);
    }
    RowWalk walk;
    walk.start_periodic(K0, a.Nrc, nrows);
    PairRows rows;
    rows.start(K0, C, S, a.Nrc);
#pragma unroll
    for (int p = 0; p < NR - 1; ++p) inv_wave_load<HLEN>(st, p, a, bin, walk);
    inv_wave_warmup<HLEN>(st, a, bin, walk);

    const int ngroups = (pairs_left + GR - 1) / GR;
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) inv_wave_group<HLEN, GUARD>(st, a, bin, bout, walk, rows, pairs_left, kx0);
}

// Range-checked row stores through a buffer descriptor (explained with the two-level kernel below; also used by
// dwt1_reg_kernels.hpp and swt2_fused_kernels.hpp).
#ifdef PDWT_CPU_EMU
struct RowBuf {
    char* base;
    unsigned bytes;
};
PDWT_DEVICE RowBuf row_buf(real_t* row, unsigned row_bytes) { return RowBuf{(char*)row, row_bytes}; }
PDWT_DEVICE void row_st8(const RowBuf& r, unsigned off, real_t x, real_t y) {
    if (off < r.bytes) { real_t* p = reinterpret_cast<real_t*>(r.base + off); p[0] = x; p[1] = y; }
}
PDWT_DEVICE void row_st4(const RowBuf& r, unsigned off, real_t x) {
    if (off < r.bytes) *reinterpret_cast<real_t*>(r.base + off) = x;
}
#else
struct RowBuf {
    __amdgpu_buffer_rsrc_t rsrc;
};
static __device__ __forceinline__ RowBuf row_buf(real_t* row, unsigned row_bytes) {
    RowBuf r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(row, (short)0, (int)row_bytes, 0x00020000);
    return r;
}
typedef unsigned pdwt_u2 __attribute__((ext_vector_type(2)));
typedef unsigned pdwt_u4 __attribute__((ext_vector_type(4)));
// two values / one value of a row (fp32: 8 B / 4 B; fp64: 16 B / 8 B)
static __device__ __forceinline__ void row_st8(const RowBuf& r, unsigned off, float x, float y) {
    pdwt_u2 d;
    d.x = __builtin_bit_cast(unsigned, x);
    d.y = __builtin_bit_cast(unsigned, y);
    __builtin_amdgcn_raw_buffer_store_b64(d, r.rsrc, (int)off, 0, 0);
}
static __device__ __forceinline__ void row_st4(const RowBuf& r, unsigned off, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r.rsrc, (int)off, 0, 0);
}
static __device__ __forceinline__ void row_st8(const RowBuf& r, unsigned off, double x, double y) {
    const pdwt_u2 a = __builtin_bit_cast(pdwt_u2, x), b = __builtin_bit_cast(pdwt_u2, y);
    pdwt_u4 d;
    d.x = a.x; d.y = a.y; d.z = b.x; d.w = b.y;
    __builtin_amdgcn_raw_buffer_store_b128(d, r.rsrc, (int)off, 0, 0);
}
static __device__ __forceinline__ void row_st4(const RowBuf& r, unsigned off, double x) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pdwt_u2, x), r.rsrc, (int)off, 0, 0);
}
#endif

#ifndef PDWT_DOUBLE  // the two-level kernel exists in fp32 only
// ------------------------------------------------------------------------------------------------
// TWO forward levels in one wavefront: in (N0r, N0c) -> H1, V1, D1 (N0r/2, N0c/2) and A2, H2, V2, D2
// (N0r/4, N0c/4).  The approximation of the first level never leaves the registers: a finished A1 row
// (two columns per lane) is filtered along x with the samples of the lanes l-2 .. l+2 (DPP shifts, one
// level-2 column per lane) and added to level-2 running column sums.  No lane can borrow A1 columns
// from ANOTHER wavefront, so strips overlap: a wavefront still loads 256 image columns (lanes 0 .. 63),
// but only the lanes 2 .. 61 own outputs -- 240 image columns, 120 level-1 and 60 level-2 columns per
// strip -- and the A1 rows above and below the segment are recomputed (3 hlen - 6 extra image rows per
// segment).  What it buys: A1 (a quarter of the level-1 output) is neither written nor read back, and
// one launch disappears.  Every store goes through a buffer descriptor whose range check drops the
// lanes that own nothing (offset 0xFFFFFFFF) and the columns right of the image (num_records = one
// row): no store sits under a branch, so the s_waitcnt counts of the load ring stay exact.
// Needs N0r % 4 == 0 and N0c % 16 == 0 (even sizes at both levels: exact periodization).
// ------------------------------------------------------------------------------------------------
struct FwdWave2Args {
    const float* in;
    float *H1, *V1, *D1;
    float *A2, *H2, *V2, *D2;
    int N0r, N0c;
    long long in_bstride, l1_bstride, l2_bstride;
    int strips;     // ceil(N0c / 240)
    int segs;       // ceil(N0r / 4 / seg2_out)
    int seg2_out;   // level-2 output rows per wavefront
    FilterBankI fb;
};

// One band row as a range-checked store target: (uniform row pointer) + (per-lane byte offset); offsets
// >= the row's byte length (in particular kDropped) are dropped.  The descriptor is rebuilt per row from
// the row pointer -- a few scalar instructions -- because on gfx950 the range check of a raw buffer covers
// voffset + soffset, so the row cannot ride in soffset (measured: everything below row 0 was dropped).
constexpr unsigned kDropped = 0xFFFFFFFFu;  // a byte offset no descriptor accepts

template <int HLEN>
struct FwdWave2State {
    using G = FwdWaveGeom<HLEN>;
    WaveReg<float, 4 * G::NR> ring;
    WaveReg<float, 4 * G::NR> hring;
    WaveReg<v2f, 4 * G::NS> acc;    // level 1: per slot (A,V) col 0, (A,V) col 1, (H,D) col 0, (H,D) col 1
    WaveReg<float, 2> a1;           // the finished A1 row: this lane's two columns
    WaveReg<v2f, 2 * G::NS> acc2;   // level 2: per slot (A,V), (H,D) of this lane's column
    WaveReg<unsigned, 4> off;       // byte offsets: own float4, halo float4, level-1 pair, level-2 sample (or kDropped)
};

struct Fwd2Bufs {  // per-image base pointers and row lengths
    const float* in;
    float *H1, *V1, *D1, *A2, *H2, *V2, *D2;
    unsigned row1_bytes, row2_bytes;
};

template <int HLEN>
PDWT_DEVICE void fwd2_wave_load(FwdWave2State<HLEN>& st, int slot, const float* in, int Nc, RowWalk& walk) {
    const float* row = in + (long long)walk.next() * Nc;
    PDWT_WAVE_LANES(lane) {
        const v4f x = wave_ld16(row, st.off.mine(lane)[0]);
        float* r = st.ring.mine(lane) + 4 * slot;
        r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w;
        if (FwdWaveGeom<HLEN>::C > 0) {
            const v4f h = wave_ld16(row, st.off.mine(lane)[1]);
            float* q = st.hring.mine(lane) + 4 * slot;
            q[0] = h.x; q[1] = h.y; q[2] = h.z; q[3] = h.w;
        }
    }
}

// level-1 part of image row r (same arithmetic as fwd_wave_row)
template <int HLEN, int SLOT, int R1, int IH, bool WARM, int DMAX>
PDWT_DEVICE void fwd2_wave_row(FwdWave2State<HLEN>& st, const FilterBankI& fb) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int C = G::C, NS = G::NS;
    PDWT_WAVE_LANES(lane) {
        float v[HLEN + 2];
        if (C > 0) {
            const float* hv = st.hring.mine(lane) + 4 * SLOT;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                v[t] = st.ring.from_prev(4 * SLOT + 4 - C + t, lane, hv[4 - C + t]);
                v[C + 4 + t] = st.ring.from_next(4 * SLOT + t, lane, hv[t]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[C + q] = st.ring.mine(lane)[4 * SLOT + q];
        v2f lh0 = mk2(0.f, 0.f), lh1 = mk2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const v2f tap = fb.t[HLEN - 1 - j];
            lh0 = fma2(bc(v[j]), tap, lh0);
            lh1 = fma2(bc(v[2 + j]), tap, lh1);
        }
        v2f* a = st.acc.mine(lane);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            if (WARM && d > DMAX) continue;
            constexpr int kNS = NS;
            const int slot = ((IH - d) % kNS + kNS) % kNS;
            const int j = R1 + 2 * d;
            const v2f tap = fb.t[HLEN - 1 - j];
            v2f* s = a + 4 * slot;
            const v2f z = mk2(0.f, 0.f);
            s[0] = fma2(lh0, bc(tap.x), j == 0 ? z : s[0]);
            s[1] = fma2(lh1, bc(tap.x), j == 0 ? z : s[1]);
            s[2] = fma2(lh0, bc(tap.y), j == 0 ? z : s[2]);
            s[3] = fma2(lh1, bc(tap.y), j == 0 ? z : s[3]);
        }
    }
}

// A finished level-1 row sits in accumulator slot SLOT1: store its details (if this wavefront owns the row),
// then run it through level 2: it is A1 row j1 of the wavefront's walk, J1 = j1 & 1, IH2 = (j1 >> 1) mod NS.
template <int HLEN, int SLOT1, int J1, int IH2>
PDWT_DEVICE void fwd2_wave_a1row(FwdWave2State<HLEN>& st, const FwdWave2Args& a, const Fwd2Bufs& b, bool own1,
                                 long long row1, bool own2, long long row2) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int C = G::C, NS = G::NS;
    // a row this wavefront does not own gets an empty descriptor: every lane is dropped
    const unsigned bytes1 = own1 ? b.row1_bytes : 0u, bytes2 = own2 ? b.row2_bytes : 0u;
    const RowBuf bH1 = row_buf(b.H1 + row1, bytes1), bV1 = row_buf(b.V1 + row1, bytes1), bD1 = row_buf(b.D1 + row1, bytes1);
    PDWT_WAVE_LANES(lane) {
        const v2f* s = st.acc.mine(lane) + 4 * SLOT1;
        const unsigned o = st.off.mine(lane)[2];
        row_st8(bV1, o, s[0].y, s[1].y);
        row_st8(bH1, o, s[2].x, s[3].x);
        row_st8(bD1, o, s[2].y, s[3].y);
        st.a1.mine(lane)[0] = s[0].x;
        st.a1.mine(lane)[1] = s[1].x;
    }
    PDWT_WAVE_LANES(lane) {
        // level-2 row filter: output column of lane l needs A1 columns 2l - C .. 2l - C + HLEN - 1 (lane-relative)
        float v[HLEN];
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const int rel = j - C;             // A1 column relative to this lane's first
            const int dl = rel >= 0 ? rel / 2 : -((1 - rel) / 2);  // lane distance (floor division by 2)
            const int e = rel - 2 * dl;        // element 0 / 1 of that lane
            float x;
            if (dl == 0) x = st.a1.mine(lane)[e];
            else if (dl == -1) x = st.a1.from_prev(e, lane, 0.f);
            else if (dl == 1) x = st.a1.from_next(e, lane, 0.f);
            else if (dl == -2) x = st.a1.from_prev2(e, lane, 0.f);
            else x = st.a1.from_next2(e, lane, 0.f);
            v[j] = x;
        }
        v2f lh = mk2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < HLEN; ++j) lh = fma2(bc(v[j]), a.fb.t[HLEN - 1 - j], lh);
        v2f* c = st.acc2.mine(lane);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            constexpr int kNS = NS;
            const int slot = ((IH2 - d) % kNS + kNS) % kNS;
            const int j = J1 + 2 * d;
            const v2f tap = a.fb.t[HLEN - 1 - j];
            const v2f z = mk2(0.f, 0.f);
            c[2 * slot] = fma2(lh, bc(tap.x), j == 0 ? z : c[2 * slot]);
            c[2 * slot + 1] = fma2(lh, bc(tap.y), j == 0 ? z : c[2 * slot + 1]);
        }
        if (J1) {  // level-2 output row (j1 >> 1) - NS + 1 is complete
            constexpr int kNS = NS;
            const int so = ((IH2 - (NS - 1)) % kNS + kNS) % kNS;
            const unsigned o = st.off.mine(lane)[3];
            row_st4(row_buf(b.A2 + row2, bytes2), o, c[2 * so].x);
            row_st4(row_buf(b.V2 + row2, bytes2), o, c[2 * so].y);
            row_st4(row_buf(b.H2 + row2, bytes2), o, c[2 * so + 1].x);
            row_st4(row_buf(b.D2 + row2, bytes2), o, c[2 * so + 1].y);
        }
    }
}

// rows r = HLEN - 2 + 2 GR it + G0 of the steady state, G0 = 0 .. 2 GR - 1 (the level-2 pattern's period)
template <int HLEN, int G0 = 0>
PDWT_DEVICE void fwd2_wave_group(FwdWave2State<HLEN>& st, const FwdWave2Args& a, const Fwd2Bufs& b, RowWalk& walk, int& j1,
                                 int own1_lo, int own1_hi, int n2, int r1_first, int oy2_0) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int NS = G::NS, GR = G::GR, NR = G::NR;
    if constexpr (G0 < 2 * GR) {
        constexpr int r = HLEN - 2 + G0;
        fwd2_wave_load<HLEN>(st, (r + NR - 1) % NR, b.in, a.N0c, walk);
        fwd2_wave_row<HLEN, r % NR, (r & 1), ((r >> 1) % NS), false, 0>(st, a.fb);
        if constexpr (G0 & 1) {
            constexpr int so = ((((r >> 1) - (NS - 1)) % NS) + NS) % NS;
            constexpr int J = (G0 - 1) / 2;  // j1 = GR it + J
            const int i2 = ((j1 - (HLEN - 1)) >> 1);  // level-2 row finished by an odd j1
            const bool own1 = j1 >= own1_lo && j1 < own1_hi;
            const bool own2 = (J & 1) && i2 >= 0 && i2 < n2;
            const long long row1 = own1 ? (long long)(r1_first + j1) * (a.N0c >> 1) : 0;  // element offsets of the rows
            const long long row2 = own2 ? (long long)(oy2_0 + i2) * (a.N0c >> 2) : 0;
            fwd2_wave_a1row<HLEN, so, (J & 1), ((J >> 1) % NS)>(st, a, b, own1, row1, own2, row2);
            ++j1;
        }
        PDWT_ROW_FENCE();
        fwd2_wave_group<HLEN, G0 + 1>(st, a, b, walk, j1, own1_lo, own1_hi, n2, r1_first, oy2_0);
    }
}

template <int HLEN, int R = 0>
PDWT_DEVICE void fwd2_wave_warmup(FwdWave2State<HLEN>& st, const FwdWave2Args& a, const Fwd2Bufs& b, RowWalk& walk) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int NS = G::NS, NR = G::NR;
    if constexpr (R < HLEN - 2) {
        fwd2_wave_load<HLEN>(st, (R + NR - 1) % NR, b.in, a.N0c, walk);
        fwd2_wave_row<HLEN, R % NR, (R & 1), ((R >> 1) % NS), true, (R >> 1)>(st, a.fb);
        PDWT_ROW_FENCE();
        fwd2_wave_warmup<HLEN, R + 1>(st, a, b, walk);
    }
}

// One wavefront: strip `strip` (level-2 columns [60 strip, 60 strip + 60)), level-2 rows
// [seg * seg2_out, (seg + 1) * seg2_out) of image bz.
template <int HLEN>
PDWT_DEVICE void dwt2_fwd2_wave(const FwdWave2Args& a, int strip, int seg, int bz) {
    using G = FwdWaveGeom<HLEN>;
    constexpr int C = G::C, GR = G::GR, NR = G::NR;
    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2;
    const int oy2_0 = seg * a.seg2_out;
    int oy2_end = oy2_0 + a.seg2_out;
    if (oy2_end > N2r) oy2_end = N2r;
    const int n2 = oy2_end - oy2_0;
    if (n2 <= 0) return;
    const int nA1 = 2 * n2 + HLEN - 2;          // A1 rows this wavefront computes
    const int r1_first = 2 * oy2_0 - C;         // ... starting here (may be negative: periodic)
    const int nrows = 2 * nA1 + HLEN - 2;       // image rows it filters
    const int a0 = 120 * strip - 4;             // first A1 column of lane 0
    const int x0 = 2 * a0;                      // first image column of lane 0

    FwdWave2State<HLEN> st;
    PDWT_WAVE_LANES(lane) {
        unsigned* o = st.off.mine(lane);
        o[0] = 4u * (unsigned)wrap_periodic(x0 + 4 * lane, a.N0c);
        o[1] = 4u * (unsigned)wrap_periodic(lane < 32 ? x0 - 4 : x0 + 256, a.N0c);
        const bool owner = lane >= 2 && lane < 62;
        o[2] = owner ? 4u * (unsigned)(a0 + 2 * lane) : kDropped;          // columns >= N1c: dropped by the range check
        o[3] = owner ? 4u * (unsigned)(60 * strip + lane - 2) : kDropped;
    }
    Fwd2Bufs b;
    b.in = a.in + (long long)bz * a.in_bstride;
    b.H1 = a.H1 + (long long)bz * a.l1_bstride;
    b.V1 = a.V1 + (long long)bz * a.l1_bstride;
    b.D1 = a.D1 + (long long)bz * a.l1_bstride;
    b.A2 = a.A2 + (long long)bz * a.l2_bstride;
    b.H2 = a.H2 + (long long)bz * a.l2_bstride;
    b.V2 = a.V2 + (long long)bz * a.l2_bstride;
    b.D2 = a.D2 + (long long)bz * a.l2_bstride;
    b.row1_bytes = (unsigned)N1c * 4u;
    b.row2_bytes = (unsigned)N2c * 4u;

    RowWalk walk;
    walk.start_periodic(2 * r1_first - C, a.N0r, nrows);
#pragma unroll
    for (int p = 0; p < NR - 1; ++p) fwd2_wave_load<HLEN>(st, p, b.in, a.N0c, walk);
    fwd2_wave_warmup<HLEN>(st, a, b, walk);
    // A1 row j1 (0-based in the walk) is image-level row r1_first + j1 (periodic); the wavefront owns, i.e.
    // stores the details of, the rows [2 oy2_0, 2 oy2_end)
    const int own1_lo = C, own1_hi = C + 2 * n2;
    int j1 = 0;
    const int ngroups = (nA1 + GR - 1) / GR;
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) fwd2_wave_group<HLEN>(st, a, b, walk, j1, own1_lo, own1_hi, n2, r1_first, oy2_0);
    (void)N1r;
}

#endif  // !PDWT_DOUBLE

// wave-tile id -> (strip, seg): XCD x (workgroup ids b with b % 8 == x share an L2) gets a contiguous band
// of segment rows, so vertically adjacent segments re-read their shared rows from that XCD's own L2.
// Placement only affects speed.
PDWT_DEVICE bool wave_tile(int block, int wave, int waves_per_block, int strips, int segs, int& strip, int& seg) {
    const int total = strips * segs;
    const int nblk = (total + waves_per_block - 1) / waves_per_block;
    const int chunk = (nblk + 7) >> 3;
    const int bt = (block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || bt >= nblk) return false;
    const int t = bt * waves_per_block + wave;
    if (t >= total) return false;
    seg = t / strips;
    strip = t - seg * strips;
    return true;
}

#ifndef PDWT_CPU_EMU
// register budget: 4 wavefronts per SIMD with the default ring, fewer when a deeper ring is compiled in; the fp64
// build's rings and running sums take twice the registers (2 wavefronts per SIMD)
#ifdef PDWT_DOUBLE
constexpr int kFwdWaveBlocks = 2, kInvWaveBlocks = 1;
#else
constexpr int kFwdWaveBlocks = PDWT_FWD_RING_H8 > 4 ? 2 : 4, kInvWaveBlocks = PDWT_INV_RING_H8 > 4 ? 2 : 3;
#endif
template <int HLEN, bool GUARD, int NT>
__global__ void __launch_bounds__(NT, kFwdWaveBlocks) dwt2_fwd_wave_kernel(const FwdWaveArgs a) {
    int strip, seg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!wave_tile(blockIdx.x, wave, NT / 64, a.strips, a.segs, strip, seg)) return;
    dwt2_fwd_wave<HLEN, GUARD>(a, strip, seg, blockIdx.y);
}

#ifndef PDWT_DOUBLE
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT, 4) dwt2_fwd2_wave_kernel(const FwdWave2Args a) {
    int strip, seg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!wave_tile(blockIdx.x, wave, NT / 64, a.strips, a.segs, strip, seg)) return;
    dwt2_fwd2_wave<HLEN>(a, strip, seg, blockIdx.y);
}
#endif

template <int HLEN, bool GUARD, int NT>
__global__ void __launch_bounds__(NT, kInvWaveBlocks) dwt2_inv_wave_kernel(const InvWaveArgs a) {
    int strip, seg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!wave_tile(blockIdx.x, wave, NT / 64, a.strips, a.segs, strip, seg)) return;
    dwt2_inv_wave<HLEN, GUARD>(a, strip, seg, blockIdx.y);
}
#endif

}  // namespace pdwt
