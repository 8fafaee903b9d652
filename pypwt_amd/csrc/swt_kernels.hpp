// swt_kernels.hpp -- undecimated (a-trous) wavelet level kernels for gfx950.
//
// Level l dilates the taps by f = 2^(l-1).  Rows y, y+f, y+2f, ... form f
// independent "phase" sub-signals when f divides Nr, so the fused 2D kernels
// tile the image as TX contiguous columns x TY rows OF ONE PHASE: the column
// pass then needs a dense (hlen-1)-row halo whatever the dilation, and it runs
// out of LDS.  The row pass reads its dilated taps straight from global memory
// (each tap is a full coalesced row segment shifted by j*f; the re-reads hit
// L1/L2), so no x-halo has to be staged.  Row and column pass are fused: the
// intermediate low/high planes live only in LDS (the reference writes and
// re-reads two full-size temporaries per level, pdwt/src/separable.cu:507-512
// and :641-646).
//
// Semantics (restated in oracle/pdwt_oracle.c, pinned to pywt.swt/iswt):
//   analysis   out[g] = sum_j x[(g + (j-c) f) mod N] * filt[hlen-1-j]
//              (pdwt/src/separable.cu:409-493)
//   synthesis  out[g] = 1/2 sum_j a[(g + (j-c') f) mod N] * rlo[hlen-1-j] + d[..] * rhi[hlen-1-j],
//              c' = hlen/2   (pdwt/src/separable.cu:553-626)
// The inverse applies the row (x) synthesis first and the column (y) synthesis
// second; the two 1D operators commute, the reference runs them in the other
// order (difference: fp32 rounding only).
//
// When f does not divide Nr (pywt cannot do these sizes; the reference can) the
// host falls back to the direct one-pass kernels at the bottom, which are also
// the (batched) 1D SWT.
#pragma once

#include "kernels_common.hpp"
#include "swt_kernels_args.hpp"

namespace pdwt {

template <int TX, int TY>
constexpr int swt2d_lds_floats(int hlen) {
    return 2 * kMaxTaps + 2 * (TY + hlen - 1) * TX;
}

#define PDWT_STAGE_TAPS()                                   \
    real_t* sTaps = smem;                                    \
    const real_t* lo = a.fb.lo;                              \
    const real_t* hi = a.fb.hi;                              \
    if (HLEN == 0) {                                        \
        PDWT_FOR_THREADS(tid, NT) {                         \
            if (tid < kMaxTaps) {                           \
                sTaps[tid] = a.fb.lo[tid];                  \
                sTaps[kMaxTaps + tid] = a.fb.hi[tid];       \
            }                                               \
        }                                                   \
        PDWT_SYNC();                                        \
        lo = sTaps;                                         \
        hi = sTaps + kMaxTaps;                              \
    }

// by enumerates (row tile, phase): ph = by % f, it = by / f
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void swt2_fwd_tile(const Swt2DArgs& a, int bx, int by, int bz, real_t* smem) {
    static_assert(NT % TX == 0 && TY % (NT / TX) == 0, "tile/thread shape");
    constexpr int NG = NT / TX;
    constexpr int R = TY / NG;
    const int hlen = HLEN ? HLEN : a.hlen;
    const int c = analysis_centre(hlen);
    const int f = a.f;
    // rows of phase ph: ph, ph + f, ...; where f does not divide Nr the periodic extension of a phase runs into the other phases
    // (the dilated taps of row y are the rows (y + (j - c) f) mod Nr), so rows are wrapped as ROWS, not as phase indices
    const int ph = by % f;
    const int it = by / f;
    const int RY = TY + hlen - 1;

    PDWT_STAGE_TAPS();
    real_t* tL = smem + 2 * kMaxTaps;
    real_t* tH = tL + RY * TX;

    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.bstride;

    // ---- phase 1: dilated row analysis straight from global -> tL, tH (RY x TX)
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        const int x = bx * TX + k;
        for (int r = tid / TX; r < RY; r += NG) {
            real_t aL = 0.f, aH = 0.f;
            if (x < a.Nc) {
                const real_t* row = in + (long long)wrap_periodic(ph + f * (it * TY - c + r), a.Nr) * a.Nc;
#pragma unroll
                for (int j = 0; j < (HLEN > 0 ? HLEN : hlen); ++j) {
                    const real_t v = row[wrap_periodic(x + (j - c) * f, a.Nc)];
                    aL = pdwt_fma(v, lo[hlen - 1 - j], aL);
                    aH = pdwt_fma(v, hi[hlen - 1 - j], aH);
                }
            }
            tL[r * TX + k] = aL;
            tH[r * TX + k] = aH;
        }
    }
    PDWT_SYNC();

    // ---- phase 2: column analysis inside the phase, out of LDS
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        const int ty0 = (tid / TX) * R;
        const int x = bx * TX + k;
        const long long boff = (long long)bz * a.bstride;
        for (int i = 0; i < R; ++i) {
            real_t rA = 0.f, rH = 0.f, rV = 0.f, rD = 0.f;
#pragma unroll
            for (int j = 0; j < (HLEN > 0 ? HLEN : hlen); ++j) {
                const real_t l = tL[(ty0 + i + j) * TX + k];
                const real_t h = tH[(ty0 + i + j) * TX + k];
                const real_t tl = lo[hlen - 1 - j], th = hi[hlen - 1 - j];
                rA = pdwt_fma(l, tl, rA);
                rH = pdwt_fma(l, th, rH);
                rV = pdwt_fma(h, tl, rV);
                rD = pdwt_fma(h, th, rD);
            }
            const int si = it * TY + ty0 + i;
            if (ph + f * si < a.Nr && x < a.Nc) {
                const long long o = boff + (long long)(ph + f * si) * a.Nc + x;
                a.A[o] = rA;
                a.H[o] = rH;
                a.V[o] = rV;
                a.D[o] = rD;
            }
        }
    }
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void swt2_inv_tile(const Swt2DArgs& a, int bx, int by, int bz, real_t* smem) {
    static_assert(NT % TX == 0 && TY % (NT / TX) == 0, "tile/thread shape");
    constexpr int NG = NT / TX;
    constexpr int R = TY / NG;
    const int hlen = HLEN ? HLEN : a.hlen;
    const int c = hlen / 2;  // synthesis centre
    const int f = a.f;
    const int ph = by % f;
    const int it = by / f;
    const int RY = TY + hlen - 1;

    PDWT_STAGE_TAPS();
    real_t* u1 = smem + 2 * kMaxTaps;
    real_t* u2 = u1 + RY * TX;

    const long long boff = (long long)bz * a.bstride;

    // ---- phase 1: dilated row synthesis from global: u1 = Lx(A) + Hx(V), u2 = Lx(H) + Hx(D)
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        const int x = bx * TX + k;
        for (int r = tid / TX; r < RY; r += NG) {
            real_t r1 = 0.f, r2 = 0.f;
            if (x < a.Nc) {
                const long long ro = boff + (long long)wrap_periodic(ph + f * (it * TY - c + r), a.Nr) * a.Nc;
#pragma unroll
                for (int j = 0; j < (HLEN > 0 ? HLEN : hlen); ++j) {
                    const long long o = ro + wrap_periodic(x + (j - c) * f, a.Nc);
                    const real_t tl = lo[hlen - 1 - j], th = hi[hlen - 1 - j];
                    // a pending soft_threshold of the plan is applied here, on the fly, to the detail
                    // bands (never to A): the thresholded coefficients are not written back
                    r1 = pdwt_fma(a.A[o], tl, r1);
                    r1 = pdwt_fma(soft_shrink(a.V[o], a.soft_beta), th, r1);
                    r2 = pdwt_fma(soft_shrink(a.H[o], a.soft_beta), tl, r2);
                    r2 = pdwt_fma(soft_shrink(a.D[o], a.soft_beta), th, r2);
                }
            }
            u1[r * TX + k] = 0.5f * r1;
            u2[r * TX + k] = 0.5f * r2;
        }
    }
    PDWT_SYNC();

    // ---- phase 2: column synthesis inside the phase
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        const int ty0 = (tid / TX) * R;
        const int x = bx * TX + k;
        for (int i = 0; i < R; ++i) {
            real_t r = 0.f;
#pragma unroll
            for (int j = 0; j < (HLEN > 0 ? HLEN : hlen); ++j) {
                r = pdwt_fma(u1[(ty0 + i + j) * TX + k], lo[hlen - 1 - j], r);
                r = pdwt_fma(u2[(ty0 + i + j) * TX + k], hi[hlen - 1 - j], r);
            }
            const int si = it * TY + ty0 + i;
            if (ph + f * si < a.Nr && x < a.Nc) a.out[boff + (long long)(ph + f * si) * a.Nc + x] = 0.5f * r;
        }
    }
}

// ---------------------------------------------------------------------------
// Vectorised twins of the two fused kernels: every thread owns FOUR consecutive columns, so the
// dilated taps are read with one 16-B load per band and tap (4-B aligned is enough for a global
// dwordx4 load: the shift (j-c) f need not be a multiple of 4), the LDS planes are read and written
// 16 B at a time and the results leave with 16-B stores.  rocprofv3 on cfg4 showed the scalar
// kernels at 55 % VALU-busy with 4x the memory instructions of a streaming kernel.
// Requirements (host): 16-B aligned planes, compile-time filter length.  Rows of ANY length (round 5; the reference's kernels
// take any width, pdwt/src/separable.cu:409-493,553-626, and odd sizes are an option of its tests, test/test_wavelets.py:43):
// where rows are not whole aligned quads the taps are 16-B loads at 4-B alignment, the outputs 16-B stores at 4-B alignment and
// the last, partial quad of a row is stored element by element -- 2047^2 used to fall to the one-sample-per-thread kernels at
// twice the time of 2048^2.
// ---------------------------------------------------------------------------
#ifdef PDWT_CPU_EMU
struct rv4 { real_t x, y, z, w; };
static inline rv4 load4u(const real_t* p) { return rv4{p[0], p[1], p[2], p[3]}; }
static inline rv4 load4(const real_t* p) { return rv4{p[0], p[1], p[2], p[3]}; }
static inline void store4(real_t* p, const rv4& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
#else
typedef real_t rv4 __attribute__((ext_vector_type(4)));
#ifdef PDWT_DOUBLE
typedef real_t rv4u __attribute__((ext_vector_type(4), aligned(8)));
#else
typedef real_t rv4u __attribute__((ext_vector_type(4), aligned(4)));
#endif
static __device__ __forceinline__ rv4 load4u(const real_t* p) { return *reinterpret_cast<const rv4u*>(p); }
static __device__ __forceinline__ rv4 load4(const real_t* p) { return *reinterpret_cast<const rv4*>(p); }
static __device__ __forceinline__ void store4(real_t* p, const rv4& v) { *reinterpret_cast<rv4*>(p) = v; }
#endif

// the quad of columns x0 .. x0 + 3 of a row (p = row + x0) of Nc columns: aligned 16-B store where rows are whole aligned quads
// (rows16), else 16 B at 4-B alignment, the partial quad at the row end element by element
PDWT_DEVICE void store4_row(real_t* p, const rv4& v, int x0, int Nc, bool rows16) {
    if (rows16) {
        store4(p, v);
    } else if (x0 + 3 < Nc) {
#ifdef PDWT_CPU_EMU
        store4(p, v);
#else
        *reinterpret_cast<rv4u*>(p) = v;
#endif
    } else {
        p[0] = v.x;
        if (x0 + 1 < Nc) p[1] = v.y;
        if (x0 + 2 < Nc) p[2] = v.z;
    }
}

// four consecutive samples of a periodic row starting at (possibly negative / overflowing) column p
PDWT_DEVICE rv4 load4_periodic(const real_t* PDWT_RESTRICT row, int p, int Nc) {
    const int q = wrap_periodic(p, Nc);
    if (q + 3 < Nc) return load4u(row + q);
    rv4 v;  // the group straddles the period: only in the first / last tile of a row
    v.x = row[q];
    v.y = row[wrap_periodic(q + 1, Nc)];
    v.z = row[wrap_periodic(q + 2, Nc)];
    v.w = row[wrap_periodic(q + 3, Nc)];
    return v;
}

PDWT_DEVICE void fma4(rv4& acc, const rv4& v, real_t t) {
    acc.x = pdwt_fma(v.x, t, acc.x);
    acc.y = pdwt_fma(v.y, t, acc.y);
    acc.z = pdwt_fma(v.z, t, acc.z);
    acc.w = pdwt_fma(v.w, t, acc.w);
}

PDWT_DEVICE rv4 soft4(const rv4& v, real_t b) {
    rv4 r;
    r.x = soft_shrink(v.x, b); r.y = soft_shrink(v.y, b); r.z = soft_shrink(v.z, b); r.w = soft_shrink(v.w, b);
    return r;
}

// Tap loops are fully unrolled up to 24 taps (taps in SGPRs, all of a row's loads in flight); beyond, hipcc hoists every
// load of the unrolled body -- 4 bands x 40 taps x 4 registers in the inverse -- caps at 256 VGPRs and spills to scratch
// (620 B per lane at 40 taps): those filters unroll 8 taps at a time.
template <int HLEN>
constexpr int kSwtTapUnroll = sizeof(real_t) == 8 ? (HLEN > 10 ? 4 : (HLEN > 0 ? HLEN : 1))  // fp64: see below
                                                  : (HLEN > 24 ? 8 : (HLEN > 0 ? HLEN : 1));
// (fp64, round 4: a value is two registers -- the fully unrolled inverse of 12-24 taps spilled 60-2120 B per lane to scratch
// and took 700 us per 2048^2 level of 16 taps, nine times the fp32 time; four taps at a time fit)

// The tiled INVERSE keeps four bands in flight per tap: 12-24 taps fully unrolled need 282-476 VGPRs (one wavefront per SIMD).
// On the 128-column tiles eight taps at a time are faster (768^2, two levels forward+inverse: db6 67.2 -> 53.9 us, sym8 79.5 ->
// 64.1); on the 64 x 8 tiles of small launches the full unroll stays ahead by 3 % (profiles/r04zo_swt_inv_unroll.txt).
// PDWT_SWT_INV_UNROLL: A/B builds (PDWT_EXTRA_DEFINES of pypwt_amd/build.py, lab variant).
#ifndef PDWT_SWT_INV_UNROLL
#define PDWT_SWT_INV_UNROLL 8
#endif
template <int HLEN, int TX>
constexpr int kSwtInvTapUnroll = (PDWT_SWT_INV_UNROLL > 0 && TX >= 128 && HLEN >= 12 && HLEN <= 24 && sizeof(real_t) == 4)
                                     ? PDWT_SWT_INV_UNROLL : kSwtTapUnroll<HLEN>;

template <int TX, int TY>
constexpr int swt2d_vec_lds_floats(int hlen) { return 2 * (TY + hlen - 1) * TX; }

// The inverse row pass reads hlen taps of FOUR bands per output quad.  Each row group stages its row of the four bands
// (tile + (hlen - 1) f halo columns, at most kSwtStageHalo, from a 4-aligned origin) in LDS once -- 4 (1 + halo / TX)
// global loads per quad instead of 4 hlen -- and takes the taps from there: aligned 16-B LDS reads where the dilation is a
// multiple of 4 (levels >= 3), four 4-B reads at dilation 1 and 2.
// (doubles: 128 columns of halo -- with 256 the staged rows of a 128 x 16 tile are 163-171 KB from 16 taps on and the fp64 inverse of
// 10+ taps fell back to one 32-B load per band and tap: 4.5-6x the fp32 time, round 5)
constexpr int kSwtStageHalo = sizeof(real_t) == 8 ? 128 : 256;
// (measured per 2048^2 level: 16 taps 193 -> 95 us at dilation 4+, 200 -> 139 us at dilation 1, 2; 8 taps 45 -> 32 and 53 -> 49 us;
// 4 taps 25 -> 25 and 28 -> 35 us: short filters keep the direct loads)
constexpr bool swt_inv_staged_filter(int hlen, int f) {
    return (hlen - 1) * f <= kSwtStageHalo && ((f & 3) == 0 ? hlen >= 6 : hlen >= 10);
}
template <int TX, int TY, int NT>
constexpr int swt2d_inv_vec_lds_floats(int hlen, bool staged) {
    return swt2d_vec_lds_floats<TX, TY>(hlen) + (staged ? (NT / (TX / 4)) * 4 * (TX + kSwtStageHalo + 4) : 0);
}
// ... and only while the staged rows fit the LDS beside the (u1, u2) tile (the fp64 build's long filters do not)
template <int TX, int TY, int NT>
constexpr bool swt_inv_staged(int hlen, int f) {  // host (LDS request) and device (path), same answer
    return swt_inv_staged_filter(hlen, f) &&
           (long long)swt2d_inv_vec_lds_floats<TX, TY, NT>(hlen, true) * (long long)sizeof(real_t) <= 150 * 1024;
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void swt2_fwd_vec_tile(const Swt2DArgs& a, int bx, int by, int bz, real_t* smem) {
    constexpr int QX = TX / 4, NG = NT / QX, R = TY / NG;
    static_assert(HLEN > 0 && TX % 4 == 0 && NT % QX == 0 && TY % NG == 0, "tile/thread shape");
    constexpr int c = (HLEN & 1) ? HLEN / 2 : HLEN / 2 - 1;  // analysis_centre
    constexpr int RY = TY + HLEN - 1;
    const int f = a.f, ph = by % f, it = by / f;
    real_t* tL = smem;
    real_t* tH = tL + RY * TX;
    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.bstride;
    const real_t zero = 0;
    const bool rows16 = !(a.Nc & 3) && !(a.bstride & 3);  // every row starts 16-B aligned and ends with a whole quad

    // tiles whose dilated taps stay inside the row (all but the first / last of a row) skip the periodic-wrap
    // arithmetic of every load: a tile-uniform branch (it was a third of the kernel's vector instructions)
    const bool interior = bx * TX - c * f >= 0 && bx * TX + TX + (HLEN - 1 - c) * f <= a.Nc;
    PDWT_FOR_THREADS(tid, NT) {
        const int k4 = tid % QX, x0 = bx * TX + 4 * k4;
        // Long filters: constant trip count, row clamped (surplus trips redo the last row with the same result) -- no exit
        // test between the trips, so the next row's HLEN loads are issued while this row is summed: 16 taps at 2048^2
        // 49 -> 41 us per level; 2-8 taps unchanged, and unrolling them only grows the binary (tools/ktimes.py).
        constexpr int TRIPS = (RY + NG - 1) / NG;
#pragma unroll((HLEN >= 10 && HLEN <= 24) ? TRIPS : 1)
        for (int t = 0; t < TRIPS; ++t) {
            int r = tid / QX + t * NG;
            if (HLEN < 10 && r >= RY) break;
            r = r < RY ? r : RY - 1;
            rv4 aL = {zero, zero, zero, zero}, aH = aL;
            if (x0 < a.Nc) {
                const real_t* row = in + (long long)wrap_periodic(ph + f * (it * TY - c + r), a.Nr) * a.Nc;
#pragma unroll(kSwtTapUnroll<HLEN>)
                for (int j = 0; j < HLEN; ++j) {
                    const rv4 v = interior ? load4u(row + x0 + (j - c) * f) : load4_periodic(row, x0 + (j - c) * f, a.Nc);
                    fma4(aL, v, a.fb.lo[HLEN - 1 - j]);
                    fma4(aH, v, a.fb.hi[HLEN - 1 - j]);
                }
            }
            store4(tL + r * TX + 4 * k4, aL);
            store4(tH + r * TX + 4 * k4, aH);
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const int k4 = tid % QX, ty0 = (tid / QX) * R, x0 = bx * TX + 4 * k4;
        const long long boff = (long long)bz * a.bstride;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            rv4 rA = {zero, zero, zero, zero}, rH = rA, rV = rA, rD = rA;
#pragma unroll(kSwtTapUnroll<HLEN>)
            for (int j = 0; j < HLEN; ++j) {
                const rv4 l = load4(tL + (ty0 + i + j) * TX + 4 * k4);
                const rv4 h = load4(tH + (ty0 + i + j) * TX + 4 * k4);
                const real_t tl = a.fb.lo[HLEN - 1 - j], th = a.fb.hi[HLEN - 1 - j];
                fma4(rA, l, tl);
                fma4(rH, l, th);
                fma4(rV, h, tl);
                fma4(rD, h, th);
            }
            const int si = it * TY + ty0 + i;
            if (ph + f * si < a.Nr && x0 < a.Nc) {
                const long long o = boff + (long long)(ph + f * si) * a.Nc + x0;
                store4_row(a.A + o, rA, x0, a.Nc, rows16);
                store4_row(a.H + o, rH, x0, a.Nc, rows16);
                store4_row(a.V + o, rV, x0, a.Nc, rows16);
                store4_row(a.D + o, rD, x0, a.Nc, rows16);
            }
        }
    }
}

// taps of one staged band row at dilation F = 1 or 2 for the four outputs of a lane: ONE window of aligned quads
// (PADP + 4 + (HLEN - 1) F floats from the lane's quad on) instead of four 4-B reads per tap
template <int HLEN, int F>
PDWT_DEVICE void swt_inv_taps_window(const real_t* b, const real_t* taps, rv4& acc) {
    constexpr int c = HLEN / 2, PADP = (4 - ((c * F) & 3)) & 3;
    constexpr int NW = (PADP + 4 + (HLEN - 1) * F + 3) / 4;
    real_t v[4 * NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const rv4 w = load4(b + 4 * q);
        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int j = 0; j < HLEN; ++j) {
        const real_t t = taps[HLEN - 1 - j];
        acc.x = pdwt_fma(v[PADP + j * F], t, acc.x);
        acc.y = pdwt_fma(v[PADP + j * F + 1], t, acc.y);
        acc.z = pdwt_fma(v[PADP + j * F + 2], t, acc.z);
        acc.w = pdwt_fma(v[PADP + j * F + 3], t, acc.w);
    }
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void swt2_inv_vec_tile(const Swt2DArgs& a, int bx, int by, int bz, real_t* smem) {
    constexpr int QX = TX / 4, NG = NT / QX, R = TY / NG;
    static_assert(HLEN > 0 && TX % 4 == 0 && NT % QX == 0 && TY % NG == 0, "tile/thread shape");
    constexpr int c = HLEN / 2;  // synthesis centre
    constexpr int RY = TY + HLEN - 1;
    const int f = a.f, ph = by % f, it = by / f;
    real_t* u1 = smem;
    real_t* u2 = u1 + RY * TX;
    const long long boff = (long long)bz * a.bstride;
    const real_t zero = 0, half = (real_t)0.5;
    const bool rows16 = !(a.Nc & 3) && !(a.bstride & 3);  // every row starts 16-B aligned and ends with a whole quad

    // dilated row synthesis: u1 = Lx(A) + Hx(V), u2 = Lx(H) + Hx(D) (pending soft threshold applied to the detail
    // bands as they are loaded, never to A)
    const bool interior = bx * TX - c * f >= 0 && bx * TX + TX + (HLEN - 1 - c) * f <= a.Nc;  // see the forward tile
    if (swt_inv_staged<TX, TY, NT>(HLEN, f)) {
        // staged: per trip every row group (QX threads) loads its row of the four bands once -- tile + halo columns,
        // periodic (whole aligned quads where the rows are; else unaligned 16-B loads) -- then takes its taps from LDS
        constexpr int SW = TX + kSwtStageHalo + 4;        // floats per staged band row
        constexpr int TRIPS = (RY + NG - 1) / NG;
        const int padp = (4 - ((c * f) & 3)) & 3;          // the staged row starts at the 4-aligned column below bx TX - c f
        const int xl = c * f + padp, W4 = (TX + (HLEN - 1) * f + padp + 3) / 4;  // columns left of the tile; staged quads per band
        const bool aligned_taps = (f & 3) == 0;
        real_t* stage = smem + 2 * RY * TX;
        for (int t = 0; t < TRIPS; ++t) {
            PDWT_FOR_THREADS(tid, NT) {
                const int k4 = tid % QX, g = tid / QX, r = g + t * NG;
                if (r < RY) {
                    const long long ro = boff + (long long)wrap_periodic(ph + f * (it * TY - c + r), a.Nr) * a.Nc;
                    real_t* sg = stage + g * 4 * SW;
                    if (rows16) {
                        for (int q = k4; q < W4; q += QX) {
                            const int p = wrap_periodic(bx * TX - xl + 4 * q, a.Nc);  // a multiple of 4: the quad never straddles
                            store4(sg + 4 * q, load4(a.A + ro + p));
                            store4(sg + SW + 4 * q, soft4(load4(a.V + ro + p), a.soft_beta));
                            store4(sg + 2 * SW + 4 * q, soft4(load4(a.H + ro + p), a.soft_beta));
                            store4(sg + 3 * SW + 4 * q, soft4(load4(a.D + ro + p), a.soft_beta));
                        }
                    } else {  // rows of any length: 16-B loads at 4-B alignment, the quads at the row ends element by element
                        for (int q = k4; q < W4; q += QX) {
                            const int p = bx * TX - xl + 4 * q;
                            store4(sg + 4 * q, load4_periodic(a.A + ro, p, a.Nc));
                            store4(sg + SW + 4 * q, soft4(load4_periodic(a.V + ro, p, a.Nc), a.soft_beta));
                            store4(sg + 2 * SW + 4 * q, soft4(load4_periodic(a.H + ro, p, a.Nc), a.soft_beta));
                            store4(sg + 3 * SW + 4 * q, soft4(load4_periodic(a.D + ro, p, a.Nc), a.soft_beta));
                        }
                    }
                }
            }
            PDWT_SYNC();
            PDWT_FOR_THREADS(tid, NT) {
                const int k4 = tid % QX, g = tid / QX, r = g + t * NG;
                if (r < RY) {
                    const real_t* sg = stage + g * 4 * SW + 4 * k4 + padp;  // tap j of this quad sits j f columns further on
                    rv4 r1 = {zero, zero, zero, zero}, r2 = r1;
                    if (bx * TX + 4 * k4 < a.Nc) {
                        if (aligned_taps) {
#pragma unroll(kSwtInvTapUnroll<HLEN, TX>)
                            for (int j = 0; j < HLEN; ++j) {
                                const real_t tl = a.fb.lo[HLEN - 1 - j], th = a.fb.hi[HLEN - 1 - j];
                                fma4(r1, load4(sg + j * f), tl);
                                fma4(r1, load4(sg + SW + j * f), th);
                                fma4(r2, load4(sg + 2 * SW + j * f), tl);
                                fma4(r2, load4(sg + 3 * SW + j * f), th);
                            }
                        } else {
                            const real_t* sq = sg - padp;  // the lane's own (aligned) quad of the staged row
                            if (f == 1) {
                                swt_inv_taps_window<HLEN, 1>(sq, a.fb.lo, r1);
                                swt_inv_taps_window<HLEN, 1>(sq + SW, a.fb.hi, r1);
                                swt_inv_taps_window<HLEN, 1>(sq + 2 * SW, a.fb.lo, r2);
                                swt_inv_taps_window<HLEN, 1>(sq + 3 * SW, a.fb.hi, r2);
                            } else {  // f == 2 (3 does not occur: dilations are powers of two)
                                swt_inv_taps_window<HLEN, 2>(sq, a.fb.lo, r1);
                                swt_inv_taps_window<HLEN, 2>(sq + SW, a.fb.hi, r1);
                                swt_inv_taps_window<HLEN, 2>(sq + 2 * SW, a.fb.lo, r2);
                                swt_inv_taps_window<HLEN, 2>(sq + 3 * SW, a.fb.hi, r2);
                            }
                        }
                    }
                    r1.x *= half; r1.y *= half; r1.z *= half; r1.w *= half;
                    r2.x *= half; r2.y *= half; r2.z *= half; r2.w *= half;
                    store4(u1 + r * TX + 4 * k4, r1);
                    store4(u2 + r * TX + 4 * k4, r2);
                }
            }
            PDWT_SYNC();
        }
    } else
    PDWT_FOR_THREADS(tid, NT) {
        const int k4 = tid % QX, x0 = bx * TX + 4 * k4;
        for (int r = tid / QX; r < RY; r += NG) {
            rv4 r1 = {zero, zero, zero, zero}, r2 = r1;
            if (x0 < a.Nc) {
                const long long ro = boff + (long long)wrap_periodic(ph + f * (it * TY - c + r), a.Nr) * a.Nc;
                if (interior) {
#pragma unroll(kSwtInvTapUnroll<HLEN, TX>)
                    for (int j = 0; j < HLEN; ++j) {
                        const long long o = ro + x0 + (j - c) * f;
                        const real_t tl = a.fb.lo[HLEN - 1 - j], th = a.fb.hi[HLEN - 1 - j];
                        fma4(r1, load4u(a.A + o), tl);
                        fma4(r1, soft4(load4u(a.V + o), a.soft_beta), th);
                        fma4(r2, soft4(load4u(a.H + o), a.soft_beta), tl);
                        fma4(r2, soft4(load4u(a.D + o), a.soft_beta), th);
                    }
                } else {
#pragma unroll(kSwtInvTapUnroll<HLEN, TX>)
                    for (int j = 0; j < HLEN; ++j) {
                        const int p = x0 + (j - c) * f;
                        const real_t tl = a.fb.lo[HLEN - 1 - j], th = a.fb.hi[HLEN - 1 - j];
                        fma4(r1, load4_periodic(a.A + ro, p, a.Nc), tl);
                        fma4(r1, soft4(load4_periodic(a.V + ro, p, a.Nc), a.soft_beta), th);
                        fma4(r2, soft4(load4_periodic(a.H + ro, p, a.Nc), a.soft_beta), tl);
                        fma4(r2, soft4(load4_periodic(a.D + ro, p, a.Nc), a.soft_beta), th);
                    }
                }
            }
            r1.x *= half; r1.y *= half; r1.z *= half; r1.w *= half;
            r2.x *= half; r2.y *= half; r2.z *= half; r2.w *= half;
            store4(u1 + r * TX + 4 * k4, r1);
            store4(u2 + r * TX + 4 * k4, r2);
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const int k4 = tid % QX, ty0 = (tid / QX) * R, x0 = bx * TX + 4 * k4;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            rv4 r = {zero, zero, zero, zero};
#pragma unroll(kSwtInvTapUnroll<HLEN, TX>)
            for (int j = 0; j < HLEN; ++j) {
                fma4(r, load4(u1 + (ty0 + i + j) * TX + 4 * k4), a.fb.lo[HLEN - 1 - j]);
                fma4(r, load4(u2 + (ty0 + i + j) * TX + 4 * k4), a.fb.hi[HLEN - 1 - j]);
            }
            const int si = it * TY + ty0 + i;
            if (ph + f * si < a.Nr && x0 < a.Nc) {
                r.x *= half; r.y *= half; r.z *= half; r.w *= half;
                store4_row(a.out + boff + (long long)(ph + f * si) * a.Nc + x0, r, x0, a.Nc, rows16);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Direct one-pass kernels: one output sample per thread, taps read from global.
// `along_y` selects the filtered axis.  Used for the (batched) 1D SWT and as the
// 2D fallback when f does not divide Nr.
// ---------------------------------------------------------------------------

template <int NT>
PDWT_DEVICE void swt_pass_fwd_tile(const SwtPassArgs& a, long long block, real_t* /*smem*/) {
    PDWT_FOR_THREADS(tid, NT) {
        const long long idx = block * NT + tid;
        const long long plane = (long long)a.Nr * a.Nc;
        if (idx < plane * a.images) {
            const long long base = (idx / plane) * plane;  // the image's plane (a.images planes back to back behind every pointer)
            const long long rem = idx - base;
            const int y = (int)(rem / a.Nc);
            const int x = (int)(rem - (long long)y * a.Nc);
            const int c = analysis_centre(a.hlen);
            real_t aL = 0.f, aH = 0.f;
            for (int j = 0; j < a.hlen; ++j) {
                long long src;
                if (a.along_y) src = base + (long long)wrap_periodic(y + (j - c) * a.f, a.Nr) * a.Nc + x;
                else src = base + (long long)y * a.Nc + wrap_periodic(x + (j - c) * a.f, a.Nc);
                const real_t v = a.in0[src];
                aL = pdwt_fma(v, a.fb.lo[a.hlen - 1 - j], aL);
                aH = pdwt_fma(v, a.fb.hi[a.hlen - 1 - j], aH);
            }
            a.out0[idx] = aL;
            a.out1[idx] = aH;
        }
    }
}

template <int NT>
PDWT_DEVICE void swt_pass_inv_tile(const SwtPassArgs& a, long long block, real_t* /*smem*/) {
    PDWT_FOR_THREADS(tid, NT) {
        const long long idx = block * NT + tid;
        const long long plane = (long long)a.Nr * a.Nc;
        if (idx < plane * a.images) {
            const long long base = (idx / plane) * plane;
            const long long rem = idx - base;
            const int y = (int)(rem / a.Nc);
            const int x = (int)(rem - (long long)y * a.Nc);
            const int c = a.hlen / 2;
            real_t r = 0.f;
            for (int j = 0; j < a.hlen; ++j) {
                long long src;
                if (a.along_y) src = base + (long long)wrap_periodic(y + (j - c) * a.f, a.Nr) * a.Nc + x;
                else src = base + (long long)y * a.Nc + wrap_periodic(x + (j - c) * a.f, a.Nc);
                r = pdwt_fma(a.in0[src], a.fb.lo[a.hlen - 1 - j], r);
                r = pdwt_fma(a.in1[src], a.fb.hi[a.hlen - 1 - j], r);
            }
            a.out0[idx] = 0.5f * r;
        }
    }
}

// ---------------------------------------------------------------------------
// Vectorised one-pass kernels for the (batched) 1D SWT (rows filtered along x; reference drivers
// pdwt/src/separable.cu:496-537,629-672 `_1d`): a work item owns FOUR consecutive samples of a row.  Tap j of the
// a-trous filter is the same four-sample window shifted by (j - c) f: one 16-B load per tap (4-B alignment suffices
// on gfx950, so every dilation takes the same path), all HLEN loads of a work item issued before the first is used
// (compile-time filter length, taps in SGPRs).  Neighbouring taps and neighbouring work items overlap and are served
// by L1 / L2: HBM sees each sample once (forward: 4 B read + 8 B written per sample; inverse 8 + 4).  Work items
// whose window crosses a row end (the periodic wrap) gather their taps element by element.
// Against the one-output-per-thread kernel above: a quarter of the memory instructions, none of the per-tap index
// arithmetic.  Needs Nc % 4 == 0 and an even filter length.
// ---------------------------------------------------------------------------
template <int HLEN, int NT>
PDWT_DEVICE void swt1_fwd_vec_tile(const SwtPassArgs& a, long long block) {
    PDWT_FOR_THREADS(tid, NT) {
        const int q4 = a.Nc >> 2;
        const long long idx = block * NT + tid;
        if (idx < (long long)a.Nr * q4) {
            const int y = (int)(idx / q4);
            const int x = 4 * (int)(idx - (long long)y * q4);
            constexpr int c = HLEN / 2 - 1;
            const real_t* PDWT_RESTRICT row = a.in0 + (long long)y * a.Nc;
            const int lo = x - c * a.f, hi = x + 3 + (HLEN - 1 - c) * a.f;
            rv4 v[HLEN];
            if (lo >= 0 && hi < a.Nc) {
#pragma unroll
                for (int j = 0; j < HLEN; ++j) v[j] = load4u(row + lo + j * a.f);
            } else {
#pragma unroll
                for (int j = 0; j < HLEN; ++j) {
                    const int s = lo + j * a.f;
                    v[j].x = row[wrap_periodic(s, a.Nc)];
                    v[j].y = row[wrap_periodic(s + 1, a.Nc)];
                    v[j].z = row[wrap_periodic(s + 2, a.Nc)];
                    v[j].w = row[wrap_periodic(s + 3, a.Nc)];
                }
            }
            rv4 L, H;
            L.x = L.y = L.z = L.w = real_t(0);
            H = L;
#pragma unroll
            for (int j = 0; j < HLEN; ++j) {
                fma4(L, v[j], a.fb.lo[HLEN - 1 - j]);
                fma4(H, v[j], a.fb.hi[HLEN - 1 - j]);
            }
            store4(a.out0 + (long long)y * a.Nc + x, L);
            store4(a.out1 + (long long)y * a.Nc + x, H);
        }
    }
}

template <int HLEN, int NT>
PDWT_DEVICE void swt1_inv_vec_tile(const SwtPassArgs& a, long long block) {
    PDWT_FOR_THREADS(tid, NT) {
        const int q4 = a.Nc >> 2;
        const long long idx = block * NT + tid;
        if (idx < (long long)a.Nr * q4) {
            const int y = (int)(idx / q4);
            const int x = 4 * (int)(idx - (long long)y * q4);
            constexpr int c = HLEN / 2;
            const real_t* PDWT_RESTRICT rA = a.in0 + (long long)y * a.Nc;
            const real_t* PDWT_RESTRICT rD = a.in1 + (long long)y * a.Nc;
            const int lo = x - c * a.f, hi = x + 3 + (HLEN - 1 - c) * a.f;
            rv4 va[HLEN], vd[HLEN];
            if (lo >= 0 && hi < a.Nc) {
#pragma unroll
                for (int j = 0; j < HLEN; ++j) {
                    va[j] = load4u(rA + lo + j * a.f);
                    vd[j] = load4u(rD + lo + j * a.f);
                }
            } else {
#pragma unroll
                for (int j = 0; j < HLEN; ++j) {
                    const int s = lo + j * a.f;
                    const int s0 = wrap_periodic(s, a.Nc), s1 = wrap_periodic(s + 1, a.Nc), s2 = wrap_periodic(s + 2, a.Nc),
                              s3 = wrap_periodic(s + 3, a.Nc);
                    va[j].x = rA[s0]; va[j].y = rA[s1]; va[j].z = rA[s2]; va[j].w = rA[s3];
                    vd[j].x = rD[s0]; vd[j].y = rD[s1]; vd[j].z = rD[s2]; vd[j].w = rD[s3];
                }
            }
            rv4 r;
            r.x = r.y = r.z = r.w = real_t(0);
#pragma unroll
            for (int j = 0; j < HLEN; ++j) {
                fma4(r, va[j], a.fb.lo[HLEN - 1 - j]);
                fma4(r, vd[j], a.fb.hi[HLEN - 1 - j]);
            }
            const real_t half = real_t(0.5);
            r.x *= half; r.y *= half; r.z *= half; r.w *= half;
            store4(a.out0 + (long long)y * a.Nc + x, r);
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) swt1_fwd_vec_kernel(const SwtPassArgs a) {
    swt1_fwd_vec_tile<HLEN, NT>(a, blockIdx.x);
}
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) swt1_inv_vec_kernel(const SwtPassArgs a) {
    swt1_inv_vec_tile<HLEN, NT>(a, blockIdx.x);
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) swt2_fwd_kernel(const Swt2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    swt2_fwd_tile<HLEN, TX, TY, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z, pdwt_smem);
}
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) swt2_inv_kernel(const Swt2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    swt2_inv_tile<HLEN, TX, TY, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z, pdwt_smem);
}
// Vector kernels: 1D grid of 8*chunk workgroups per image.  Workgroup ids b and b+8 share an XCD (and its
// L2); XCD x gets the contiguous range [x*chunk, (x+1)*chunk) of tiles ordered (phase, row tile, column
// tile), so the tiles that re-read each other's cache lines -- column neighbours (taps shifted by
// (j-c) f) and row-tile neighbours of one phase (the hlen-1 halo rows) -- meet in one L2.  rocprofv3: the
// haar inverse fetched 1.4x its algorithmic bytes with the plain 3D grid.
template <int TX, int TY>
__device__ __forceinline__ bool swt_vec_tile(const Swt2DArgs& a, int block, int& bx, int& by) {
    const int tiles_x = (a.Nc + TX - 1) / TX;
    const int nrt = ((a.Nr + a.f - 1) / a.f + TY - 1) / TY;  // row tiles per phase (the longest one)
    const int total = tiles_x * nrt * a.f;
    const int chunk = (total + 7) >> 3;
    const int tile = (block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || tile >= total) return false;
    const int row = tile / tiles_x;  // = ph * nrt + it
    bx = tile - row * tiles_x;
    const int ph = row / nrt, it = row - ph * nrt;
    by = it * a.f + ph;              // the tile functions decode by as (it, ph) = (by / f, by % f)
    return true;
}
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) swt2_fwd_vec_kernel(const Swt2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int bx, by;
    if (!swt_vec_tile<TX, TY>(a, blockIdx.x, bx, by)) return;
    swt2_fwd_vec_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) swt2_inv_vec_kernel(const Swt2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int bx, by;
    if (!swt_vec_tile<TX, TY>(a, blockIdx.x, bx, by)) return;
    swt2_inv_vec_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
template <int NT>
__global__ void __launch_bounds__(NT) swt_pass_fwd_kernel(const SwtPassArgs a) {
    swt_pass_fwd_tile<NT>(a, blockIdx.x, nullptr);
}
template <int NT>
__global__ void __launch_bounds__(NT) swt_pass_inv_kernel(const SwtPassArgs a) {
    swt_pass_inv_tile<NT>(a, blockIdx.x, nullptr);
}
#endif

}  // namespace pdwt
