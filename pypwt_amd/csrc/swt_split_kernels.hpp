// swt_split_kernels.hpp -- one undecimated (a-trous) 2D level as TWO register-blocked launches, for the filter lengths
// at which the LDS-tiled level kernels of swt_kernels.hpp run out of operand bandwidth (gfx950).
//
// Why.  In the tiled kernels every multiply-add pair of the row pass has its own 16-B load (a tap is a shifted row
// segment: nothing is reused between taps once the dilation is >= 4) and the tile's (hlen - 1) halo rows are filtered
// again by every tile.  rocprofv3 on 2048^2: the 16-tap inverse issues 257 loads per output quad and sits at 35 % VALU
// utilisation (90 us per level), the 40-tap kernels keep one or two wavefronts per SIMD (72 KB of LDS per workgroup) and
// take 144 / 160 us where their arithmetic needs ~15 us.
//
// How.  Outputs that are ONE DILATION STEP apart share all but one of their inputs.  A work item therefore owns R
// outputs spaced f apart along the filtered axis (four adjacent columns each) and streams the R + hlen - 1 input quads
// past R x 4 stationary accumulator pairs: (R + hlen - 1) loads for R hlen packed multiply-adds per column instead of
// R hlen loads.  That blocking needs a different decomposition for each axis, so the level runs as a row launch and a
// column launch with the intermediate planes in scratch (the reference's own structure, pdwt/src/separable.cu:496-537
// and :629-672, there with one output per thread and one load per tap).  No LDS, no barriers, any row count (no phase
// tiles: the periodic index is advanced by f and wrapped per load).
//   forward   row kernel : in -> lo, hi planes (scratch)              column kernel : lo, hi -> A, H, V, D
//   inverse   column kernel : A, H, V, D -> (L', H') interleaved      row kernel : (L', H') -> out
// Every multiply-add is a v_pk_fma_f32 on a register pair: (lo, hi) outputs of one input in the forward kernels,
// two adjacent columns in the inverse column kernel, the interleaved (L', H') pair in the inverse row kernel.
// Dilation 1 and 2 (levels 1, 2) in the row kernels: a work item owns 16 CONSECUTIVE columns and streams its window of
// 16 + (hlen - 1) f samples; dilations >= 4 are multiples of 4, so whole quads are shared.
//
// Semantics: swt_kernels.hpp (analysis centre hlen/2 - 1, synthesis centre hlen/2, taps reversed, 1/2 per synthesis
// pass); the inverse runs its column pass first like the reference (the tiled kernels run the row pass first: fp32
// rounding differs, nothing else).  fp32 only.
#pragma once

#include "packed_math.hpp"
#include "swt_kernels.hpp"

namespace pdwt {

struct SwtSplitArgs {
    const real_t* in[4];  // row fwd: in ; col fwd: lo, hi ; col inv: A, H, V, D ; row inv: interleaved (L', H')
    real_t* out[4];       // row fwd: lo, hi ; col fwd: A, H, V, D ; col inv: interleaved ; row inv: out
    int Nr, Nc, f, batch;
    long long in_bstride, out_bstride;  // elements between the images of a batch, per plane as laid out
    real_t soft_beta;                   // col inv: soft threshold applied to H, V, D as they are loaded (0: none)
    FilterBankI t;                      // t[j] = (lo[hlen-1-j], hi[hlen-1-j])
};

#ifdef PDWT_CPU_EMU
#define PDWT_SCHED_FENCE() ((void)0)
#else
// keeps the machine scheduler from hoisting every load of the unrolled body to the top (40 taps: > 256 VGPRs, scratch)
#define PDWT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// The next chunk's addresses are made to depend (through an empty asm) on a value the previous chunk's arithmetic
// produced, so at most two chunks of loads are ever in flight.
#ifdef PDWT_CPU_EMU
#define PDWT_ORDER_AFTER(ivar, r0, r1, r2, r3) ((void)0)
#define PDWT_ORDER_AFTER_S(ivar, r0, r1, r2, r3) ((void)0)
#else
#define PDWT_ORDER_AFTER(ivar, r0, r1, r2, r3) asm volatile("" : "+v"(ivar), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3))
#define PDWT_ORDER_AFTER_S(ivar, r0, r1, r2, r3) asm volatile("" : "+s"(ivar), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3))
#endif

PDWT_DEVICE rv4 mk4(real_t a, real_t b, real_t c, real_t d) {
    rv4 v;
    v.x = a; v.y = b; v.z = c; v.w = d;
    return v;
}
// wave-uniform values are forced into SGPRs: the row walk of the column kernels (index, wrap, 64-bit row address) then
// costs no vector instruction at all and the loads take the scalar-base form
#ifdef PDWT_CPU_EMU
#define PDWT_UNIFORM(x) (x)
#else
#define PDWT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
PDWT_DEVICE int step_wrap(int p, int step, int n) {  // p in [0, n), 0 < step < n
    p += step;
    return p >= n ? p - n : p;
}
// per-lane variant in three vector instructions: p + step and p + step - n as unsigned, the smaller one is the wrapped index
PDWT_DEVICE unsigned step_wrap_u(unsigned p, unsigned step, unsigned n) {
    const unsigned q = p + step, r = q - n;
    return q < r ? q : r;
}

// rows per chunk of loads in flight (a chunk is issued while the previous one is consumed)
constexpr int kSplitChunk = 4;
#ifndef PDWT_SPLIT_CH_COLF
#define PDWT_SPLIT_CH_COLF 2
#define PDWT_SPLIT_CH_COLI 2
#define PDWT_SPLIT_NTC 1024
#endif

// Work decomposition of the column kernels: a WAVEFRONT owns 64 adjacent quad columns of (image, phase, block of R phase
// rows); rows of a phase are f apart.  Returns false for wavefronts beyond the work.
struct SplitColWork {
    int q, ph, blk;
    long long bz;
    bool active;  // lanes beyond the last quad of a row load (clamped) and do not store
};
template <int R, int NT>
PDWT_DEVICE bool split_col_work(const SwtSplitArgs& a, long long block, int tid, SplitColWork& w) {
    // the blocks of R rows are the FASTEST index: the NT / 64 wavefronts of a workgroup walk consecutive blocks of one
    // column group, so of the R + hlen - 1 rows a wavefront reads all but R were just read by its neighbour on the same CU
    // (L1 / the XCD's L2); with one block per wavefront anywhere, every row came hlen / R + 1 times from the Infinity Cache
    // (2048^2, 40 taps: 360 MB per launch at the 6.5 TB/s that cache delivers = the whole 55 us of the kernel)
    const int QW = a.Nc >> 2, QG = (QW + 63) >> 6, f = a.f;
    const int blocks = ((a.Nr + f - 1) / f + R - 1) / R;
    const long long wave = block * (NT / 64) + PDWT_UNIFORM(tid >> 6);
    if (wave >= (long long)a.batch * f * blocks * QG) return false;
    w.blk = (int)(wave % blocks);
    long long t = wave / blocks;
    const int qg = (int)(t % QG);
    t /= QG;
    w.ph = (int)(t % f);
    w.bz = t / f;
    const int q = qg * 64 + (tid & 63);
    w.active = q < QW;
    w.q = w.active ? q : QW - 1;
    return true;
}
constexpr long long split_col_waves(int batch, int Nr, int Nc, int f, int R) {
    return (long long)batch * f * (((Nr + f - 1) / f + R - 1) / R) * (((Nc >> 2) + 63) >> 6);
}

// ------------------------------------------------------------------------------------------------------------------
// forward, column pass: (lo, hi) planes -> A = Ly lo, H = Hy lo, V = Ly hi, D = Hy hi.  Work item: one quad column, R
// output rows f apart (phase ph, rows ph + f (blk R + m)); lanes = adjacent quads of a row (1 KiB per wavefront and load).
template <int HLEN, int R, int NT>
PDWT_DEVICE void swt_col_fwd_tile(const SwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2 - 1, NIN = R + HLEN - 1, CH = PDWT_SPLIT_CH_COLF, NCH = (NIN + CH - 1) / CH;
    const int f = a.f;
    PDWT_FOR_THREADS(tid, NT) {
        SplitColWork w;
        if (!split_col_work<R, NT>(a, block, tid, w)) continue;
        const real_t* PDWT_RESTRICT lo = a.in[0] + w.bz * a.in_bstride;
        const real_t* PDWT_RESTRICT hi = a.in[1] + w.bz * a.in_bstride;
        const int xq = 4 * w.q;
        const real_t zero = 0;
        v2f accAH[R][4], accVD[R][4];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) accAH[m][i] = accVD[m][i] = mk2(zero, zero);
        int y = true_mod(w.ph + f * (w.blk * R - c), a.Nr);  // wave-uniform
        rv4 bl[2][CH], bh[2][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                bl[0][u] = load4(lo + (long long)y * a.Nc + xq);
                bh[0][u] = load4(hi + (long long)y * a.Nc + xq);
                y = step_wrap(y, f, a.Nr);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    bl[(ch + 1) & 1][u] = load4(lo + (long long)y * a.Nc + xq);
                    bh[(ch + 1) & 1][u] = load4(hi + (long long)y * a.Nc + xq);
                    y = step_wrap(y, f, a.Nr);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - m;
                        if (j >= 0 && j < HLEN) {
                            const rv4 &vl = bl[ch & 1][u], &vh = bh[ch & 1][u];
                            accAH[m][0] = fma2_bx(mk2(vl.x, vl.y), a.t.t[j], accAH[m][0]);
                            accAH[m][1] = fma2_by(mk2(vl.x, vl.y), a.t.t[j], accAH[m][1]);
                            accAH[m][2] = fma2_bx(mk2(vl.z, vl.w), a.t.t[j], accAH[m][2]);
                            accAH[m][3] = fma2_by(mk2(vl.z, vl.w), a.t.t[j], accAH[m][3]);
                            accVD[m][0] = fma2_bx(mk2(vh.x, vh.y), a.t.t[j], accVD[m][0]);
                            accVD[m][1] = fma2_by(mk2(vh.x, vh.y), a.t.t[j], accVD[m][1]);
                            accVD[m][2] = fma2_bx(mk2(vh.z, vh.w), a.t.t[j], accVD[m][2]);
                            accVD[m][3] = fma2_by(mk2(vh.z, vh.w), a.t.t[j], accVD[m][3]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER_S(y, accAH[0][0], accAH[R - 1][0], accVD[0][0], accVD[R - 1][0]);
        }
        if (!w.active) continue;
        const long long ob = w.bz * a.out_bstride + xq;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int yo = w.ph + f * (w.blk * R + m);
            if (yo < a.Nr) {
                const long long o = ob + (long long)yo * a.Nc;
                store4(a.out[0] + o, mk4(accAH[m][0].x, accAH[m][1].x, accAH[m][2].x, accAH[m][3].x));
                store4(a.out[1] + o, mk4(accAH[m][0].y, accAH[m][1].y, accAH[m][2].y, accAH[m][3].y));
                store4(a.out[2] + o, mk4(accVD[m][0].x, accVD[m][1].x, accVD[m][2].x, accVD[m][3].x));
                store4(a.out[3] + o, mk4(accVD[m][0].y, accVD[m][1].y, accVD[m][2].y, accVD[m][3].y));
            }
        }
    }
}

// inverse, column pass: L' = (Ly A + Hy H) / 2, H' = (Ly V + Hy D) / 2, written interleaved (L', H') per column.
// Packed over two adjacent columns (the natural halves of a loaded quad) with the tap broadcast.  A pending soft
// threshold (uniform: one branch per row) is applied to the detail rows when they are consumed.
template <int HLEN, int R, int NT>
PDWT_DEVICE void swt_col_inv_tile(const SwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2, NIN = R + HLEN - 1, CH = PDWT_SPLIT_CH_COLI, NCH = (NIN + CH - 1) / CH;
    const int f = a.f;
    PDWT_FOR_THREADS(tid, NT) {
        SplitColWork w;
        if (!split_col_work<R, NT>(a, block, tid, w)) continue;
        const long long ib = w.bz * a.in_bstride;
        const real_t* PDWT_RESTRICT pA = a.in[0] + ib;
        const real_t* PDWT_RESTRICT pH = a.in[1] + ib;
        const real_t* PDWT_RESTRICT pV = a.in[2] + ib;
        const real_t* PDWT_RESTRICT pD = a.in[3] + ib;
        const int xq = 4 * w.q;
        const real_t zero = 0, half = (real_t)0.5;
        const bool soft = a.soft_beta != zero;
        v2f accL[R][2], accH[R][2];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int k = 0; k < 2; ++k) accL[m][k] = accH[m][k] = mk2(zero, zero);
        int y = true_mod(w.ph + f * (w.blk * R - c), a.Nr);  // wave-uniform
        rv4 b[2][CH][4];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                const long long o = (long long)y * a.Nc + xq;
                b[0][u][0] = load4(pA + o);
                b[0][u][1] = load4(pH + o);
                b[0][u][2] = load4(pV + o);
                b[0][u][3] = load4(pD + o);
                y = step_wrap(y, f, a.Nr);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    const long long o = (long long)y * a.Nc + xq;
                    b[(ch + 1) & 1][u][0] = load4(pA + o);
                    b[(ch + 1) & 1][u][1] = load4(pH + o);
                    b[(ch + 1) & 1][u][2] = load4(pV + o);
                    b[(ch + 1) & 1][u][3] = load4(pD + o);
                    y = step_wrap(y, f, a.Nr);
                }
            }
            PDWT_SCHED_FENCE();
            if (soft) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (ch * CH + u < NIN) {
#pragma unroll
                        for (int k = 1; k < 4; ++k) b[ch & 1][u][k] = soft4(b[ch & 1][u][k], a.soft_beta);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
                    const rv4 &vA = b[ch & 1][u][0], &vH = b[ch & 1][u][1], &vV = b[ch & 1][u][2], &vD = b[ch & 1][u][3];
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - m;
                        if (j >= 0 && j < HLEN) {
                            accL[m][0] = fma2_tx(mk2(vA.x, vA.y), a.t.t[j], accL[m][0]);
                            accL[m][1] = fma2_tx(mk2(vA.z, vA.w), a.t.t[j], accL[m][1]);
                            accL[m][0] = fma2_ty(mk2(vH.x, vH.y), a.t.t[j], accL[m][0]);
                            accL[m][1] = fma2_ty(mk2(vH.z, vH.w), a.t.t[j], accL[m][1]);
                            accH[m][0] = fma2_tx(mk2(vV.x, vV.y), a.t.t[j], accH[m][0]);
                            accH[m][1] = fma2_tx(mk2(vV.z, vV.w), a.t.t[j], accH[m][1]);
                            accH[m][0] = fma2_ty(mk2(vD.x, vD.y), a.t.t[j], accH[m][0]);
                            accH[m][1] = fma2_ty(mk2(vD.z, vD.w), a.t.t[j], accH[m][1]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER_S(y, accL[0][0], accL[R - 1][0], accH[0][0], accH[R - 1][0]);
        }
        if (!w.active) continue;
        real_t* PDWT_RESTRICT o2 = a.out[0] + w.bz * a.out_bstride + 2 * xq;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int yo = w.ph + f * (w.blk * R + m);
            if (yo < a.Nr) {
                real_t* o = o2 + (long long)yo * 2 * a.Nc;
                store4(o, mk4(half * accL[m][0].x, half * accH[m][0].x, half * accL[m][0].y, half * accH[m][0].y));
                store4(o + 4, mk4(half * accL[m][1].x, half * accH[m][1].x, half * accL[m][1].y, half * accH[m][1].y));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Work decomposition of the row kernels: a wavefront owns 64 adjacent work items of ONE row (the row address is
// wave-uniform); `trow` work items per row.
struct SplitRowWork {
    int tr, y;
    long long bz;
    bool active;
};
template <int NT>
PDWT_DEVICE bool split_row_work(const SwtSplitArgs& a, int trow, long long block, int tid, SplitRowWork& w) {
    const int TG = (trow + 63) >> 6;
    const long long wave = block * (NT / 64) + PDWT_UNIFORM(tid >> 6);
    if (wave >= (long long)a.batch * a.Nr * TG) return false;
    const int tg = (int)(wave % TG);
    const long long ty = wave / TG;
    w.y = (int)(ty % a.Nr);
    w.bz = ty / a.Nr;
    const int tr = tg * 64 + (tid & 63);
    w.active = tr < trow;
    w.tr = w.active ? tr : trow - 1;
    return true;
}
constexpr long long split_row_waves(int batch, int Nr, int trow) { return (long long)batch * Nr * ((trow + 63) >> 6); }
constexpr int split_row_items4(int Nc, int f, int R) { return ((Nc + R * f - 1) / (R * f)) * (f >> 2); }

// forward, row pass, dilation a multiple of 4: in -> lo = Lx in, hi = Hx in.  Work item: R quads f apart,
// x = b R f + 4 g + m f (block b of R f columns, quad g of the f / 4 quads of a dilation period).
template <int HLEN, int R, int NT>
PDWT_DEVICE void swt_row_fwd4_tile(const SwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2 - 1, NIN = R + HLEN - 1, CH = 2 * kSplitChunk, NCH = (NIN + CH - 1) / CH;
    const int f = a.f, G = f >> 2, span = R * f;
    const int trow = split_row_items4(a.Nc, f, R);
    PDWT_FOR_THREADS(tid, NT) {
        SplitRowWork w;
        if (!split_row_work<NT>(a, trow, block, tid, w)) continue;
        const int x0 = (w.tr / G) * span + 4 * (w.tr % G);
        const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
        const real_t zero = 0;
        v2f acc[R][4];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][i] = mk2(zero, zero);
        unsigned p = (unsigned)true_mod(x0 - c * f, a.Nc);
        rv4 b[2][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                b[0][u] = load4(row + p);
                p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    b[(ch + 1) & 1][u] = load4(row + p);
                    p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - m;
                        if (j >= 0 && j < HLEN) {
                            const rv4& v = b[ch & 1][u];
                            acc[m][0] = fma2_bx(mk2(v.x, v.y), a.t.t[j], acc[m][0]);
                            acc[m][1] = fma2_by(mk2(v.x, v.y), a.t.t[j], acc[m][1]);
                            acc[m][2] = fma2_bx(mk2(v.z, v.w), a.t.t[j], acc[m][2]);
                            acc[m][3] = fma2_by(mk2(v.z, v.w), a.t.t[j], acc[m][3]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER(p, acc[0][0], acc[R - 1][0], acc[0][3], acc[R - 1][3]);
        }
        if (!w.active) continue;
        const long long ob = w.bz * a.out_bstride + (long long)w.y * a.Nc;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int xm = x0 + m * f;
            if (xm < a.Nc) {
                store4(a.out[0] + ob + xm, mk4(acc[m][0].x, acc[m][1].x, acc[m][2].x, acc[m][3].x));
                store4(a.out[1] + ob + xm, mk4(acc[m][0].y, acc[m][1].y, acc[m][2].y, acc[m][3].y));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// inverse, row pass, dilation a multiple of 4: out = (Lx L' + Hx H') / 2 from the interleaved plane; a loaded 16 B is
// two columns' (L', H') pairs, the packed multiply-add applies (rlo, rhi) to a pair, the two halves are added at the end.
template <int HLEN, int R, int NT>
PDWT_DEVICE void swt_row_inv4_tile(const SwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2, NIN = R + HLEN - 1, CH = kSplitChunk, NCH = (NIN + CH - 1) / CH;
    const int f = a.f, G = f >> 2, span = R * f;
    const int trow = split_row_items4(a.Nc, f, R);
    PDWT_FOR_THREADS(tid, NT) {
        SplitRowWork w;
        if (!split_row_work<NT>(a, trow, block, tid, w)) continue;
        const int x0 = (w.tr / G) * span + 4 * (w.tr % G);
        const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * 2 * a.Nc;
        const real_t zero = 0, half = (real_t)0.5;
        v2f acc[R][4];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][i] = mk2(zero, zero);
        unsigned p = (unsigned)true_mod(x0 - c * f, a.Nc);
        rv4 b[2][CH][2];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                b[0][u][0] = load4(row + 2 * p);
                b[0][u][1] = load4(row + 2 * p + 4);
                p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    b[(ch + 1) & 1][u][0] = load4(row + 2 * p);
                    b[(ch + 1) & 1][u][1] = load4(row + 2 * p + 4);
                    p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
                    const rv4 &v0 = b[ch & 1][u][0], &v1 = b[ch & 1][u][1];
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - m;
                        if (j >= 0 && j < HLEN) {
                            acc[m][0] = fma2_s(mk2(v0.x, v0.y), a.t.t[j], acc[m][0]);
                            acc[m][1] = fma2_s(mk2(v0.z, v0.w), a.t.t[j], acc[m][1]);
                            acc[m][2] = fma2_s(mk2(v1.x, v1.y), a.t.t[j], acc[m][2]);
                            acc[m][3] = fma2_s(mk2(v1.z, v1.w), a.t.t[j], acc[m][3]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER(p, acc[0][0], acc[R - 1][0], acc[0][3], acc[R - 1][3]);
        }
        if (!w.active) continue;
        const long long ob = w.bz * a.out_bstride + (long long)w.y * a.Nc;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int xm = x0 + m * f;
            if (xm < a.Nc)
                store4(a.out[0] + ob + xm, mk4(half * (acc[m][0].x + acc[m][0].y), half * (acc[m][1].x + acc[m][1].y),
                                               half * (acc[m][2].x + acc[m][2].y), half * (acc[m][3].x + acc[m][3].y)));
        }
    }
}

// the same from TWO separate planes (the 1D transform's approximation and detail rows): packed over two adjacent
// columns with the tap broadcast, like the column kernel of the inverse
template <int HLEN, int R, int NT>
PDWT_DEVICE void swt_row_inv4p_tile(const SwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2, NIN = R + HLEN - 1, CH = kSplitChunk, NCH = (NIN + CH - 1) / CH;
    const int f = a.f, G = f >> 2, span = R * f;
    const int trow = split_row_items4(a.Nc, f, R);
    PDWT_FOR_THREADS(tid, NT) {
        SplitRowWork w;
        if (!split_row_work<NT>(a, trow, block, tid, w)) continue;
        const int x0 = (w.tr / G) * span + 4 * (w.tr % G);
        const real_t* PDWT_RESTRICT rowA = a.in[0] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
        const real_t* PDWT_RESTRICT rowD = a.in[1] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
        const real_t zero = 0, half = (real_t)0.5;
        v2f acc[R][2];
#pragma unroll
        for (int m = 0; m < R; ++m) acc[m][0] = acc[m][1] = mk2(zero, zero);
        unsigned p = (unsigned)true_mod(x0 - c * f, a.Nc);
        rv4 b[2][CH][2];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                b[0][u][0] = load4(rowA + p);
                b[0][u][1] = load4(rowD + p);
                p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    b[(ch + 1) & 1][u][0] = load4(rowA + p);
                    b[(ch + 1) & 1][u][1] = load4(rowD + p);
                    p = step_wrap_u(p, (unsigned)f, (unsigned)a.Nc);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
                    const rv4 &vA = b[ch & 1][u][0], &vD = b[ch & 1][u][1];
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - m;
                        if (j >= 0 && j < HLEN) {
                            acc[m][0] = fma2_tx(mk2(vA.x, vA.y), a.t.t[j], acc[m][0]);
                            acc[m][1] = fma2_tx(mk2(vA.z, vA.w), a.t.t[j], acc[m][1]);
                            acc[m][0] = fma2_ty(mk2(vD.x, vD.y), a.t.t[j], acc[m][0]);
                            acc[m][1] = fma2_ty(mk2(vD.z, vD.w), a.t.t[j], acc[m][1]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER(p, acc[0][0], acc[R - 1][0], acc[0][1], acc[R - 1][1]);
        }
        if (!w.active) continue;
        const long long ob = w.bz * a.out_bstride + (long long)w.y * a.Nc;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int xm = x0 + m * f;
            if (xm < a.Nc) store4(a.out[0] + ob + xm, mk4(half * acc[m][0].x, half * acc[m][0].y, half * acc[m][1].x, half * acc[m][1].y));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Row passes at dilation F = 1, 2, 4 through LDS.  At these dilations a work item owns 16 CONSECUTIVE columns (at
// F = 4 the four quads f apart ARE 16 consecutive columns; at 1 and 2 it streams its window of 16 + (hlen - 1) F samples:
// sample k feeds output p with tap (k - OFF - p) / F).  Straight from global memory every lane of a load or store would
// touch its own 64-B piece -- 64 cache-line requests per wavefront instruction instead of 16, and partial-line stores
// (measured: 21 us per 2048^2 launch of 40 taps, 17.6-18.9 through LDS).  Here a wavefront stages its row segment (1024 outputs + the window
// halo) in LDS with coalesced 16-B loads, the lanes read their windows from LDS (4 pad floats per 16 -- per 32 for the
// interleaved input of the inverse -- make the 8 lanes of an LDS pass hit 32 different banks), and the results go back
// through the same LDS region so that the global stores are whole 1-KiB rows.  Workgroup = NT / 64 independent
// wavefronts with a private region each; the barriers only separate the phases.
constexpr int kRowSpan = 1024;

template <int HLEN, int F>
struct RowFwdLds {
    static constexpr int c = HLEN / 2 - 1, LEAD = (c * F + 3) / 4 * 4, OFF = LEAD - c * F;
    static constexpr int NQ = (LEAD + kRowSpan + (HLEN - 1 - c) * F + 3) / 4;  // staged quads per wavefront
    static constexpr int NWQ = (LEAD + 16 + (HLEN - 1 - c) * F + 3) / 4;       // window quads per lane
    static constexpr int WAVE_FLOATS = (NQ + 3) / 4 * 20;
};
template <int HLEN, int F, int PLANES>
struct RowInvLds {
    static constexpr int AL = PLANES == 2 ? 4 : 2;  // separate planes are staged with 16-B loads of four columns
    static constexpr int c = HLEN / 2, LEAD = (c * F + AL - 1) / AL * AL, OFF = LEAD - c * F;
    static constexpr int NQ = (2 * (LEAD + kRowSpan + (HLEN - 1 - c) * F) + 3) / 4;  // staged quads (two (L', H') pairs each)
    static constexpr int NWQ = (LEAD + 16 + (HLEN - 1 - c) * F + 1) / 2;
    static constexpr int WAVE_FLOATS = (NQ + 7) / 8 * 36;
};
template <int HLEN, int F, int PLANES = 1>
constexpr int swt_row_lds_floats(bool inverse, int NT) {
    return (NT / 64) * (inverse ? RowInvLds<HLEN, F, PLANES>::WAVE_FLOATS : RowFwdLds<HLEN, F>::WAVE_FLOATS);
}

struct RowLdsWork {
    int y, xs, lane;
    long long bz;
    real_t* reg;
    bool valid;
};
template <int NT>
PDWT_DEVICE RowLdsWork row_lds_work(const SwtSplitArgs& a, long long block, int tid, real_t* smem, int wave_floats) {
    RowLdsWork w;
    const int TG = (a.Nc + kRowSpan - 1) / kRowSpan;
    const int wv = PDWT_UNIFORM(tid >> 6);
    const long long wave = block * (NT / 64) + wv;
    w.valid = wave < (long long)a.batch * a.Nr * TG;
    const long long ty = wave / TG;
    w.xs = (int)(wave % TG) * kRowSpan;
    w.y = (int)(ty % a.Nr);
    w.bz = ty / a.Nr;
    w.lane = tid & 63;
    w.reg = smem + wv * wave_floats;
    return w;
}
constexpr long long split_row_lds_waves(int batch, int Nr, int Nc) { return (long long)batch * Nr * ((Nc + kRowSpan - 1) / kRowSpan); }

template <int HLEN, int F, int NT>
PDWT_DEVICE void swt_row_fwd_lds_tile(const SwtSplitArgs& a, long long block, real_t* smem) {
    using G = RowFwdLds<HLEN, F>;
    constexpr int CH = kSplitChunk, NCH = (G::NWQ + CH - 1) / CH;
    PDWT_PER_THREAD(v2f, acc, 16, NT);
    PDWT_FOR_THREADS(tid, NT) {  // phase 1: the row segment, coalesced, into the wavefront's region
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        if (w.valid) {
            const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
#pragma unroll
            for (int k = 0; k < (G::NQ + 63) / 64; ++k) {
                const int qq = w.lane + 64 * k;
                if (qq < G::NQ) store4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3), load4(row + true_mod(w.xs - G::LEAD + 4 * qq, a.Nc)));
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {  // phase 2: 16 outputs per lane from its window
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        v2f* acc = PDWT_MINE(acc, tid);
        const real_t zero = 0;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) acc[pp] = mk2(zero, zero);
        if (w.valid) {
            unsigned lbo = 20u * (unsigned)w.lane;
            rv4 b[2][CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (u < G::NWQ) b[0][u] = load4(w.reg + lbo + 20 * (u >> 2) + 4 * (u & 3));
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int n = (ch + 1) * CH + u;
                    if (n < G::NWQ) b[(ch + 1) & 1][u] = load4(w.reg + lbo + 20 * (n >> 2) + 4 * (n & 3));
                }
                PDWT_SCHED_FENCE();
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int wq = ch * CH + u;
                    if (wq < G::NWQ) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = 4 * wq + i;
#pragma unroll
                            for (int pp = 0; pp < 16; ++pp) {
                                const int d = k - G::OFF - pp;
                                if (d >= 0 && d % F == 0 && d / F < HLEN) {
                                    const rv4& v = b[ch & 1][u];
                                    const v2f pr = i < 2 ? mk2(v.x, v.y) : mk2(v.z, v.w);
                                    acc[pp] = (i & 1) ? fma2_by(pr, a.t.t[d / F], acc[pp]) : fma2_bx(pr, a.t.t[d / F], acc[pp]);
                                }
                            }
                        }
                    }
                }
                PDWT_SCHED_FENCE();
                PDWT_ORDER_AFTER(lbo, acc[0], acc[1], acc[14], acc[15]);
            }
        }
    }
    PDWT_SYNC();
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {  // phase 3: lo, then hi, through the region to whole-row stores
        PDWT_FOR_THREADS(tid, NT) {
            const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
            const v2f* acc = PDWT_MINE(acc, tid);
            if (w.valid) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    store4(w.reg + 20 * w.lane + 4 * m,
                           plane == 0 ? mk4(acc[4 * m].x, acc[4 * m + 1].x, acc[4 * m + 2].x, acc[4 * m + 3].x)
                                      : mk4(acc[4 * m].y, acc[4 * m + 1].y, acc[4 * m + 2].y, acc[4 * m + 3].y));
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {
            const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
            if (w.valid) {
                real_t* PDWT_RESTRICT orow = a.out[plane] + w.bz * a.out_bstride + (long long)w.y * a.Nc;
#pragma unroll
                for (int k = 0; k < kRowSpan / 256; ++k) {
                    const int qq = w.lane + 64 * k, col = w.xs + 4 * qq;
                    if (col < a.Nc) store4(orow + col, load4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3)));
                }
            }
        }
        PDWT_SYNC();
    }
}

// inverse twin.  PLANES = 1: a.in[0] is the interleaved (L', H') plane of the column kernel; PLANES = 2: a.in[0], a.in[1]
// are separate planes (the 1D transform's approximation and detail rows), interleaved while they are staged.
template <int HLEN, int F, int PLANES, int NT>
PDWT_DEVICE void swt_row_inv_lds_tile(const SwtSplitArgs& a, long long block, real_t* smem) {
    using G = RowInvLds<HLEN, F, PLANES>;
    constexpr int CH = 2 * kSplitChunk, NCH = (G::NWQ + CH - 1) / CH;
    PDWT_PER_THREAD(v2f, acc, 16, NT);
    PDWT_FOR_THREADS(tid, NT) {
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        if (w.valid) {
            if (PLANES == 1) {
                const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * 2 * a.Nc;
#pragma unroll
                for (int k = 0; k < (G::NQ + 63) / 64; ++k) {
                    const int qq = w.lane + 64 * k;  // pairs 2 qq, 2 qq + 1 of the window
                    if (qq < G::NQ)
                        store4(w.reg + 36 * (qq >> 3) + 4 * (qq & 7), load4(row + 2 * true_mod(w.xs - G::LEAD + 2 * qq, a.Nc)));
                }
            } else {
                const real_t* PDWT_RESTRICT r0 = a.in[0] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
                const real_t* PDWT_RESTRICT r1 = a.in[1] + w.bz * a.in_bstride + (long long)w.y * a.Nc;
#pragma unroll
                for (int k = 0; k < (G::NQ / 2 + 1 + 63) / 64; ++k) {
                    const int q4i = w.lane + 64 * k;  // columns 4 q4i .. 4 q4i + 3 of the window: staged quads 2 q4i, 2 q4i + 1
                    if (2 * q4i < G::NQ) {
                        const int p = true_mod(w.xs - G::LEAD + 4 * q4i, a.Nc);  // LEAD is a multiple of 4 here (RowInvLds::AL)
                        const rv4 v0 = load4(r0 + p), v1 = load4(r1 + p);
                        const int qa = 2 * q4i, qb = 2 * q4i + 1;
                        store4(w.reg + 36 * (qa >> 3) + 4 * (qa & 7), mk4(v0.x, v1.x, v0.y, v1.y));
                        if (qb < G::NQ) store4(w.reg + 36 * (qb >> 3) + 4 * (qb & 7), mk4(v0.z, v1.z, v0.w, v1.w));
                    }
                }
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        v2f* acc = PDWT_MINE(acc, tid);
        const real_t zero = 0;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) acc[pp] = mk2(zero, zero);
        if (w.valid) {
            unsigned lbo = 36u * (unsigned)w.lane;  // the lane's 16 columns = 32 floats = one padded block
            rv4 b[2][CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (u < G::NWQ) b[0][u] = load4(w.reg + lbo + 36 * (u >> 3) + 4 * (u & 7));
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int n = (ch + 1) * CH + u;
                    if (n < G::NWQ) b[(ch + 1) & 1][u] = load4(w.reg + lbo + 36 * (n >> 3) + 4 * (n & 7));
                }
                PDWT_SCHED_FENCE();
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int wq = ch * CH + u;
                    if (wq < G::NWQ) {
                        const rv4& v = b[ch & 1][u];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int k = 2 * wq + i;
                            const v2f pr = i == 0 ? mk2(v.x, v.y) : mk2(v.z, v.w);
#pragma unroll
                            for (int pp = 0; pp < 16; ++pp) {
                                const int d = k - G::OFF - pp;
                                if (d >= 0 && d % F == 0 && d / F < HLEN) acc[pp] = fma2_s(pr, a.t.t[d / F], acc[pp]);
                            }
                        }
                    }
                }
                PDWT_SCHED_FENCE();
                PDWT_ORDER_AFTER(lbo, acc[0], acc[1], acc[14], acc[15]);
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        const v2f* acc = PDWT_MINE(acc, tid);
        const real_t half = (real_t)0.5;
        if (w.valid) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                store4(w.reg + 20 * w.lane + 4 * m,
                       mk4(half * (acc[4 * m].x + acc[4 * m].y), half * (acc[4 * m + 1].x + acc[4 * m + 1].y),
                           half * (acc[4 * m + 2].x + acc[4 * m + 2].y), half * (acc[4 * m + 3].x + acc[4 * m + 3].y)));
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const RowLdsWork w = row_lds_work<NT>(a, block, tid, smem, G::WAVE_FLOATS);
        if (w.valid) {
            real_t* PDWT_RESTRICT orow = a.out[0] + w.bz * a.out_bstride + (long long)w.y * a.Nc;
#pragma unroll
            for (int k = 0; k < kRowSpan / 256; ++k) {
                const int qq = w.lane + 64 * k, col = w.xs + 4 * qq;
                if (col < a.Nc) store4(orow + col, load4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3)));
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int R, int NT>
__global__ void __launch_bounds__(NT) swt_col_fwd_kernel(const SwtSplitArgs a) { swt_col_fwd_tile<HLEN, R, NT>(a, blockIdx.x); }
template <int HLEN, int R, int NT>
__global__ void __launch_bounds__(NT) swt_col_inv_kernel(const SwtSplitArgs a) { swt_col_inv_tile<HLEN, R, NT>(a, blockIdx.x); }
template <int HLEN, int R, int NT>
__global__ void __launch_bounds__(NT) swt_row_fwd4_kernel(const SwtSplitArgs a) { swt_row_fwd4_tile<HLEN, R, NT>(a, blockIdx.x); }
template <int HLEN, int R, int NT>
__global__ void __launch_bounds__(NT) swt_row_inv4_kernel(const SwtSplitArgs a) { swt_row_inv4_tile<HLEN, R, NT>(a, blockIdx.x); }
template <int HLEN, int R, int NT>
__global__ void __launch_bounds__(NT) swt_row_inv4p_kernel(const SwtSplitArgs a) { swt_row_inv4p_tile<HLEN, R, NT>(a, blockIdx.x); }
template <int HLEN, int F, int NT>
__global__ void __launch_bounds__(NT) swt_row_fwd_lds_kernel(const SwtSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    swt_row_fwd_lds_tile<HLEN, F, NT>(a, blockIdx.x, pdwt_smem);
}
template <int HLEN, int F, int PLANES, int NT>
__global__ void __launch_bounds__(NT) swt_row_inv_lds_kernel(const SwtSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    swt_row_inv_lds_tile<HLEN, F, PLANES, NT>(a, blockIdx.x, pdwt_smem);
}
#endif

}  // namespace pdwt
