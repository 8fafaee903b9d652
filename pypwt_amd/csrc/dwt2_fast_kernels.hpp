// dwt2_fast_kernels.hpp -- tuned fused 2D DWT level kernels for gfx950 (the hot path of
// BASELINE config 2/5: 4096x4096 db4).
//
// Same tile decomposition and arithmetic as dwt2_kernels.hpp (which stays as the generic path for
// odd filter lengths, row lengths that are not a multiple of 4, and unaligned buffers), rebuilt
// around what rocprofv3 showed on the first version (profiles/r01a_*): the level-1 kernel was
// VALU-issue bound (~75 vector instructions per input sample, most of them per-element index
// arithmetic in the staging loop), not HBM bound.  Changes:
//   * staging uses aligned 16-B global loads and ds_write_b128; periodization is resolved per
//     float4 (a row of 4k samples never straddles the wrap) and per staged row, not per sample;
//   * the low/high intermediates are kept INTERLEAVED in LDS as (L,H) pairs and every
//     multiply-add is a packed v_pk_fma_f32 on such a pair: taps come from the kernel-argument
//     segment as (lo,hi) SGPR pairs, broadcast through op_sel, two FMAs per instruction;
//   * each thread produces two adjacent outputs per LDS read group (ds_read_b128 / _b64);
//   * workgroups are renumbered so that the 8 XCDs each work on a contiguous band of tiles and
//     halo rows are re-read from that XCD's own L2.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

#ifdef PDWT_CPU_EMU
struct v2f {
    float x, y;
};
static inline v2f mk2(float a, float b) { return v2f{a, b}; }
static inline v2f fma2(v2f a, v2f b, v2f c) { return v2f{a.x * b.x + c.x, a.y * b.y + c.y}; }
#else
typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2f mk2(float a, float b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif
static PDWT_DEVICE v2f bc(float a) { return mk2(a, a); }

// Reads 16 B from LDS as ONE ds_read_b128 (256 B/clk) even when only some components are used
// afterwards; without the barrier hipcc narrows it to ds_read2_b32 / ds_read2_b64 pairs, which
// run at half the LDS rate (MI355X_MICROARCH.md, LDS table).
#ifdef PDWT_CPU_EMU
typedef f32x4 v4f;
static inline v4f lds_read16(const void* p) { return *reinterpret_cast<const v4f*>(p); }
#else
typedef float v4f __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ v4f lds_read16(const void* p) {
    v4f w = *reinterpret_cast<const v4f*>(p);
    asm volatile("" : "+v"(w));
    return w;
}
#endif

// taps interleaved as (lo[j], hi[j]) pairs
struct FilterBankI {
    v2f t[kMaxTaps];
};

struct Fwd2DFastArgs {
    const float* in;
    float *A, *H, *V, *D;
    int Nr, Nc, Nr2, Nc2;
    long long in_bstride, out_bstride;
    int tiles_x, tiles_y;  // tile grid of one image
    FilterBankI fb;        // (dec_lo, dec_hi)
};

struct Inv2DFastArgs {
    const float *A, *H, *V, *D;
    float* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int tiles_x, tiles_y;
    FilterBankI fb;  // (rec_lo, rec_hi)
};

// Block renumbering: hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
// ids b and b+8 share an L2.  Give XCD x the contiguous tile range [x*chunk, (x+1)*chunk) (row-major),
// i.e. a horizontal band of the image, so vertically adjacent tiles (which share halo rows) meet
// in the same L2.  Placement only affects speed, never results.  Returns false for padding ids.
PDWT_DEVICE bool xcd_tile(int block, int tiles_x, int tiles_y, int& bx, int& by) {
    const int total = tiles_x * tiles_y;
    const int chunk = (total + 7) >> 3;
    const int tile = (block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || tile >= total) return false;
    by = tile / tiles_x;
    bx = tile - by * tiles_x;
    return true;
}

template <int HLEN, int TX>
struct FwdFastGeom {
    static constexpr int C = HLEN / 2 - 1;           // analysis centre
    static constexpr int PADL = (4 - (C & 3)) & 3;   // samples between the aligned load origin and x0
    static constexpr int RXA = (PADL + 2 * TX + HLEN - 2 + 3) & ~3;  // staged row length (multiple of 4)
    static constexpr int NV = (PADL + HLEN + 2 + 3) & ~3;            // LDS floats read per thread per row
};

template <int HLEN, int TX, int TY>
constexpr int fwd2d_fast_lds_floats() {
    return (2 * TY + HLEN - 2) * (FwdFastGeom<HLEN, TX>::RXA + 2 * TX);
}

// Requirements (checked by the host): HLEN even, Nc % 4 == 0, 16-B aligned image rows.
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_fwd_fast_tile(const Fwd2DFastArgs& a, int bx, int by, int bz, float* smem) {
    using G = FwdFastGeom<HLEN, TX>;
    constexpr int C = G::C, PADL = G::PADL, RXA = G::RXA, NV = G::NV;
    constexpr int RY = 2 * TY + HLEN - 2;
    constexpr int V4 = RXA / 4;        // float4 per staged row
    constexpr int HT = TX / 2;         // threads per row in the row pass (2 outputs each)
    static_assert(NT % HT == 0, "thread layout");
    constexpr int NG = NT / HT;        // thread groups along y in the column pass
    static_assert(TY % NG == 0, "tile height must split over the thread groups");
    constexpr int R = TY / NG;         // output rows per thread (x 2 columns)
    static_assert(2 * TX - 4 + NV <= RXA, "row-pass reads stay inside the staged row");

    float* sIn = smem;                                   // RY x RXA floats
    v2f* tLH = reinterpret_cast<v2f*>(smem + RY * RXA);  // RY x TX (L,H) pairs

    const float* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const int xa = 2 * bx * TX - C - PADL;  // multiple of 4
    const int y0 = 2 * by * TY - C;

    // ---- phase 1: stage rows [y0, y0+RY) x cols [xa, xa+RXA) with 16-B loads
    PDWT_FOR_THREADS(tid, NT) {
        for (int idx = tid; idx < RY * V4; idx += NT) {
            const int r = idx / V4;
            const int g = idx - r * V4;
            const int sy = wrap_analysis(y0 + r, a.Nr);
            const int sx = wrap_periodic(xa + 4 * g, a.Nc);  // Nc % 4 == 0: the group never straddles
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + (long long)sy * a.Nc + sx);
            *reinterpret_cast<f32x4*>(sIn + r * RXA + 4 * g) = v;
        }
    }
    PDWT_SYNC();

    // ---- phase 2: row analysis, two adjacent outputs per thread, packed (L,H) accumulators
    PDWT_FOR_THREADS(tid, NT) {
        const int t = tid % HT;
        for (int r = tid / HT; r < RY; r += NT / HT) {
            float v[NV];
            const float* p4 = sIn + r * RXA + 4 * t;
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) {
                const v4f w = lds_read16(p4 + 4 * q);
                v[4 * q + 0] = w.x;
                v[4 * q + 1] = w.y;
                v[4 * q + 2] = w.z;
                v[4 * q + 3] = w.w;
            }
            v2f acc0 = mk2(0.f, 0.f), acc1 = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < HLEN; ++j) {
                const v2f tap = a.fb.t[HLEN - 1 - j];
                acc0 = fma2(bc(v[PADL + j]), tap, acc0);
                acc1 = fma2(bc(v[PADL + 2 + j]), tap, acc1);
            }
            f32x4 o;
            o.x = acc0.x;
            o.y = acc0.y;
            o.z = acc1.x;
            o.w = acc1.y;
            *reinterpret_cast<f32x4*>(&tLH[r * TX + 2 * t]) = o;
        }
    }
    PDWT_SYNC();

    // ---- phase 3: column analysis; (L,H) pair x tap -> (A,V) with lo, (H,D) with hi.
    // Each thread owns two adjacent columns (one ds_read_b128 per staged row, 8-B stores).
    PDWT_FOR_THREADS(tid, NT) {
        const int t = tid % HT;
        const int ty0 = (tid / HT) * R;
        const int ox = bx * TX + 2 * t;
        v2f accAV[R][2], accHD[R][2];
#pragma unroll
        for (int i = 0; i < R; ++i) accAV[i][0] = accAV[i][1] = accHD[i][0] = accHD[i][1] = mk2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 2 * R + HLEN - 2; ++r) {
            const v4f w = lds_read16(&tLH[(2 * ty0 + r) * TX + 2 * t]);
            const v2f lh0 = mk2(w.x, w.y), lh1 = mk2(w.z, w.w);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int j = r - 2 * i;
                if (j >= 0 && j < HLEN) {
                    const v2f tap = a.fb.t[HLEN - 1 - j];
                    accAV[i][0] = fma2(lh0, bc(tap.x), accAV[i][0]);
                    accHD[i][0] = fma2(lh0, bc(tap.y), accHD[i][0]);
                    accAV[i][1] = fma2(lh1, bc(tap.x), accAV[i][1]);
                    accHD[i][1] = fma2(lh1, bc(tap.y), accHD[i][1]);
                }
            }
        }
        const long long boff = (long long)bz * a.out_bstride;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int oy = by * TY + ty0 + i;
            if (oy < a.Nr2 && ox < a.Nc2) {  // Nc2 is even: ox + 1 is inside too
                const long long o = boff + (long long)oy * a.Nc2 + ox;
                f32x2 v;
                v.x = accAV[i][0].x; v.y = accAV[i][1].x; *reinterpret_cast<f32x2*>(a.A + o) = v;
                v.x = accAV[i][0].y; v.y = accAV[i][1].y; *reinterpret_cast<f32x2*>(a.V + o) = v;
                v.x = accHD[i][0].x; v.y = accHD[i][1].x; *reinterpret_cast<f32x2*>(a.H + o) = v;
                v.x = accHD[i][0].y; v.y = accHD[i][1].y; *reinterpret_cast<f32x2*>(a.D + o) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------
template <int HLEN, int TX>
struct InvFastGeom {
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int S = (H2 & 1) ? 0 : 1;
    static constexpr int PADL = (4 - (C & 3)) & 3;  // cx0 = bx*TX - C ; aligned origin = cx0 - PADL (multiple of 4)
    static constexpr int CXA = (PADL + TX + H2 + 1 + 3) & ~3;
};

template <int HLEN, int TX, int TY>
constexpr int inv2d_fast_lds_floats() {
    using G = InvFastGeom<HLEN, TX>;
    return 4 * (TY + G::H2 + 1) * G::CXA + 2 * (2 * TY) * G::CXA;
}

// Requirements: HLEN even, Ncc % 4 == 0 (so Nc = 2*Ncc or 2*Ncc-1 ... the host requires Nc == 2*Ncc),
// 16-B aligned coefficient rows.
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_inv_fast_tile(const Inv2DFastArgs& a, int bx, int by, int bz, float* smem) {
    using G = InvFastGeom<HLEN, TX>;
    constexpr int H2 = G::H2, C = G::C, S = G::S, PADL = G::PADL, CXA = G::CXA;
    constexpr int CR = TY + H2 + 1;
    constexpr int V4 = CXA / 4;
    constexpr int OY = 2 * TY;

    v2f* sAV = reinterpret_cast<v2f*>(smem);  // CR x CXA (A,V) pairs
    v2f* sHD = sAV + CR * CXA;                // CR x CXA (H,D) pairs
    v2f* tt = sHD + CR * CXA;                 // OY x CXA (t1,t2) pairs

    const long long boff = (long long)bz * a.in_bstride;
    const int cy0 = by * TY - C;
    const int cxa = bx * TX - C - PADL;  // multiple of 4

    // ---- phase 1: stage the four coefficient tiles, interleaved as (A,V) and (H,D)
    PDWT_FOR_THREADS(tid, NT) {
        for (int idx = tid; idx < CR * V4; idx += NT) {
            const int r = idx / V4;
            const int g = idx - r * V4;
            const int sy = wrap_periodic(cy0 + r, a.Nrc);
            const int sx = wrap_periodic(cxa + 4 * g, a.Ncc);
            const long long o = boff + (long long)sy * a.Ncc + sx;
            const f32x4 vA = *reinterpret_cast<const f32x4*>(a.A + o);
            const f32x4 vV = *reinterpret_cast<const f32x4*>(a.V + o);
            const f32x4 vH = *reinterpret_cast<const f32x4*>(a.H + o);
            const f32x4 vD = *reinterpret_cast<const f32x4*>(a.D + o);
            f32x4 w;
            f32x4* dAV = reinterpret_cast<f32x4*>(sAV + r * CXA + 4 * g);
            f32x4* dHD = reinterpret_cast<f32x4*>(sHD + r * CXA + 4 * g);
            w.x = vA.x; w.y = vV.x; w.z = vA.y; w.w = vV.y; dAV[0] = w;
            w.x = vA.z; w.y = vV.z; w.z = vA.w; w.w = vV.w; dAV[1] = w;
            w.x = vH.x; w.y = vD.x; w.z = vH.y; w.w = vD.y; dHD[0] = w;
            w.x = vH.z; w.y = vD.z; w.z = vH.w; w.w = vD.w; dHD[1] = w;
        }
    }
    PDWT_SYNC();

    // ---- phase 2: column synthesis.  Work item = (m, column pair): the two output rows p = 2m, 2m+1
    // (p = gy + S) share the coefficient rows m .. m+H2-1 (local); even taps feed p odd, odd taps p even.
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int NM = TY + S;  // m = 0 .. TY-1+S
        constexpr int Q2 = CXA / 2;
        for (int idx = tid; idx < NM * Q2; idx += NT) {
            const int m = idx / Q2;
            const int q = 2 * (idx - m * Q2);
            v2f e0 = mk2(0.f, 0.f), o0 = e0, e1 = e0, o1 = e0;  // p even / p odd, columns q / q+1
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const v4f wav = lds_read16(&sAV[(m + j) * CXA + q]);
                const v4f whd = lds_read16(&sHD[(m + j) * CXA + q]);
                const v2f te = a.fb.t[HLEN - 2 - 2 * j];  // p even: par = 1
                const v2f to = a.fb.t[HLEN - 1 - 2 * j];  // p odd : par = 0
                const v2f av0 = mk2(wav.x, wav.y), av1 = mk2(wav.z, wav.w);
                const v2f hd0 = mk2(whd.x, whd.y), hd1 = mk2(whd.z, whd.w);
                e0 = fma2(av0, bc(te.x), e0); e0 = fma2(hd0, bc(te.y), e0);
                o0 = fma2(av0, bc(to.x), o0); o0 = fma2(hd0, bc(to.y), o0);
                e1 = fma2(av1, bc(te.x), e1); e1 = fma2(hd1, bc(te.y), e1);
                o1 = fma2(av1, bc(to.x), o1); o1 = fma2(hd1, bc(to.y), o1);
            }
            const int ge = 2 * m - S, go = 2 * m + 1 - S;  // local output rows
            f32x4 w;
            if (ge >= 0 && ge < OY) {
                w.x = e0.x; w.y = e0.y; w.z = e1.x; w.w = e1.y;
                *reinterpret_cast<f32x4*>(&tt[ge * CXA + q]) = w;
            }
            if (go >= 0 && go < OY) {
                w.x = o0.x; w.y = o0.y; w.z = o1.x; w.w = o1.y;
                *reinterpret_cast<f32x4*>(&tt[go * CXA + q]) = w;
            }
        }
    }
    PDWT_SYNC();

    // ---- phase 3: row synthesis, four adjacent samples per thread, 16-B stores
    PDWT_FOR_THREADS(tid, NT) {
        float* PDWT_RESTRICT out = a.out + (long long)bz * a.out_bstride;
        constexpr int HT = TX / 2;
        constexpr int PE = PADL & 1;                    // read origin rounded down to an even pair index
        constexpr int NP = (PE + H2 + 2 + 1) & ~1;      // (t1,t2) pairs read per thread
        for (int idx = tid; idx < OY * HT; idx += NT) {
            const int gy = idx / HT;
            const int k = 2 * (idx - gy * HT);
            v2f u[NP];
            const v2f* base = tt + gy * CXA + (PADL - PE) + k;
#pragma unroll
            for (int q = 0; q < NP / 2; ++q) {
                const v4f w = lds_read16(base + 2 * q);
                u[2 * q] = mk2(w.x, w.y);
                u[2 * q + 1] = mk2(w.z, w.w);
            }
            float res[4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {  // coefficient column k + kk -> samples 2(k+kk), 2(k+kk)+1
                v2f r0 = mk2(0.f, 0.f), r1 = mk2(0.f, 0.f);
                if (S == 0) {
#pragma unroll
                    for (int j = 0; j < H2; ++j) {
                        const v2f w = u[PE + kk + j];
                        r0 = fma2(w, a.fb.t[HLEN - 2 - 2 * j], r0);  // p even
                        r1 = fma2(w, a.fb.t[HLEN - 1 - 2 * j], r1);  // p odd
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < H2 + 1; ++j) {
                        const v2f w = u[PE + kk + j];
                        if (j < H2) r0 = fma2(w, a.fb.t[HLEN - 1 - 2 * j], r0);        // p = 2k+1 (odd), base k
                        if (j >= 1) r1 = fma2(w, a.fb.t[HLEN - 2 - 2 * (j - 1)], r1);  // p = 2k+2 (even), base k+1
                    }
                }
                res[2 * kk] = r0.x + r0.y;
                res[2 * kk + 1] = r1.x + r1.y;
            }
            const int oy = 2 * by * TY + gy;
            const int ox = 2 * (bx * TX + k);
            if (oy < a.Nr && ox < a.Nc) {  // Nc % 8 == 0 (Ncc % 4 == 0): the float4 is inside and aligned
                f32x4 v;
                v.x = res[0]; v.y = res[1]; v.z = res[2]; v.w = res[3];
                *reinterpret_cast<f32x4*>(out + (long long)oy * a.Nc + ox) = v;
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_fast_kernel(const Fwd2DFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int bx, by;
    if (!xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_fwd_fast_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_fast_kernel(const Inv2DFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int bx, by;
    if (!xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_inv_fast_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
