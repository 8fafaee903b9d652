// dwt2_kernels.hpp -- fused separable 2D DWT level kernels for gfx950.
//
// One workgroup (NT threads = NT/64 wavefronts) produces one TY x TX tile of
// each of the four sub-bands (forward) or one 2TY x 2TX tile of the image
// (inverse).  Row pass and column pass of a level are fused: the input tile and
// its (hlen-2)-wide halo are staged in LDS once, the intermediate low/high
// planes never leave LDS, and each level reads its input once and writes its
// outputs once (the reference runs two kernels per level through a global
// temporary, pdwt/src/separable.cu:196-207 and :351-361).
//
// Filter length is a template parameter (HLEN > 0): every tap loop is fully
// unrolled and the taps come out of the kernel-argument segment as scalar
// loads.  HLEN == 0 is the generic runtime-length path (odd / custom lengths).
//
// What each tile computes (restated in oracle/pdwt_oracle.c and pinned to pywt):
//   analysis   out[k] = sum_j x[per(2k - c + j)] * f[hlen-1-j]          (reference
//              semantics: pdwt/src/separable.cu:91-131 rows, :135-176 columns)
//   synthesis  polyphase form of pdwt/src/separable.cu:246-328, see inv tile.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

// LDS floats needed by the forward tile
template <int TX, int TY>
constexpr int fwd2d_lds_floats(int hlen) {
    const int RY = 2 * TY + hlen - 2;
    const int RXp = (2 * TX + hlen - 2 + 1) & ~1;
    return 2 * kMaxTaps + RY * RXp + 2 * RY * TX;
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_fwd_tile(const Fwd2DArgs& a, int bx, int by, int bz, real_t* smem) {
    static_assert(NT % TX == 0, "a wavefront row must cover the tile width");
    constexpr int NG = NT / TX;  // thread groups stacked along y
    static_assert(TY % NG == 0, "tile height must split over the thread groups");
    constexpr int R = TY / NG;  // output rows per thread in the column pass

    const int hlen = HLEN ? HLEN : a.hlen;
    const int c = analysis_centre(hlen);
    const int RY = 2 * TY + hlen - 2;          // input rows staged
    const int RXp = (2 * TX + hlen - 2 + 1) & ~1;  // input cols staged (even stride: 8-B aligned pairs)

    real_t* sTaps = smem;
    real_t* sIn = smem + 2 * kMaxTaps;
    real_t* tL = sIn + RY * RXp;
    real_t* tH = tL + RY * TX;

    const real_t* lo = a.fb.lo;
    const real_t* hi = a.fb.hi;
    if (HLEN == 0) {
        PDWT_FOR_THREADS(tid, NT) {
            if (tid < kMaxTaps) {
                sTaps[tid] = a.fb.lo[tid];
                sTaps[kMaxTaps + tid] = a.fb.hi[tid];
            }
        }
        lo = sTaps;
        hi = sTaps + kMaxTaps;
    }

    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const int x0 = 2 * bx * TX - c;
    const int y0 = 2 * by * TY - c;

    // ---- phase 1: stage the input tile + halo, periodized, coalesced along x
    // eight loads per thread are issued before the first of them is stored (a loop with an exit test per element waits
    // for every load before it issues the next: DESIGN.md, "staging loop")
    PDWT_FOR_THREADS(tid, NT) {
        const int total = RY * RXp;
        for (int base = 0; base < total; base += 8 * NT) {
            real_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int idx = base + u * NT + tid;
                idx = idx < total ? idx : total - 1;
                const int r = idx / RXp;
                const int q = idx - r * RXp;
                const int sy = wrap_analysis(y0 + r, a.Nr);
                const int sx = wrap_analysis(x0 + q, a.Nc);
                v[u] = in[(long long)sy * a.Nc + sx];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int idx = base + u * NT + tid;
                idx = idx < total ? idx : total - 1;
                sIn[idx] = v[u];
            }
        }
    }
    PDWT_SYNC();

    // ---- phase 2: row analysis + decimation -> tL, tH (RY x TX), in LDS
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        for (int r = tid / TX; r < RY; r += NG) {
            const real_t* p = sIn + r * RXp + 2 * k;
            real_t aL = 0.f, aH = 0.f;
            if (HLEN > 0 && (HLEN % 2) == 0) {
                const real2_t* p2 = reinterpret_cast<const real2_t*>(p);
#pragma unroll
                for (int m = 0; m < (HLEN > 0 ? HLEN / 2 : 1); ++m) {
                    const real2_t v = p2[m];
                    aL = pdwt_fma(v.x, lo[hlen - 1 - 2 * m], aL);
                    aH = pdwt_fma(v.x, hi[hlen - 1 - 2 * m], aH);
                    aL = pdwt_fma(v.y, lo[hlen - 2 - 2 * m], aL);
                    aH = pdwt_fma(v.y, hi[hlen - 2 - 2 * m], aH);
                }
            } else {
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = p[j];
                    aL = pdwt_fma(v, lo[hlen - 1 - j], aL);
                    aH = pdwt_fma(v, hi[hlen - 1 - j], aH);
                }
            }
            tL[r * TX + k] = aL;
            tH[r * TX + k] = aH;
        }
    }
    PDWT_SYNC();

    // ---- phase 3: column analysis + decimation -> A,H,V,D, coalesced stores
    PDWT_FOR_THREADS(tid, NT) {
        const int k = tid % TX;
        const int ty0 = (tid / TX) * R;
        const int ox = bx * TX + k;
        real_t* PDWT_RESTRICT oA = a.A + (long long)bz * a.out_bstride;
        real_t* PDWT_RESTRICT oH = a.H + (long long)bz * a.out_bstride;
        real_t* PDWT_RESTRICT oV = a.V + (long long)bz * a.out_bstride;
        real_t* PDWT_RESTRICT oD = a.D + (long long)bz * a.out_bstride;
        if (HLEN > 0) {
            // sliding window: each staged row is read once and feeds every output
            // row whose support covers it
            real_t accA[R], accH[R], accV[R], accD[R];
#pragma unroll
            for (int i = 0; i < R; ++i) accA[i] = accH[i] = accV[i] = accD[i] = 0.f;
#pragma unroll
            for (int r = 0; r < 2 * R + (HLEN > 0 ? HLEN : 2) - 2; ++r) {
                const real_t l = tL[(2 * ty0 + r) * TX + k];
                const real_t h = tH[(2 * ty0 + r) * TX + k];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int j = r - 2 * i;
                    if (j >= 0 && j < hlen) {
                        const real_t tl = lo[hlen - 1 - j], th = hi[hlen - 1 - j];
                        accA[i] = pdwt_fma(l, tl, accA[i]);
                        accH[i] = pdwt_fma(l, th, accH[i]);
                        accV[i] = pdwt_fma(h, tl, accV[i]);
                        accD[i] = pdwt_fma(h, th, accD[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int oy = by * TY + ty0 + i;
                if (oy < a.Nr2 && ox < a.Nc2) {
                    const long long o = (long long)oy * a.Nc2 + ox;
                    oA[o] = accA[i];
                    oH[o] = accH[i];
                    oV[o] = accV[i];
                    oD[o] = accD[i];
                }
            }
        } else {
            for (int i = 0; i < R; ++i) {
                real_t rA = 0.f, rH = 0.f, rV = 0.f, rD = 0.f;
                for (int j = 0; j < hlen; ++j) {
                    const real_t l = tL[(2 * (ty0 + i) + j) * TX + k];
                    const real_t h = tH[(2 * (ty0 + i) + j) * TX + k];
                    const real_t tl = lo[hlen - 1 - j], th = hi[hlen - 1 - j];
                    rA = pdwt_fma(l, tl, rA);
                    rH = pdwt_fma(l, th, rH);
                    rV = pdwt_fma(h, tl, rV);
                    rD = pdwt_fma(h, th, rD);
                }
                const int oy = by * TY + ty0 + i;
                if (oy < a.Nr2 && ox < a.Nc2) {
                    const long long o = (long long)oy * a.Nc2 + ox;
                    oA[o] = rA;
                    oH[o] = rH;
                    oV[o] = rV;
                    oD[o] = rD;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Inverse level.  Polyphase synthesis with h2 = hlen/2, c = h2/2 and
// s = (h2 even ? 1 : 0):   for output sample g let p = g + s, then
//   out[g] = sum_{j<h2}  a[(p/2 - c + j) mod Nin] * rlo[t] + d[...] * rhi[t],
//   t = hlen - 1 - (2j + 1 - (p & 1)).
// Column synthesis first ((A,H) -> t1, (V,D) -> t2), then row synthesis, as the
// reference orders them (pdwt/src/separable.cu:351-361).
// ---------------------------------------------------------------------------
template <int TX, int TY>
constexpr int inv2d_lds_floats(int hlen) {
    const int h2 = hlen / 2;
    const int CR = TY + h2 + 1;
    const int CXp = TX + h2 + 1;
    return 2 * kMaxTaps + 4 * CR * CXp + 2 * (2 * TY) * CXp;
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_inv_tile(const Inv2DArgs& a, int bx, int by, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen;
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int s = (h2 & 1) ? 0 : 1;
    const int CR = TY + h2 + 1;   // coefficient rows staged
    const int CXp = TX + h2 + 1;  // coefficient cols staged
    const int OY = 2 * TY;

    real_t* sTaps = smem;
    real_t* sA = smem + 2 * kMaxTaps;
    real_t* sH = sA + CR * CXp;
    real_t* sV = sH + CR * CXp;
    real_t* sD = sV + CR * CXp;
    real_t* t1 = sD + CR * CXp;  // OY x CXp
    real_t* t2 = t1 + OY * CXp;

    const real_t* lo = a.fb.lo;
    const real_t* hi = a.fb.hi;
    if (HLEN == 0) {
        PDWT_FOR_THREADS(tid, NT) {
            if (tid < kMaxTaps) {
                sTaps[tid] = a.fb.lo[tid];
                sTaps[kMaxTaps + tid] = a.fb.hi[tid];
            }
        }
        lo = sTaps;
        hi = sTaps + kMaxTaps;
    }

    const long long boff = (long long)bz * a.in_bstride;
    const real_t* PDWT_RESTRICT gA = a.A + boff;
    const real_t* PDWT_RESTRICT gH = a.H + boff;
    const real_t* PDWT_RESTRICT gV = a.V + boff;
    const real_t* PDWT_RESTRICT gD = a.D + boff;

    const int cy0 = by * TY - c;  // first coefficient row staged
    const int cx0 = bx * TX - c;

    // ---- phase 1: stage the four coefficient tiles (+ halo), periodic
    PDWT_FOR_THREADS(tid, NT) {  // two elements of all four bands (eight loads) in flight per thread
        const int total = CR * CXp;
        for (int base = 0; base < total; base += 2 * NT) {
            real_t v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int idx = base + u * NT + tid;
                idx = idx < total ? idx : total - 1;
                const int r = idx / CXp;
                const int q = idx - r * CXp;
                const int sy = wrap_periodic(cy0 + r, a.Nrc);
                const int sx = wrap_periodic(cx0 + q, a.Ncc);
                const long long g = (long long)sy * a.Ncc + sx;
                v[u][0] = gA[g]; v[u][1] = gH[g]; v[u][2] = gV[g]; v[u][3] = gD[g];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int idx = base + u * NT + tid;
                idx = idx < total ? idx : total - 1;
                sA[idx] = v[u][0]; sH[idx] = v[u][1]; sV[idx] = v[u][2]; sD[idx] = v[u][3];
            }
        }
    }
    PDWT_SYNC();

    // ---- phase 2: column synthesis -> t1, t2 (OY x CXp)
    PDWT_FOR_THREADS(tid, NT) {
        const int total = OY * CXp;
        for (int idx = tid; idx < total; idx += NT) {
            const int gy = idx / CXp;
            const int q = idx - gy * CXp;
            const int p = gy + s;  // tile origin 2*by*TY is even: parity of p is local
            const int r0 = p >> 1;
            const int par = 1 - (p & 1);
            real_t r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int j = 0; j < (HLEN > 0 ? HLEN / 2 : h2); ++j) {
                const int t = hlen - 1 - (2 * j + par);
                if (HLEN == 0 && t < 0) continue;
                const int o = (r0 + j) * CXp + q;
                const real_t tl = lo[t], th = hi[t];
                r1 = pdwt_fma(sA[o], tl, r1);
                r1 = pdwt_fma(sH[o], th, r1);
                r2 = pdwt_fma(sV[o], tl, r2);
                r2 = pdwt_fma(sD[o], th, r2);
            }
            t1[idx] = r1;
            t2[idx] = r2;
        }
    }
    PDWT_SYNC();

    // ---- phase 3: row synthesis -> image tile, two adjacent samples per thread
    PDWT_FOR_THREADS(tid, NT) {
        real_t* PDWT_RESTRICT out = a.out + (long long)bz * a.out_bstride;
        const int total = OY * TX;
        const bool vec_ok = ((a.Nc & 1) == 0);
        for (int idx = tid; idx < total; idx += NT) {
            const int gy = idx / TX;
            const int k = idx - gy * TX;
            const int oy = 2 * by * TY + gy;
            const int ox = 2 * (bx * TX + k);
            const real_t* u1 = t1 + gy * CXp;
            const real_t* u2 = t2 + gy * CXp;
            real_t res[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = 2 * k + e + s;
                const int q0 = p >> 1;
                const int par = 1 - (p & 1);
                real_t r = 0.f;
#pragma unroll
                for (int j = 0; j < (HLEN > 0 ? HLEN / 2 : h2); ++j) {
                    const int t = hlen - 1 - (2 * j + par);
                    if (HLEN == 0 && t < 0) continue;
                    r = pdwt_fma(u1[q0 + j], lo[t], r);
                    r = pdwt_fma(u2[q0 + j], hi[t], r);
                }
                res[e] = r;
            }
            if (oy < a.Nr) {
                real_t* dst = out + (long long)oy * a.Nc + ox;
                if (vec_ok && ox + 1 < a.Nc) {
                    real2_t v;
                    v.x = res[0];
                    v.y = res[1];
                    *reinterpret_cast<real2_t*>(dst) = v;
                } else {
                    if (ox < a.Nc) dst[0] = res[0];
                    if (ox + 1 < a.Nc) dst[1] = res[1];
                }
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_kernel(const Fwd2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    dwt2_fwd_tile<HLEN, TX, TY, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z, pdwt_smem);
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_kernel(const Inv2DArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    dwt2_inv_tile<HLEN, TX, TY, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z, pdwt_smem);
}
#endif

}  // namespace pdwt
