// dwt1_kernels.hpp -- (batched) 1D DWT level kernels for gfx950.
//
// A workgroup of NT threads produces TXO consecutive outputs of ONE row:
// the 2*TXO + hlen - 2 input samples are staged in LDS once (coalesced), each
// thread then computes TXO/NT low/high output pairs from LDS with the taps in
// SGPRs.  The grid is flattened over (row, tile) so a single 2^24-sample row
// fills the chip (the reference launches 16x16 blocks and idles 15 of 16 thread
// rows when Nr == 1, pdwt/src/separable.cu:224-226).
//
// Semantics: rows of pdwt/src/separable.cu:91-131 (analysis) and :293-328
// (synthesis), restated in oracle/pdwt_oracle.c.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

template <int TXO>
constexpr int fwd1d_lds_floats(int hlen) {
    return 2 * kMaxTaps + ((2 * TXO + hlen - 2 + 1) & ~1);
}

template <int HLEN, int TXO, int NT>
PDWT_DEVICE void dwt1_fwd_tile(const Fwd1DArgs& a, int bx, int row, real_t* smem) {
    static_assert(TXO % NT == 0, "outputs per thread must be integral");
    const int hlen = HLEN ? HLEN : a.hlen;
    const int c = analysis_centre(hlen);
    const int RX = (2 * TXO + hlen - 2 + 1) & ~1;

    real_t* sTaps = smem;
    real_t* sIn = smem + 2 * kMaxTaps;
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

    const real_t* PDWT_RESTRICT in = a.in + (long long)row * a.Nc;
    const int x0 = 2 * bx * TXO - c;

    PDWT_FOR_THREADS(tid, NT) {  // four loads per thread in flight (see the staging note in dwt2_kernels.hpp)
        for (int base = 0; base < RX; base += 4 * NT) {
            real_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int q = base + u * NT + tid;
                q = q < RX ? q : RX - 1;
                v[u] = in[wrap_analysis(x0 + q, a.Nc)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int q = base + u * NT + tid;
                q = q < RX ? q : RX - 1;
                sIn[q] = v[u];
            }
        }
    }
    PDWT_SYNC();

    PDWT_FOR_THREADS(tid, NT) {
        real_t* PDWT_RESTRICT oL = a.L + (long long)row * a.Nc2;
        real_t* PDWT_RESTRICT oH = a.H + (long long)row * a.Nc2;
#pragma unroll
        for (int i = 0; i < TXO / NT; ++i) {
            const int k = tid + i * NT;
            const real_t* p = sIn + 2 * k;
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
            const int ox = bx * TXO + k;
            if (ox < a.Nc2) {
                oL[ox] = aL;
                oH[ox] = aH;
            }
        }
    }
}

template <int TXO>
constexpr int inv1d_lds_floats(int hlen) {
    return 2 * kMaxTaps + 2 * (TXO + hlen / 2 + 1);
}

// produces 2*TXO consecutive samples of one row from TXO (+halo) coefficients
template <int HLEN, int TXO, int NT>
PDWT_DEVICE void dwt1_inv_tile(const Inv1DArgs& a, int bx, int row, real_t* smem) {
    static_assert(TXO % NT == 0, "outputs per thread must be integral");
    const int hlen = HLEN ? HLEN : a.hlen;
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int s = (h2 & 1) ? 0 : 1;
    const int CX = TXO + h2 + 1;

    real_t* sTaps = smem;
    real_t* sL = smem + 2 * kMaxTaps;
    real_t* sH = sL + CX;
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

    const real_t* PDWT_RESTRICT gL = a.L + (long long)row * a.Ncc;
    const real_t* PDWT_RESTRICT gH = a.H + (long long)row * a.Ncc;
    const int cx0 = bx * TXO - c;

    PDWT_FOR_THREADS(tid, NT) {  // two positions of both bands (four loads) in flight per thread
        for (int base = 0; base < CX; base += 2 * NT) {
            real_t vl[2], vh[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int q = base + u * NT + tid;
                q = q < CX ? q : CX - 1;
                const int sx = wrap_periodic(cx0 + q, a.Ncc);
                vl[u] = gL[sx];
                vh[u] = gH[sx];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int q = base + u * NT + tid;
                q = q < CX ? q : CX - 1;
                sL[q] = vl[u];
                sH[q] = vh[u];
            }
        }
    }
    PDWT_SYNC();

    PDWT_FOR_THREADS(tid, NT) {
        real_t* PDWT_RESTRICT out = a.out + (long long)row * a.Nc;
        const bool vec_ok = ((a.Nc & 1) == 0);
#pragma unroll
        for (int i = 0; i < TXO / NT; ++i) {
            const int k = tid + i * NT;
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
                    r = pdwt_fma(sL[q0 + j], lo[t], r);
                    r = pdwt_fma(sH[q0 + j], hi[t], r);
                }
                res[e] = r;
            }
            const int ox = 2 * (bx * TXO + k);
            real_t* dst = out + ox;
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

#ifndef PDWT_CPU_EMU
template <int HLEN, int TXO, int NT>
__global__ void __launch_bounds__(NT) dwt1_fwd_kernel(const Fwd1DArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    const int row = blockIdx.x / tiles_x;
    const int bx = blockIdx.x - row * tiles_x;
    dwt1_fwd_tile<HLEN, TXO, NT>(a, bx, row, pdwt_smem);
}

template <int HLEN, int TXO, int NT>
__global__ void __launch_bounds__(NT) dwt1_inv_kernel(const Inv1DArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    const int row = blockIdx.x / tiles_x;
    const int bx = blockIdx.x - row * tiles_x;
    dwt1_inv_tile<HLEN, TXO, NT>(a, bx, row, pdwt_smem);
}
#endif

}  // namespace pdwt
