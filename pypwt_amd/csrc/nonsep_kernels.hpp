// nonsep_kernels.hpp -- non-separable 2D wavelet level kernels (hlen x hlen filter banks).
//
// Semantics of pdwt/src/nonseparable.cu:114-225 (DWT) and :304-401 (SWT), restated: one thread
// per output sample, the four hlen*hlen banks (A,H,V,D <- filter1..4, the reference's LL,LH,HL,HH
// order) staged in LDS once per workgroup, input read from global (the hlen^2 * 4 MACs per sample
// make this path compute-bound for anything longer than Haar: it exists for user-supplied
// non-separable banks, `set_wavelets_filters` with do_separable = 0; built-in wavelets give the
// same result as the separable kernels at a fraction of the cost).
//
//   forward   out_b[y,x] = sum_{jy,jx} in[sy(jy), sx(jx)] * F_b[(hlen-1-jy)*hlen + (hlen-1-jx)]
//             DWT: s(j) = per(2*o - c + j) (odd sizes: last sample repeated) ; SWT: (o + (j-c) f) mod N
//   inverse   DWT: polyphase, taps (hlen-1-(2j+par)) along each axis; SWT: all taps, scaled by 1/4
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

struct NonsepArgs {
    const real_t* in;       // forward: image plane
    real_t *A, *H, *V, *D;  // forward: outputs ; inverse: inputs
    real_t* out;            // inverse: output plane
    const real_t* filt;     // device memory: 4 banks of hlen*hlen (already the right direction)
    int Nr, Nc;            // image dims of this level
    int Nrc, Ncc;          // coefficient dims of this level (== Nr,Nc for SWT)
    int f;                 // SWT dilation (1 for DWT)
    int do_swt;
    long long img_bstride, coef_bstride;
    int hlen;
};

constexpr int nonsep_lds_floats(int hlen) { return 4 * hlen * hlen; }

template <int NT>
PDWT_DEVICE void nonsep_fwd_tile(const NonsepArgs& a, long long block, int bz, real_t* smem) {
    const int hlen = a.hlen, n2 = hlen * hlen;
    PDWT_FOR_THREADS(tid, NT) {
        for (int i = tid; i < 4 * n2; i += NT) smem[i] = a.filt[i];
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const long long idx = block * NT + tid;
        const long long total = (long long)a.Nrc * a.Ncc;
        if (idx < total) {
            const int y = (int)(idx / a.Ncc), x = (int)(idx - (long long)y * a.Ncc);
            const int c = analysis_centre(hlen);
            const real_t* in = a.in + (long long)bz * a.img_bstride;
            real_t rA = 0.f, rH = 0.f, rV = 0.f, rD = 0.f;
            for (int jy = 0; jy < hlen; ++jy) {
                const int sy = a.do_swt ? wrap_periodic(y + (jy - c) * a.f, a.Nr) : wrap_analysis(2 * y - c + jy, a.Nr);
                const real_t* row = in + (long long)sy * a.Nc;
                for (int jx = 0; jx < hlen; ++jx) {
                    const int sx = a.do_swt ? wrap_periodic(x + (jx - c) * a.f, a.Nc) : wrap_analysis(2 * x - c + jx, a.Nc);
                    const real_t v = row[sx];
                    const int t = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                    rA = pdwt_fma(v, smem[t], rA);
                    rH = pdwt_fma(v, smem[n2 + t], rH);
                    rV = pdwt_fma(v, smem[2 * n2 + t], rV);
                    rD = pdwt_fma(v, smem[3 * n2 + t], rD);
                }
            }
            const long long o = (long long)bz * a.coef_bstride + idx;
            a.A[o] = rA;
            a.H[o] = rH;
            a.V[o] = rV;
            a.D[o] = rD;
        }
    }
}

template <int NT>
PDWT_DEVICE void nonsep_inv_tile(const NonsepArgs& a, long long block, int bz, real_t* smem) {
    const int hlen = a.hlen, n2 = hlen * hlen;
    PDWT_FOR_THREADS(tid, NT) {
        for (int i = tid; i < 4 * n2; i += NT) smem[i] = a.filt[i];
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const long long idx = block * NT + tid;
        const long long total = (long long)a.Nr * a.Nc;
        if (idx < total) {
            const int gy = (int)(idx / a.Nc), gx = (int)(idx - (long long)gy * a.Nc);
            const long long cb = (long long)bz * a.coef_bstride;
            real_t r = 0.f;
            if (!a.do_swt) {
                const int h2 = hlen / 2, c = h2 / 2, s = (h2 & 1) ? 0 : 1;
                const int py = gy + s, px = gx + s;
                const int by0 = (py >> 1) - c, bx0 = (px >> 1) - c;
                const int pary = 1 - (py & 1), parx = 1 - (px & 1);
                for (int jy = 0; jy < h2; ++jy) {
                    const int ty = hlen - 1 - (2 * jy + pary);
                    if (ty < 0) continue;
                    const long long ro = cb + (long long)wrap_periodic(by0 + jy, a.Nrc) * a.Ncc;
                    for (int jx = 0; jx < h2; ++jx) {
                        const int tx = hlen - 1 - (2 * jx + parx);
                        if (tx < 0) continue;
                        const long long o = ro + wrap_periodic(bx0 + jx, a.Ncc);
                        const int t = ty * hlen + tx;
                        r = pdwt_fma(a.A[o], smem[t], r);
                        r = pdwt_fma(a.H[o], smem[n2 + t], r);
                        r = pdwt_fma(a.V[o], smem[2 * n2 + t], r);
                        r = pdwt_fma(a.D[o], smem[3 * n2 + t], r);
                    }
                }
            } else {
                const int c = hlen / 2;
                for (int jy = 0; jy < hlen; ++jy) {
                    const long long ro = cb + (long long)wrap_periodic(gy + (jy - c) * a.f, a.Nr) * a.Nc;
                    for (int jx = 0; jx < hlen; ++jx) {
                        const long long o = ro + wrap_periodic(gx + (jx - c) * a.f, a.Nc);
                        const int t = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                        r = pdwt_fma(a.A[o], smem[t], r);
                        r = pdwt_fma(a.H[o], smem[n2 + t], r);
                        r = pdwt_fma(a.V[o], smem[2 * n2 + t], r);
                        r = pdwt_fma(a.D[o], smem[3 * n2 + t], r);
                    }
                }
                r *= 0.25f;  // nonseparable.cu:393-396
            }
            a.out[(long long)bz * a.img_bstride + idx] = r;
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int NT>
__global__ void __launch_bounds__(NT) nonsep_fwd_kernel(const NonsepArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    nonsep_fwd_tile<NT>(a, blockIdx.x, blockIdx.y, pdwt_smem);
}
template <int NT>
__global__ void __launch_bounds__(NT) nonsep_inv_kernel(const NonsepArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    nonsep_inv_tile<NT>(a, blockIdx.x, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
