// dwt1_rows_kernels.hpp -- ALL levels of a batched 1D DWT of SHORT rows, several rows per wavefront (gfx950).
//
// Why: a batched 1D plan over rows of 64-256 samples (the reference's ndim = 1 on a 2D array, pdwt/src/separable.cu:214-236, :368-395)
// ran one row per wavefront (dwt1_*_fused_rows_kernel): 512 bytes and ~4.4 us of latency per wavefront whatever the grouping -- 65 536
// rows of 64 samples took 35 + 45 us for 16.8 MB each way.  Here a workgroup of ONE wavefront owns G consecutive rows (G x N0 <= 1024
// samples: a contiguous block of every band), stages them with all loads in flight and walks the levels out of two LDS buffers: a work
// item is one output pair of any of the G rows, so the lanes stay busy down to the coarsest level.  The 1D twin of the tail launch of
// dwt2_tail_kernels.hpp (same staging scheme, same taps-in-registers, same pair synthesis).
// Row lengths are multiples of 2^(K+2) (dwt1_fused_supported): every level halves exactly.
// Index conventions as in oracle/pdwt_oracle.c (analysis centre hlen/2 - 1; polyphase synthesis with h2 = hlen/2, c = h2/2, s = 1 - (h2 & 1)).
#pragma once

#include "dwt2_tail_kernels.hpp"  // TailAxis, TailTaps, kTailTrips
#include "kernels_common.hpp"

namespace pdwt {

constexpr int kRowsTailMaxLevels = 8;
constexpr int kRowsTailSamples = 1024;  // samples of the G rows of a workgroup (64 threads x kTailTrips)

struct RowsTailArgs {
    const real_t* in;                  // forward: (rows, N0); inverse: A_K (rows, N0 >> K)
    real_t* out;                       // forward: A_K; inverse: (rows, N0)
    real_t* det[kRowsTailMaxLevels];   // det[k] = D_{k+1}, (rows, N0 >> (k + 1))
    int rows, N0, K, G, hlen;
    FilterBank fb;                     // forward: (dec_lo, dec_hi); inverse: (rec_lo, rec_hi)
};

// LDS: two buffers of G N0 samples, the details of all levels (inverse: < G N0), the taps
constexpr size_t rows_tail_lds_elems(int gn0) { return (size_t)3 * gn0 + 2 * kMaxTaps + kTailPtrElems; }

template <int HLEN, int NT>
PDWT_DEVICE void dwt1_rows_tail_fwd(const RowsTailArgs& a, int block, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, C = hlen / 2 - 1;
    const int g0 = block * a.G, gn = a.rows - g0 < a.G ? a.rows - g0 : a.G;
    const int n0 = gn * a.N0;
    real_t* cur = smem;
    real_t* nxt = smem + a.G * a.N0;
    real_t* fLo = smem + 3 * a.G * a.N0;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        const real_t* PDWT_RESTRICT in = a.in + (long long)g0 * a.N0;
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int idx = tid + t * NT;
            if (t * NT < n0) v[t] = in[idx < n0 ? idx : n0 - 1];
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int idx = tid + t * NT;
            if (idx < n0) cur[idx] = v[t];
        }
        for (int j = tid; j < hlen; j += NT) {  // reversed: tap j of the window multiplies f[hlen - 1 - j]
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    for (int k = 0; k < a.K; ++k) {
        const int c = a.N0 >> k, c2 = c >> 1;
        const TailAxis ac = tail_axis(c);
        const bool last = k == a.K - 1;
        PDWT_FOR_THREADS(tid, NT) {
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            real_t* PDWT_RESTRICT gD = a.det[k] + (long long)g0 * c2;
            real_t* PDWT_RESTRICT gA = a.out + (long long)g0 * c2;
            for (int idx = tid; idx < gn * c2; idx += NT) {
                const int y = idx / c2, x = idx - y * c2;
                const real_t* row = cur + y * c;
                const int base = 2 * x - C;
                real_t l = 0, h = 0;
#pragma unroll
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = row[ac.wrap(base + j)];
                    l = pdwt_fma(v, f.l(j), l);
                    h = pdwt_fma(v, f.h(j), h);
                }
                if (last) gA[idx] = l;
                else nxt[idx] = l;
                gD[idx] = h;
            }
        }
        PDWT_SYNC();
        real_t* t = cur; cur = nxt; nxt = t;
    }
}

template <int HLEN, int NT>
PDWT_DEVICE void dwt1_rows_tail_inv(const RowsTailArgs& a, int block, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, H2 = hlen / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    const int g0 = block * a.G, gn = a.rows - g0 < a.G ? a.rows - g0 : a.G;
    const int cK = a.N0 >> a.K;
    const int sL = gn * cK, total = gn * a.N0;  // A_K and the details of all levels: exactly the samples of the G rows
    real_t* X = smem;
    real_t* Y = smem + a.G * a.N0;
    real_t* Dl = smem + 2 * a.G * a.N0;  // details, coarsest level first: level k at Dl + gn ((N0 >> (k + 1)) - cK)
    real_t* fLo = smem + 3 * a.G * a.N0;
    real_t* fHi = fLo + kMaxTaps;
    real_t** ptab = tail_ptr_table(smem, 3 * a.G * a.N0 + 2 * kMaxTaps);  // the bands' addresses (see dwt2_inv_tail_image)
    PDWT_FOR_THREADS(tid, NT) {
        if (tid == 0)
            for (int k = 0; k < a.K; ++k) ptab[k] = a.det[k];
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        // flat order: [0, sL) = A_K; then level K-1, K-2, ... 0 (sizes sL, 2 sL, 4 sL, ...): position f >= sL lies in the level whose
        // block starts at the largest sL 2^m <= f -- its details go to Dl + (f - sL)
        const real_t* src[kTailTrips];
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            if (t * NT >= total) continue;
            int f = tid + t * NT;
            f = f < total ? f : total - 1;
            if (f < sL) {
                src[t] = a.in + (long long)g0 * cK + f;
            } else {
                int m = 0, pm = sL;  // block [pm, 2 pm) = level K-1-m, gn (cK << m) values
                while (f >= 2 * pm) { pm *= 2; ++m; }
                src[t] = ptab[a.K - 1 - m] + (long long)g0 * (cK << m) + (f - pm);
            }
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t)
            if (t * NT < total) v[t] = *src[t];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int f = tid + t * NT;
            if (f < total) {
                if (f < sL) X[f] = v[t];
                else Dl[f - sL] = v[t];
            }
        }
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    // pairs of outputs per work item (see dwt2_inv_tail_image): out[2m] = sum_j s[j] f[2j + 1 - S], out[2m + 1] = sum_j s[j + S] f[2j + S]
    for (int k = a.K - 1; k >= 0; --k) {
        const int ci = a.N0 >> (k + 1), co = 2 * ci;
        const TailAxis aci = tail_axis(ci);
        const real_t* dk = Dl + gn * (ci - cK);
        PDWT_FOR_THREADS(tid, NT) {
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            real_t* PDWT_RESTRICT gout = a.out + (long long)g0 * co;
            for (int idx = tid; idx < gn * ci; idx += NT) {
                const int y = idx / ci, m = idx - y * ci;
                real_t e = 0, od = 0;
#pragma unroll
                for (int i = 0; i < H2 + S; ++i) {
                    const int src = y * ci + aci.wrap(m - C + i);
                    const real_t v1 = X[src], v2 = dk[src];
                    if (i < H2) {
                        const int t = 2 * i + 1 - S;
                        e = pdwt_fma(v1, f.l(t), e);
                        e = pdwt_fma(v2, f.h(t), e);
                    }
                    if (i >= S) {
                        const int t = 2 * (i - S) + S;
                        od = pdwt_fma(v1, f.l(t), od);
                        od = pdwt_fma(v2, f.h(t), od);
                    }
                }
                real_t* dst = (k == 0 ? gout : Y) + 2 * idx;  // (y, 2m) of co = 2 ci columns
                dst[0] = e;
                dst[1] = od;
            }
        }
        PDWT_SYNC();
        real_t* t = X; X = Y; Y = t;
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt1_rows_tail_fwd_kernel(const RowsTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_tail_smem[];
    dwt1_rows_tail_fwd<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(rows_tail_smem));
}
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt1_rows_tail_inv_kernel(const RowsTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_tail_smem[];
    dwt1_rows_tail_inv<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(rows_tail_smem));
}
#endif

}  // namespace pdwt
