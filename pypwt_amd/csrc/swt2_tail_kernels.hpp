// swt2_tail_kernels.hpp -- the WHOLE undecimated 2D transform of a tiny image in one launch, one workgroup per image
// (gfx950): the batch regime of dwt2_tail_kernels.hpp for the SWT.
//
// Why: a batch of thousands of tiny images (patches of 16 x 16 ... 64 x 64) through the level kernels is a launch per
// level whose tiles are mostly padding, and the inverse reads four bands x hlen taps per output straight from global memory:
// 4096 images of 64^2, db4, two levels: 1006 us forward+inverse for 2 x 134 MB of compulsory traffic; 16384 images of 32^2,
// haar L3: 1177 us.  Here a workgroup keeps its image in LDS and walks through all levels: the forward reads the image once
// and writes every band once (4 (3 L + 2) B per sample instead of 20 L); the inverse stages the three detail planes of a
// level in LDS (one coalesced read each, the pending soft threshold applied on the way) and takes its taps from there.
// Any size (round 4 first took powers of two only): the periodic a-trous offsets ((j - c) 2^(l-1)) mod n of a level are tabulated
// in LDS once per level, an index is then one add and one conditional subtract.
// Reference: w_kern_forward_swt_pass1/2, w_kern_inverse_swt_pass1/2 (pdwt/src/separable.cu:409-493, 553-626), the level
// loops w_forward_swt_separable / w_inverse_swt_separable (:496-537, :629-672); conventions as in swt_kernels.hpp
// (analysis centre hlen/2 - 1, hlen/2 for odd lengths; synthesis centre hlen/2; each synthesis pass scales by 1/2).
// Written over real_t with run-time filter length: the launch is latency- and LDS-bound by construction.
// CPU emulation (tests/cpu_emu): PDWT_FOR_THREADS / PDWT_SYNC as in the other LDS kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

constexpr int kSwtTailMaxLevels = 12;
constexpr int kSwtTailMaxSamples = 4096;   // one image; the inverse keeps four planes of it in LDS
constexpr int kSwtTailTrips = 16;          // values a thread stages per plane (256 threads x 16 = 4096)
constexpr int kSwtTailWaveSamples = 256;  // largest image the one-wavefront launch (64 threads x 4 trips) takes

struct SwtTailArgs {
    const real_t* in;                      // forward: the image; inverse: A_L
    real_t* out;                           // forward: A_L; inverse: the image
    real_t* det[kSwtTailMaxLevels][3];     // det[l-1] = (H, V, D) of level l (n samples per image each)
    real_t beta[kSwtTailMaxLevels];        // inverse: soft threshold applied to level l's details as they are staged (0 = none)
    int R, C;                              // image of R x C samples (R C <= kSwtTailMaxSamples)
    int lgR, lgC;                          // their base-2 logarithms when R and C are powers of two (the _p2 kernels), else -1
    int L;                                 // levels
    int hlen;
    FilterBank fb;                         // forward: (dec_lo, dec_hi); inverse: (rec_lo, rec_hi)
};

constexpr size_t swt_tail_lds_elems(int n, bool inverse) { return (size_t)(inverse ? 4 : 3) * n + 4 * kMaxTaps; }  // planes, taps, offsets

// offsets of a level's taps along an axis of length n, reduced into [0, n): index = x + off[j], minus n when it reaches n
template <int NT>
PDWT_DEVICE void swt_tail_offsets(int tid, int* offX, int* offY, int hlen, int c, int d, int C, int R) {
    for (int j = tid; j < hlen; j += NT) {
        offX[j] = true_mod((j - c) * d, C);
        offY[j] = true_mod((j - c) * d, R);
    }
}
PDWT_DEVICE int swt_tail_at(int x, int off, int n) {
    const int q = x + off;
    return q >= n ? q - n : q;
}
// position of tap j of sample x on an axis of length n: powers of two wrap with a mask (16384 x 32^2 haar L3 forward+inverse 352 us
// against 465 through the table), other sizes take the tabulated offset
template <bool POW2>
PDWT_DEVICE int swt_tail_tap(int x, int j, int c, int d, const int* off, int n) {
    return POW2 ? ((x + (j - c) * d) & (n - 1)) : swt_tail_at(x, off[j], n);
}

// global plane -> LDS plane, all of a thread's loads in flight together (a compile-time trip count TR with NT x TR >= the image's
// samples, clamped index: the launcher picks 16 or 4 trips.  Skipping unneeded trips by a uniform run-time test instead made the
// 256-thread inverse 15-25 % slower: 16384 x 32^2 haar L3 inverse 208 -> 257 us.)
template <int NT, int kTrips>
PDWT_DEVICE void swt_tail_stage(int tid, const real_t* PDWT_RESTRICT src, real_t* dst, int n, real_t beta) {
    static_assert(kTrips >= 1 && kTrips <= kSwtTailTrips, "trips");
    real_t v[kTrips];
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
        const int idx = tid + t * NT;
        v[t] = src[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
        const int idx = tid + t * NT;
        if (idx < n) dst[idx] = soft_shrink(v[t], beta);
    }
}

template <int NT, bool POW2, int TR = kSwtTailTrips>
PDWT_DEVICE void swt2_fwd_tail_image(const SwtTailArgs& a, int bz, real_t* smem) {
    const int hlen = a.hlen, c = analysis_centre(hlen);
    const int C = a.C, R = a.R, n = R * C;
    real_t* cur = smem;
    real_t* tL = smem + n;
    real_t* tH = smem + 2 * n;
    real_t* fLo = smem + 3 * n;
    real_t* fHi = fLo + kMaxTaps;
    int* offX = reinterpret_cast<int*>(fHi + kMaxTaps);
    int* offY = offX + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        swt_tail_stage<NT, TR>(tid, a.in + (long long)bz * n, cur, n, (real_t)0);
        for (int j = tid; j < hlen; j += NT) {  // reversed: tap j of the window multiplies f[hlen - 1 - j]
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
        if (!POW2) swt_tail_offsets<NT>(tid, offX, offY, hlen, c, 1, C, R);
    }
    PDWT_SYNC();
    for (int l = 1; l <= a.L; ++l) {
        const int d = 1 << (l - 1);  // dilation
        PDWT_FOR_THREADS(tid, NT) {  // dilated row analysis: cur -> tL, tH
            for (int idx = tid; idx < n; idx += NT) {
                const int y = POW2 ? (idx >> a.lgC) : (idx / C), x = idx - y * C;
                const real_t* row = cur + (idx - x);
                real_t aL = 0, aH = 0;
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = row[swt_tail_tap<POW2>(x, j, c, d, offX, C)];
                    aL = pdwt_fma(v, fLo[j], aL);
                    aH = pdwt_fma(v, fHi[j], aH);
                }
                tL[idx] = aL;
                tH[idx] = aH;
            }
        }
        PDWT_SYNC();
        const bool last = l == a.L;
        PDWT_FOR_THREADS(tid, NT) {  // dilated column analysis: A -> cur (the last level: global), H, V, D -> global
            const long long b = (long long)bz * n;
            real_t* PDWT_RESTRICT gA = a.out + b;
            real_t* PDWT_RESTRICT gH = a.det[l - 1][0] + b;
            real_t* PDWT_RESTRICT gV = a.det[l - 1][1] + b;
            real_t* PDWT_RESTRICT gD = a.det[l - 1][2] + b;
            for (int idx = tid; idx < n; idx += NT) {
                const int y = POW2 ? (idx >> a.lgC) : (idx / C), x = idx - y * C;
                real_t rA = 0, rH = 0, rV = 0, rD = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = swt_tail_tap<POW2>(y, j, c, d, offY, R) * C + x;
                    const real_t lv = tL[o], hv = tH[o];
                    rA = pdwt_fma(lv, fLo[j], rA);
                    rH = pdwt_fma(lv, fHi[j], rH);
                    rV = pdwt_fma(hv, fLo[j], rV);
                    rD = pdwt_fma(hv, fHi[j], rD);
                }
                if (last) gA[idx] = rA;
                else cur[idx] = rA;
                gH[idx] = rH;
                gV[idx] = rV;
                gD[idx] = rD;
            }
        }
        PDWT_SYNC();  // the row pass of the next level reads cur; its offsets are written after every reader of this level's is done
        if (!POW2 && l < a.L) {
            PDWT_FOR_THREADS(tid, NT) { swt_tail_offsets<NT>(tid, offX, offY, hlen, c, 1 << l, C, R); }
            PDWT_SYNC();
        }
    }
}

// Inverse of one level out of four LDS planes X (A, then the result), P, U1, U2:
//   V -> P;  U1 = 1/2 (Lx(X) + Hx(P));  H -> P, D -> X;  U2 = 1/2 (Lx(P) + Hx(X));  X = 1/2 (Ly(U1) + Hy(U2))
template <int NT, bool POW2, int TR = kSwtTailTrips>
PDWT_DEVICE void swt2_inv_tail_image(const SwtTailArgs& a, int bz, real_t* smem) {
    const int hlen = a.hlen, c = hlen / 2;  // synthesis centre
    const int C = a.C, R = a.R, n = R * C;
    const real_t half = (real_t)0.5;
    real_t* X = smem;
    real_t* P = smem + n;
    real_t* U1 = smem + 2 * n;
    real_t* U2 = smem + 3 * n;
    real_t* fLo = smem + 4 * n;
    real_t* fHi = fLo + kMaxTaps;
    int* offX = reinterpret_cast<int*>(fHi + kMaxTaps);
    int* offY = offX + kMaxTaps;
    const long long b = (long long)bz * n;
    PDWT_FOR_THREADS(tid, NT) {
        swt_tail_stage<NT, TR>(tid, a.in + b, X, n, (real_t)0);
        swt_tail_stage<NT, TR>(tid, a.det[a.L - 1][1] + b, P, n, a.beta[a.L - 1]);  // V of the coarsest level
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
        if (!POW2) swt_tail_offsets<NT>(tid, offX, offY, hlen, c, 1 << (a.L - 1), C, R);
    }
    PDWT_SYNC();
    for (int l = a.L; l >= 1; --l) {
        const int d = 1 << (l - 1);  // dilation
        PDWT_FOR_THREADS(tid, NT) {  // U1 = 1/2 (Lx(A) + Hx(V))
            for (int idx = tid; idx < n; idx += NT) {
                const int y = POW2 ? (idx >> a.lgC) : (idx / C), x = idx - y * C, ro = idx - x;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = ro + swt_tail_tap<POW2>(x, j, c, d, offX, C);
                    r = pdwt_fma(X[o], fLo[j], r);
                    r = pdwt_fma(P[o], fHi[j], r);
                }
                U1[idx] = half * r;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // A and V are consumed: H -> P, D -> X
            swt_tail_stage<NT, TR>(tid, a.det[l - 1][0] + b, P, n, a.beta[l - 1]);
            swt_tail_stage<NT, TR>(tid, a.det[l - 1][2] + b, X, n, a.beta[l - 1]);
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // U2 = 1/2 (Lx(H) + Hx(D))
            for (int idx = tid; idx < n; idx += NT) {
                const int y = POW2 ? (idx >> a.lgC) : (idx / C), x = idx - y * C, ro = idx - x;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = ro + swt_tail_tap<POW2>(x, j, c, d, offX, C);
                    r = pdwt_fma(P[o], fLo[j], r);
                    r = pdwt_fma(X[o], fHi[j], r);
                }
                U2[idx] = half * r;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // 1/2 (Ly(U1) + Hy(U2)) -> X (the finest level: global); then V of the next level -> P
            real_t* PDWT_RESTRICT gout = a.out + b;
            for (int idx = tid; idx < n; idx += NT) {
                const int y = POW2 ? (idx >> a.lgC) : (idx / C), x = idx - y * C;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = swt_tail_tap<POW2>(y, j, c, d, offY, R) * C + x;
                    r = pdwt_fma(U1[o], fLo[j], r);
                    r = pdwt_fma(U2[o], fHi[j], r);
                }
                if (l == 1) gout[idx] = half * r;
                else X[idx] = half * r;
            }
            if (l > 1) swt_tail_stage<NT, TR>(tid, a.det[l - 2][1] + b, P, n, a.beta[l - 2]);  // P (H) was consumed before the barrier
        }
        PDWT_SYNC();
        if (!POW2 && l > 1) {  // the offsets of the next (finer) level, after every reader of this level's is done
            PDWT_FOR_THREADS(tid, NT) { swt_tail_offsets<NT>(tid, offX, offY, hlen, c, 1 << (l - 2), C, R); }
            PDWT_SYNC();
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int NT, bool POW2, int TR>
__global__ void __launch_bounds__(NT) swt2_fwd_tail_kernel(const SwtTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char swt_tail_smem[];
    swt2_fwd_tail_image<NT, POW2, TR>(a, blockIdx.x, reinterpret_cast<real_t*>(swt_tail_smem));
}
template <int NT, bool POW2, int TR>
__global__ void __launch_bounds__(NT) swt2_inv_tail_kernel(const SwtTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char swt_tail_smem[];
    swt2_inv_tail_image<NT, POW2, TR>(a, blockIdx.x, reinterpret_cast<real_t*>(swt_tail_smem));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// The power-of-two kernels as first written (mask / shift indexing), kept as their own functions: see dwt2_tail_kernels.hpp.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NT, int TR = kSwtTailTrips>
PDWT_DEVICE void swt2_fwd_tail_image_p2(const SwtTailArgs& a, int bz, real_t* smem) {
    const int hlen = a.hlen, c = analysis_centre(hlen);
    const int lgC = a.lgC, C = 1 << lgC, R = 1 << a.lgR, n = 1 << (a.lgR + a.lgC);
    real_t* cur = smem;
    real_t* tL = smem + n;
    real_t* tH = smem + 2 * n;
    real_t* fLo = smem + 3 * n;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        swt_tail_stage<NT, TR>(tid, a.in + (long long)bz * n, cur, n, (real_t)0);
        for (int j = tid; j < hlen; j += NT) {  // reversed: tap j of the window multiplies f[hlen - 1 - j]
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    for (int l = 1; l <= a.L; ++l) {
        const int d = 1 << (l - 1);  // dilation
        PDWT_FOR_THREADS(tid, NT) {  // dilated row analysis: cur -> tL, tH
            for (int idx = tid; idx < n; idx += NT) {
                const int x = idx & (C - 1);
                const real_t* row = cur + (idx - x);
                real_t aL = 0, aH = 0;
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = row[(x + (j - c) * d) & (C - 1)];
                    aL = pdwt_fma(v, fLo[j], aL);
                    aH = pdwt_fma(v, fHi[j], aH);
                }
                tL[idx] = aL;
                tH[idx] = aH;
            }
        }
        PDWT_SYNC();
        const bool last = l == a.L;
        PDWT_FOR_THREADS(tid, NT) {  // dilated column analysis: A -> cur (the last level: global), H, V, D -> global
            const long long b = (long long)bz * n;
            real_t* PDWT_RESTRICT gA = a.out + b;
            real_t* PDWT_RESTRICT gH = a.det[l - 1][0] + b;
            real_t* PDWT_RESTRICT gV = a.det[l - 1][1] + b;
            real_t* PDWT_RESTRICT gD = a.det[l - 1][2] + b;
            for (int idx = tid; idx < n; idx += NT) {
                const int x = idx & (C - 1), y = idx >> lgC;
                real_t rA = 0, rH = 0, rV = 0, rD = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = (((y + (j - c) * d) & (R - 1)) << lgC) + x;
                    const real_t lv = tL[o], hv = tH[o];
                    rA = pdwt_fma(lv, fLo[j], rA);
                    rH = pdwt_fma(lv, fHi[j], rH);
                    rV = pdwt_fma(hv, fLo[j], rV);
                    rD = pdwt_fma(hv, fHi[j], rD);
                }
                if (last) gA[idx] = rA;
                else cur[idx] = rA;
                gH[idx] = rH;
                gV[idx] = rV;
                gD[idx] = rD;
            }
        }
        PDWT_SYNC();
    }
}

// Inverse of one level out of four LDS planes X (A, then the result), P, U1, U2:
//   V -> P;  U1 = 1/2 (Lx(X) + Hx(P));  H -> P, D -> X;  U2 = 1/2 (Lx(P) + Hx(X));  X = 1/2 (Ly(U1) + Hy(U2))
template <int NT, int TR = kSwtTailTrips>
PDWT_DEVICE void swt2_inv_tail_image_p2(const SwtTailArgs& a, int bz, real_t* smem) {
    const int hlen = a.hlen, c = hlen / 2;  // synthesis centre
    const int lgC = a.lgC, C = 1 << lgC, R = 1 << a.lgR, n = 1 << (a.lgR + a.lgC);
    const real_t half = (real_t)0.5;
    real_t* X = smem;
    real_t* P = smem + n;
    real_t* U1 = smem + 2 * n;
    real_t* U2 = smem + 3 * n;
    real_t* fLo = smem + 4 * n;
    real_t* fHi = fLo + kMaxTaps;
    const long long b = (long long)bz * n;
    PDWT_FOR_THREADS(tid, NT) {
        swt_tail_stage<NT, TR>(tid, a.in + b, X, n, (real_t)0);
        swt_tail_stage<NT, TR>(tid, a.det[a.L - 1][1] + b, P, n, a.beta[a.L - 1]);  // V of the coarsest level
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    for (int l = a.L; l >= 1; --l) {
        const int d = 1 << (l - 1);  // dilation
        PDWT_FOR_THREADS(tid, NT) {  // U1 = 1/2 (Lx(A) + Hx(V))
            for (int idx = tid; idx < n; idx += NT) {
                const int x = idx & (C - 1), ro = idx - x;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = ro + ((x + (j - c) * d) & (C - 1));
                    r = pdwt_fma(X[o], fLo[j], r);
                    r = pdwt_fma(P[o], fHi[j], r);
                }
                U1[idx] = half * r;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // A and V are consumed: H -> P, D -> X
            swt_tail_stage<NT, TR>(tid, a.det[l - 1][0] + b, P, n, a.beta[l - 1]);
            swt_tail_stage<NT, TR>(tid, a.det[l - 1][2] + b, X, n, a.beta[l - 1]);
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // U2 = 1/2 (Lx(H) + Hx(D))
            for (int idx = tid; idx < n; idx += NT) {
                const int x = idx & (C - 1), ro = idx - x;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = ro + ((x + (j - c) * d) & (C - 1));
                    r = pdwt_fma(P[o], fLo[j], r);
                    r = pdwt_fma(X[o], fHi[j], r);
                }
                U2[idx] = half * r;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // 1/2 (Ly(U1) + Hy(U2)) -> X (the finest level: global); then V of the next level -> P
            real_t* PDWT_RESTRICT gout = a.out + b;
            for (int idx = tid; idx < n; idx += NT) {
                const int x = idx & (C - 1), y = idx >> lgC;
                real_t r = 0;
                for (int j = 0; j < hlen; ++j) {
                    const int o = (((y + (j - c) * d) & (R - 1)) << lgC) + x;
                    r = pdwt_fma(U1[o], fLo[j], r);
                    r = pdwt_fma(U2[o], fHi[j], r);
                }
                if (l == 1) gout[idx] = half * r;
                else X[idx] = half * r;
            }
            if (l > 1) swt_tail_stage<NT, TR>(tid, a.det[l - 2][1] + b, P, n, a.beta[l - 2]);  // P (H) was consumed before the barrier
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int NT, int TR>
__global__ void __launch_bounds__(NT) swt2_fwd_tail_p2_kernel(const SwtTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char swt_tail_smem[];
    swt2_fwd_tail_image_p2<NT, TR>(a, blockIdx.x, reinterpret_cast<real_t*>(swt_tail_smem));
}
template <int NT, int TR>
__global__ void __launch_bounds__(NT) swt2_inv_tail_p2_kernel(const SwtTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char swt_tail_smem[];
    swt2_inv_tail_image_p2<NT, TR>(a, blockIdx.x, reinterpret_cast<real_t*>(swt_tail_smem));
}
#endif

}  // namespace pdwt
