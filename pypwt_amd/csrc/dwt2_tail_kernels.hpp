// dwt2_tail_kernels.hpp -- ALL remaining levels of a 2D DWT in one launch once an image's approximation fits the LDS
// of one CU (gfx950: 160 KB).
//
// Why: the reference's own benchmark and test plans ask for the MAXIMUM number of levels (test/benchmark.py:20-38 passes
// levels = 99 and lets the constructor clamp it: 2048^2 haar = 11 levels, 128^2 = 7).  Below 128 x 128 a level is a few
// thousand samples and costs exactly what a dependent launch costs (3-4 us each, one to three levels per launch with the
// pyramids): dwt2 haar 128^2 L7 forward was three launches = 12.1 us, of which the arithmetic is a fraction of a
// microsecond.  Here ONE workgroup per image keeps the approximation in LDS and walks through every remaining level:
// row pass -> (L | H) planes in a second LDS buffer -> column pass -> the next approximation back into the first buffer,
// details straight to global memory.  The inverse fetches the coarsest approximation and the details of ALL its levels in one
// batch of loads up front (together exactly R0 x C0 values: one global-memory round trip for the whole launch, not one per
// level) and then walks back up out of LDS.
// No halo, no tile geometry: the whole plane is resident.  Power-of-two sizes: the periodic wrap is a mask and the index
// split a shift (POW2 instantiations); any other size whose halves stay even through the group: conditional wrap + division.
// Reference: w_kern_forward_pass1/2 (pdwt/src/separable.cu:91-176), w_kern_inverse_pass1/2 (:246-328), the level loops
// w_forward_separable / w_inverse_separable (:179-236, :332-395); index conventions as in oracle/pdwt_oracle.c
// (analysis centre hlen/2 - 1; polyphase synthesis with h2 = hlen/2, c = h2/2, s = 1 - (h2 & 1)).
// Written over real_t; HLEN = 0 is the run-time filter length (every even length), 2-8 are unrolled.  The taps are copied
// to LDS once.  CPU emulation (tests/cpu_emu): PDWT_FOR_THREADS / PDWT_SYNC as in the other LDS kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

constexpr int kTailMaxLevels = 14;
// samples of one image entering the first level of the group: two buffers of that size + the taps must fit 160 KB
constexpr int kTailMaxSamples = sizeof(real_t) == 4 ? 16384 : 8192;
constexpr int kTailTrips = 16;  // values a thread stages: the launcher picks NT with R0 x C0 <= kTailTrips x NT

struct TailArgs {
    const real_t* in;                 // forward: A_{l-1} (R0 x C0 per image); inverse: A_L
    real_t* out;                      // forward: A_L; inverse: A_{l-1}
    real_t* det[kTailMaxLevels][3];   // det[k] = (H, V, D) of the group's k-th level, finest first: forward written, inverse read
    int R0, C0;                       // size entering the finest level; R0 >> k and C0 >> k are even for k < K
    int lgR, lgC;                     // their base-2 logarithms when both are powers of two (the POW2 instantiations)
    int K;                            // levels in the group
    int hlen;                         // even
    FilterBank fb;                    // forward: analysis (dec_lo, dec_hi); inverse: synthesis (rec_lo, rec_hi)
};

constexpr size_t tail_lds_elems(int n0) { return (size_t)2 * n0 + 2 * kMaxTaps; }

// One axis of a level: its length n, the periodic wrap of an index near [0, n) and the split of a flat index by n.  POW2: a
// mask and a shift (lg = log2 n); otherwise a conditional add / subtract (the modulo only where a tiny level is shorter than the
// filter) and a division -- sizes like 28, 48, 96, 100 (round 4: batches of such images were 3-10x slower than their power-of-two
// neighbours on the level kernels).
template <bool POW2>
struct TailAxis {
    int n, lg;
    PDWT_DEVICE int wrap(int q) const {
        if (POW2) return q & (n - 1);
        q = q < 0 ? q + n : (q >= n ? q - n : q);
        return (unsigned)q < (unsigned)n ? q : true_mod(q, n);
    }
    PDWT_DEVICE int div(int idx) const { return POW2 ? (idx >> lg) : (idx / n); }
};
template <bool POW2>
PDWT_DEVICE TailAxis<POW2> tail_axis(int n, int lg) { TailAxis<POW2> a; a.n = n; a.lg = lg; return a; }

template <int HLEN, int NT, bool POW2>
PDWT_DEVICE void dwt2_fwd_tail_image(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, C = hlen / 2 - 1;
    const int n0 = a.R0 * a.C0;
    real_t* cur = smem;
    real_t* tmp = smem + n0;
    real_t* fLo = smem + 2 * n0;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        const real_t* PDWT_RESTRICT in = a.in + (long long)bz * n0;
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {  // constant trip count, clamped index: all loads of a thread in flight together
            const int idx = tid + t * NT;
            v[t] = in[idx < n0 ? idx : n0 - 1];
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
        const int r = a.R0 >> k, c = a.C0 >> k, c2 = c >> 1, r2 = r >> 1;
        const TailAxis<POW2> ac = tail_axis<POW2>(c, a.lgC - k), ac2 = tail_axis<POW2>(c2, a.lgC - k - 1), ar = tail_axis<POW2>(r, a.lgR - k);
        const int n_half = r * c2, n_quarter = r2 * c2;
        real_t* tL = tmp;
        real_t* tH = tmp + n_half;
        PDWT_FOR_THREADS(tid, NT) {  // rows: (r x c) -> L | H, (r x c2) each
            for (int idx = tid; idx < n_half; idx += NT) {
                const int y = ac2.div(idx), x = idx - y * c2;
                const real_t* row = cur + y * c;
                const int base = 2 * x - C;
                real_t l = 0, h = 0;
#pragma unroll
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = row[ac.wrap(base + j)];
                    l = pdwt_fma(v, fLo[j], l);
                    h = pdwt_fma(v, fHi[j], h);
                }
                tL[idx] = l;
                tH[idx] = h;
            }
        }
        PDWT_SYNC();
        const bool last = k == a.K - 1;
        PDWT_FOR_THREADS(tid, NT) {  // columns: -> A (LDS, or global for the last level), H, V, D (global)
            const long long b = (long long)bz * n_quarter;
            real_t* PDWT_RESTRICT gA = a.out + b;
            real_t* PDWT_RESTRICT gH = a.det[k][0] + b;
            real_t* PDWT_RESTRICT gV = a.det[k][1] + b;
            real_t* PDWT_RESTRICT gD = a.det[k][2] + b;
            for (int idx = tid; idx < n_quarter; idx += NT) {
                const int i = ac2.div(idx), x = idx - i * c2;
                const int base = 2 * i - C;
                real_t ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
                for (int j = 0; j < hlen; ++j) {
                    const int o = ar.wrap(base + j) * c2 + x;
                    const real_t vL = tL[o], vH = tH[o];
                    ll = pdwt_fma(vL, fLo[j], ll);
                    lh = pdwt_fma(vL, fHi[j], lh);
                    hl = pdwt_fma(vH, fLo[j], hl);
                    hh = pdwt_fma(vH, fHi[j], hh);
                }
                if (last) gA[idx] = ll;
                else cur[idx] = ll;
                gH[idx] = lh;  // (row low, column high), (row high, column low), (row high, column high): separable.cu:135-176
                gV[idx] = hl;
                gD[idx] = hh;
            }
        }
        PDWT_SYNC();
    }
}

// Flat order of everything the inverse reads, from the coarse end: with sL = samples of A_L, positions [0, sL) are A_L and
// [4^m sL, 4^(m+1) sL) the three detail planes (4^m sL each) of the group's level K-1-m -- together exactly n0 values (every
// level halves both sizes exactly).  LDS homes: positions below sL and from n0/4 on (the finest level's details) live at X[f];
// the rest at U[n0/4 + f].
template <int HLEN, int NT, bool POW2>
PDWT_DEVICE void dwt2_inv_tail_image(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, H2 = hlen / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    const int n0 = a.R0 * a.C0, q0 = n0 >> 2;
    const int sL = (a.R0 >> a.K) * (a.C0 >> a.K);
    real_t* X = smem;       // [0, n0/4): the approximation being rebuilt; [n0/4, n0): H, V, D of the finest level
    real_t* U = smem + n0;  // column-synthesis results (u1 | u2); [n0/4, n0/2): the details of the coarser levels until the last step
    real_t* fLo = smem + 2 * n0;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        // three unrolled passes so that a thread's loads are in flight TOGETHER: the plane pointers (a lane-indexed read of the
        // kernel arguments: memory loads), then the values, then the LDS stores.  Interleaved, the in-order load counter makes
        // every pointer wait drain the data loads before it: sixteen round trips instead of two.
        const real_t* src[kTailTrips];
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            int f = tid + t * NT;
            f = f < n0 ? f : n0 - 1;
            if (f < sL) {
                src[t] = a.in + (long long)bz * sL + f;
            } else {
                int m = 0, pm = sL;  // plane size of level K-1-m: 4^m sL
                while (f >= 4 * pm) { pm *= 4; ++m; }
                const int g = f - pm, b = (g >= pm) + (g >= 2 * pm), idx = g - b * pm;
                src[t] = a.det[a.K - 1 - m][b] + (long long)bz * pm + idx;
            }
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) v[t] = *src[t];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int f = tid + t * NT;
            if (f < n0) {
                if (f < sL || f >= q0) X[f] = v[t];
                else U[q0 + f] = v[t];
            }
        }
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    for (int k = a.K - 1; k >= 0; --k) {
        const int ri = a.R0 >> (k + 1), ci = a.C0 >> (k + 1), co = 2 * ci;
        const TailAxis<POW2> ari = tail_axis<POW2>(ri, a.lgR - k - 1), aci = tail_axis<POW2>(ci, a.lgC - k - 1), aco = tail_axis<POW2>(co, a.lgC - k);
        const int n_in = ri * ci;  // = 4^(K-1-k) sL
        const real_t* dH = k == 0 ? X + q0 : U + q0 + n_in;
        const real_t* dV = dH + n_in;
        const real_t* dD = dV + n_in;
        real_t* u1 = U;
        real_t* u2 = U + 2 * n_in;
        PDWT_FOR_THREADS(tid, NT) {  // column synthesis: (A, H) -> u1, (V, D) -> u2, (2 ri x ci) each
            for (int idx = tid; idx < 2 * n_in; idx += NT) {
                const int q = aci.div(idx), x = idx - q * ci;
                const int p = q + S;
                const int rel = (p >> 1) - C, par = 1 - (p & 1);
                real_t r1 = 0, r2 = 0;
#pragma unroll
                for (int j = 0; j < H2; ++j) {
                    const int t = 2 * j + par;  // reversed taps: f[hlen - 1 - t]
                    const int src = ari.wrap(rel + j) * ci + x;
                    r1 = pdwt_fma(X[src], fLo[t], r1);
                    r1 = pdwt_fma(dH[src], fHi[t], r1);
                    r2 = pdwt_fma(dV[src], fLo[t], r2);
                    r2 = pdwt_fma(dD[src], fHi[t], r2);
                }
                u1[idx] = r1;
                u2[idx] = r2;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // row synthesis: (u1, u2) -> (2 ri x 2 ci), to LDS or (the last step) global
            real_t* PDWT_RESTRICT gout = a.out + (long long)bz * 4 * n_in;
            for (int idx = tid; idx < 4 * n_in; idx += NT) {
                const int q = aco.div(idx), g = idx - q * co;
                const int p = g + S;
                const int rel = (p >> 1) - C, par = 1 - (p & 1);
                real_t r = 0;
#pragma unroll
                for (int j = 0; j < H2; ++j) {
                    const int t = 2 * j + par;
                    const int src = q * ci + aci.wrap(rel + j);
                    r = pdwt_fma(u1[src], fLo[t], r);
                    r = pdwt_fma(u2[src], fHi[t], r);
                }
                if (k == 0) gout[idx] = r;
                else X[idx] = r;
            }
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT, bool POW2>
__global__ void __launch_bounds__(NT) dwt2_fwd_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_fwd_tail_image<HLEN, NT, POW2>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
}
template <int HLEN, int NT, bool POW2>
__global__ void __launch_bounds__(NT) dwt2_inv_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_inv_tail_image<HLEN, NT, POW2>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// The power-of-two kernels as first written (mask / shift indexing, the flat staging order by leading-zero count).  Kept as their
// own functions: folded into the general templates above as a POW2 flag, the SAME source ran 15-25 % slower in the inverse
// (same-box A/B, profiles/r04zq_tail_pow2_ab.txt) -- the dispatch uses these for power-of-two sizes and the general ones otherwise.
// ---------------------------------------------------------------------------------------------------------------------------
template <int HLEN, int NT>
PDWT_DEVICE void dwt2_fwd_tail_image_p2(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, C = hlen / 2 - 1;
    const int n0 = 1 << (a.lgR + a.lgC);
    real_t* cur = smem;
    real_t* tmp = smem + n0;
    real_t* fLo = smem + 2 * n0;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        const real_t* PDWT_RESTRICT in = a.in + (long long)bz * n0;
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {  // constant trip count, clamped index: all loads of a thread in flight together
            const int idx = tid + t * NT;
            v[t] = in[idx < n0 ? idx : n0 - 1];
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
    int lgr = a.lgR, lgc = a.lgC;
    for (int k = 0; k < a.K; ++k) {
        const int r = 1 << lgr, c = 1 << lgc, lgc2 = lgc - 1, c2 = c >> 1;
        const int n_half = 1 << (lgr + lgc2), n_quarter = n_half >> 1;
        real_t* tL = tmp;
        real_t* tH = tmp + n_half;
        PDWT_FOR_THREADS(tid, NT) {  // rows: (r x c) -> L | H, (r x c2) each
            for (int idx = tid; idx < n_half; idx += NT) {
                const int y = idx >> lgc2, x = idx & (c2 - 1);
                const real_t* row = cur + (y << lgc);
                const int base = 2 * x - C;
                real_t l = 0, h = 0;
#pragma unroll
                for (int j = 0; j < hlen; ++j) {
                    const real_t v = row[(base + j) & (c - 1)];
                    l = pdwt_fma(v, fLo[j], l);
                    h = pdwt_fma(v, fHi[j], h);
                }
                tL[idx] = l;
                tH[idx] = h;
            }
        }
        PDWT_SYNC();
        const bool last = k == a.K - 1;
        PDWT_FOR_THREADS(tid, NT) {  // columns: -> A (LDS, or global for the last level), H, V, D (global)
            const long long b = (long long)bz * n_quarter;
            real_t* PDWT_RESTRICT gA = a.out + b;
            real_t* PDWT_RESTRICT gH = a.det[k][0] + b;
            real_t* PDWT_RESTRICT gV = a.det[k][1] + b;
            real_t* PDWT_RESTRICT gD = a.det[k][2] + b;
            for (int idx = tid; idx < n_quarter; idx += NT) {
                const int i = idx >> lgc2, x = idx & (c2 - 1);
                const int base = 2 * i - C;
                real_t ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
                for (int j = 0; j < hlen; ++j) {
                    const int o = (((base + j) & (r - 1)) << lgc2) + x;
                    const real_t vL = tL[o], vH = tH[o];
                    ll = pdwt_fma(vL, fLo[j], ll);
                    lh = pdwt_fma(vL, fHi[j], lh);
                    hl = pdwt_fma(vH, fLo[j], hl);
                    hh = pdwt_fma(vH, fHi[j], hh);
                }
                if (last) gA[idx] = ll;
                else cur[idx] = ll;
                gH[idx] = lh;  // (row low, column high), (row high, column low), (row high, column high): separable.cu:135-176
                gV[idx] = hl;
                gD[idx] = hh;
            }
        }
        PDWT_SYNC();
        --lgr;
        --lgc;
    }
}

// Flat order of everything the inverse reads, from the coarse end: with sL = samples of A_L, positions [0, sL) are A_L and
// [4^m sL, 4^(m+1) sL) the three detail planes (4^m sL each) of the group's level K-1-m -- together exactly n0 values.
// LDS homes: positions below sL and from n0/4 on (the finest level's details) live at X[f]; the rest at U[n0/4 + f].
PDWT_DEVICE int tail_log4(unsigned q) { return (31 - __builtin_clz(q)) >> 1; }  // floor(log4(q)), q >= 1

template <int HLEN, int NT>
PDWT_DEVICE void dwt2_inv_tail_image_p2(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, H2 = hlen / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    const int n0 = 1 << (a.lgR + a.lgC), q0 = n0 >> 2;
    const int lgsL = a.lgR + a.lgC - 2 * a.K, sL = 1 << lgsL;
    real_t* X = smem;       // [0, n0/4): the approximation being rebuilt; [n0/4, n0): H, V, D of the finest level
    real_t* U = smem + n0;  // column-synthesis results (u1 | u2); [n0/4, n0/2): the details of the coarser levels until the last step
    real_t* fLo = smem + 2 * n0;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        // three unrolled passes so that a thread's loads are in flight TOGETHER: the plane pointers (a lane-indexed read of the
        // kernel arguments: memory loads), then the values, then the LDS stores.  Interleaved, the in-order load counter makes
        // every pointer wait drain the data loads before it: sixteen round trips instead of two.
        const real_t* src[kTailTrips];
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            int f = tid + t * NT;
            f = f < n0 ? f : n0 - 1;
            if (f < sL) {
                src[t] = a.in + (long long)bz * sL + f;
            } else {
                const int m = tail_log4((unsigned)f >> lgsL), lgp = lgsL + 2 * m;  // plane size 2^lgp
                const int g = f - (1 << lgp), b = g >> lgp, idx = g & ((1 << lgp) - 1);
                src[t] = a.det[a.K - 1 - m][b] + ((long long)bz << lgp) + idx;
            }
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) v[t] = *src[t];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int f = tid + t * NT;
            if (f < n0) {
                if (f < sL || f >= q0) X[f] = v[t];
                else U[q0 + f] = v[t];
            }
        }
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    for (int k = a.K - 1; k >= 0; --k) {
        const int lgri = a.lgR - k - 1, lgci = a.lgC - k - 1, ri = 1 << lgri, ci = 1 << lgci, lgco = lgci + 1;
        const int n_in = 1 << (lgri + lgci);  // = 4^(K-1-k) sL
        const real_t* dH = k == 0 ? X + q0 : U + q0 + n_in;
        const real_t* dV = dH + n_in;
        const real_t* dD = dV + n_in;
        real_t* u1 = U;
        real_t* u2 = U + 2 * n_in;
        PDWT_FOR_THREADS(tid, NT) {  // column synthesis: (A, H) -> u1, (V, D) -> u2, (2 ri x ci) each
            for (int idx = tid; idx < 2 * n_in; idx += NT) {
                const int q = idx >> lgci, x = idx & (ci - 1);
                const int p = q + S;
                const int rel = (p >> 1) - C, par = 1 - (p & 1);
                real_t r1 = 0, r2 = 0;
#pragma unroll
                for (int j = 0; j < H2; ++j) {
                    const int t = 2 * j + par;  // reversed taps: f[hlen - 1 - t]
                    const int src = (((rel + j) & (ri - 1)) << lgci) + x;
                    r1 = pdwt_fma(X[src], fLo[t], r1);
                    r1 = pdwt_fma(dH[src], fHi[t], r1);
                    r2 = pdwt_fma(dV[src], fLo[t], r2);
                    r2 = pdwt_fma(dD[src], fHi[t], r2);
                }
                u1[idx] = r1;
                u2[idx] = r2;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // row synthesis: (u1, u2) -> (2 ri x 2 ci), to LDS or (the last step) global
            real_t* PDWT_RESTRICT gout = a.out + (long long)bz * 4 * n_in;
            for (int idx = tid; idx < 4 * n_in; idx += NT) {
                const int q = idx >> lgco, g = idx & (2 * ci - 1);
                const int p = g + S;
                const int rel = (p >> 1) - C, par = 1 - (p & 1);
                real_t r = 0;
#pragma unroll
                for (int j = 0; j < H2; ++j) {
                    const int t = 2 * j + par;
                    const int src = (q << lgci) + ((rel + j) & (ci - 1));
                    r = pdwt_fma(u1[src], fLo[t], r);
                    r = pdwt_fma(u2[src], fHi[t], r);
                }
                if (k == 0) gout[idx] = r;
                else X[idx] = r;
            }
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_tail_p2_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_fwd_tail_image_p2<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
}
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_tail_p2_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_inv_tail_image_p2<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
}
#endif

}  // namespace pdwt
