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
// batch of loads up front (together exactly R0 x C0 values when every size is even: one global-memory round trip for the whole
// launch, not one per level; the planes' addresses come from a table in LDS, see tail_ptr_table) and then walks back up out of LDS,
// a work item producing a pair of outputs.  Workgroups of 1024 / 256 threads, or ONE wavefront per image for large batches of tiny
// images (launch_dwt2_tail.hip).
// No halo, no tile geometry: the whole plane is resident.  Power-of-two sizes: the periodic wrap is a mask and the index
// split a shift (the _p2 kernels); any other size, odd ones included (sizes by ceil-halving, the analysis repeats the last sample
// of an odd length, the synthesis drops it): conditional wrap + division (the general kernels).
// Reference: w_kern_forward_pass1/2 (pdwt/src/separable.cu:91-176), w_kern_inverse_pass1/2 (:246-328), the level loops
// w_forward_separable / w_inverse_separable (:179-236, :332-395); index conventions as in oracle/pdwt_oracle.c
// (analysis centre hlen/2 - 1; polyphase synthesis with h2 = hlen/2, c = h2/2, s = 1 - (h2 & 1)).
// Written over real_t; HLEN = 0 is the run-time filter length (every even length), 2-8 are unrolled.  The taps are copied
// to LDS once (unrolled lengths: and from there into registers, TailTaps).  CPU emulation (tests/cpu_emu): PDWT_FOR_THREADS / PDWT_SYNC as in the other LDS kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

constexpr int kTailMaxLevels = 14;
// samples of one image entering the first level of the group: the LDS planes of tail_geometry (two of that size for even sizes)
// + the taps must fit 160 KB
constexpr int kTailMaxSamples = sizeof(real_t) == 4 ? 16384 : 8192;
constexpr int kTailTrips = 16;  // values a thread stages: the launcher picks NT with R0 x C0 <= kTailTrips x NT
// the inverse keeps its 3 K plane pointers in LDS (behind the taps, 8-byte aligned): elements of real_t that takes
constexpr int kTailPtrElems = (int)(3 * kTailMaxLevels * sizeof(void*) / sizeof(real_t)) + 2;

struct TailArgs {
    const real_t* in;                 // forward: A_{l-1} (R0 x C0 per image); inverse: A_L
    real_t* out;                      // forward: A_L; inverse: A_{l-1}
    real_t* det[kTailMaxLevels][3];   // det[k] = (H, V, D) of the group's k-th level, finest first: forward written, inverse read
    int R0, C0;                       // size entering the finest level
    int lgR, lgC;                     // their base-2 logarithms when both are powers of two (the _p2 kernels), else -1
    int K;                            // levels in the group
    int hlen;                         // even
    // the general kernels' geometry (tail_geometry): sizes by ceil-halving (pdwt/src/utils.cu:24-27), LDS homes, staging order
    int r[kTailMaxLevels + 1], c[kTailMaxLevels + 1];  // size entering level k of the group; [K] = the coarsest approximation
    int fs[kTailMaxLevels + 1];       // inverse staging: flat positions [fs[k], fs[k+1]) = H | V | D of level k ([0, fs[0]) = A_L)
    int doff[kTailMaxLevels];         // inverse: LDS offset of level k's H plane (V and D follow)
    int uoff, taps;                   // LDS offsets: second buffer (forward: L | H rows; inverse: u1 | u2); the reversed taps
    FilterBank fb;                    // forward: analysis (dec_lo, dec_hi); inverse: synthesis (rec_lo, rec_hi)
};

// LDS elements of the _p2 kernels (two planes of n0 samples + the taps)
constexpr size_t tail_lds_elems(int n0) { return (size_t)2 * n0 + 2 * kMaxTaps + kTailPtrElems; }

// Geometry of the general kernels for a.R0, a.C0, a.K: level sizes, LDS layout, staging order.  Returns the LDS elements needed.
//   forward: [0, n0) the approximation being reduced | [n0, n0 + 2 r0 c1) the row pass' (L | H) | taps
//   inverse: [0, r1 c1) the approximation being rebuilt | 3 r1 c1: H, V, D of the FINEST level | U: (u1 | u2) of the step, at most
//            2 r1 c2 elements before the last step -- behind them the details of the coarser levels, all consumed when the last
//            step's 2 r0 c1 elements overwrite them | taps.
// Even sizes throughout: 2 n0 elements + taps either way, exactly the _p2 layout.
inline size_t tail_geometry(TailArgs& a, bool inverse) {
    a.r[0] = a.R0; a.c[0] = a.C0;
    for (int k = 0; k < a.K; k++) { a.r[k + 1] = (a.r[k] + 1) >> 1; a.c[k + 1] = (a.c[k] + 1) >> 1; }
    for (int k = a.K + 1; k <= kTailMaxLevels; k++) { a.r[k] = a.r[a.K]; a.c[k] = a.c[a.K]; }
    const int n0 = a.R0 * a.C0, n1 = a.r[1] * a.c[1];
    const int u_last = 2 * a.r[0] * a.c[1];
    for (int k = 0; k <= kTailMaxLevels; k++) a.fs[k] = 0;
    for (int k = 0; k < kTailMaxLevels; k++) a.doff[k] = 0;
    if (!inverse) {
        a.uoff = n0;
        a.taps = n0 + u_last;
        return (size_t)a.taps + 2 * kMaxTaps;
    }
    a.uoff = 4 * n1;
    a.fs[0] = a.r[a.K] * a.c[a.K];
    for (int k = 0; k < a.K; k++) a.fs[k + 1] = a.fs[k] + 3 * a.r[k + 1] * a.c[k + 1];
    for (int k = a.K + 1; k <= kTailMaxLevels; k++) a.fs[k] = a.fs[a.K];
    a.doff[0] = n1;
    int coarse = a.uoff + (a.K > 1 ? 2 * a.r[1] * a.c[2] : 0);
    for (int k = 1; k < a.K; k++) { a.doff[k] = coarse; coarse += 3 * a.r[k + 1] * a.c[k + 1]; }
    const int u_end = a.uoff + u_last;
    a.taps = coarse > u_end ? coarse : u_end;
    return (size_t)a.taps + 2 * kMaxTaps + kTailPtrElems;
}

// One axis of a level: its length n and the periodic wrap of an index near it.  Analysis of an odd length: the signal is first
// extended by one sample repeating its last (pdwt/src/separable.cu:114-121; per_src of oracle/pdwt_oracle.c), period np = n + 1.
// A conditional add / subtract; the modulo only where a tiny level is shorter than the filter.
struct TailAxis {
    int n, np;
    PDWT_DEVICE int wrap(int q) const {
        q = q < 0 ? q + np : (q >= np ? q - np : q);
        q = (unsigned)q < (unsigned)np ? q : true_mod(q, np);
        return q < n ? q : n - 1;
    }
};
PDWT_DEVICE TailAxis tail_axis_analysis(int n) { TailAxis a; a.n = n; a.np = n + (n & 1); return a; }
PDWT_DEVICE TailAxis tail_axis(int n) { TailAxis a; a.n = n; a.np = n; return a; }

// The detail planes' addresses as the inverse's staging looks them up: a table in LDS, written once by one thread with UNIFORM
// indices (scalar loads of the kernel arguments).  Indexing the argument array by a per-lane level instead makes every lookup a
// vector memory load from the argument buffer: measured, those lookups were more than half of the whole inverse launch (16384
// images of 32 x 32, db2 L3: 92.6 us, 44.1 with the addresses computed arithmetically; 65536 of 16 x 16: 290 / 101).
PDWT_DEVICE real_t** tail_ptr_table(real_t* smem, int after_taps) {  // smem is 16-byte aligned: an even element offset is 8-byte aligned
    return reinterpret_cast<real_t**>(smem + ((after_taps + 1) & ~1));
}

// The reversed taps as a thread keeps them: registers for the unrolled lengths (the LDS copy cannot be hoisted out of the loops by
// the compiler: the loops store to LDS), the LDS copy for the run-time length.
template <int HLEN>
struct TailTaps {
    real_t lo[HLEN ? HLEN : 1], hi[HLEN ? HLEN : 1];
    const real_t* sLo;
    const real_t* sHi;
    PDWT_DEVICE void load(const real_t* fLo, const real_t* fHi) {
        sLo = fLo; sHi = fHi;
#pragma unroll
        for (int j = 0; j < HLEN; ++j) { lo[j] = fLo[j]; hi[j] = fHi[j]; }
    }
    PDWT_DEVICE real_t l(int t) const { return HLEN ? lo[t] : sLo[t]; }
    PDWT_DEVICE real_t h(int t) const { return HLEN ? hi[t] : sHi[t]; }
};

// Any sizes (sizes like 28, 48, 96, 100, and the odd ones they halve into -- round 4: batches of such images were 3-10x slower
// than their power-of-two neighbours on the level kernels).
template <int HLEN, int NT>
PDWT_DEVICE void dwt2_fwd_tail_image(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, C = hlen / 2 - 1;
    const int n0 = a.R0 * a.C0;
    real_t* cur = smem;
    real_t* tmp = smem + a.uoff;
    real_t* fLo = smem + a.taps;
    real_t* fHi = fLo + kMaxTaps;
    PDWT_FOR_THREADS(tid, NT) {
        const real_t* PDWT_RESTRICT in = a.in + (long long)bz * n0;
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {  // constant trip count, clamped index: all loads of a thread in flight together
            const int idx = tid + t * NT;
            if (t * NT < n0) v[t] = in[idx < n0 ? idx : n0 - 1];  // (a uniform test: tiny images skip the trips they do not need)
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
        const int r = a.r[k], c = a.c[k], c2 = a.c[k + 1], r2 = a.r[k + 1];
        const TailAxis ac = tail_axis_analysis(c), ar = tail_axis_analysis(r);
        const int n_half = r * c2, n_quarter = r2 * c2;
        real_t* tL = tmp;
        real_t* tH = tmp + n_half;
        PDWT_FOR_THREADS(tid, NT) {  // rows: (r x c) -> L | H, (r x c2) each
            for (int idx = tid; idx < n_half; idx += NT) {
                const int y = idx / c2, x = idx - y * c2;
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
                const int i = idx / c2, x = idx - i * c2;
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

// The inverse stages everything it reads in one batch of loads, in the flat order of TailArgs::fs (A_L, then H | V | D of the
// finest level, the next coarser one, ...), into the LDS homes of tail_geometry.  An output size may be odd (2 n_in - 1: the virtual
// last sample is dropped, separable.cu:296).
template <int HLEN, int NT>
PDWT_DEVICE void dwt2_inv_tail_image(const TailArgs& a, int bz, real_t* smem) {
    const int hlen = HLEN ? HLEN : a.hlen, H2 = hlen / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    const int sL = a.fs[0], total = a.fs[a.K];
    real_t* X = smem;
    real_t* U = smem + a.uoff;
    real_t* fLo = smem + a.taps;
    real_t* fHi = fLo + kMaxTaps;
    real_t** ptab = tail_ptr_table(smem, a.taps + 2 * kMaxTaps);
    PDWT_FOR_THREADS(tid, NT) {
        if (tid == 0)
            for (int k = 0; k < a.K; ++k)
                for (int b = 0; b < 3; ++b) ptab[3 * k + b] = a.det[k][b];
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        // three unrolled passes so that a thread's loads are in flight TOGETHER: the addresses (plane pointers out of the LDS
        // table), then the values, then the LDS stores.
        // (the level of a flat position by walking the sizes in registers: a lane-indexed read of fs[] / doff[] would be a dependent
        // memory load per step of the search)
        const int n1 = a.r[1] * a.c[1], coarse0 = a.uoff + (a.K > 1 ? 2 * a.r[1] * a.c[2] : 0);
        auto locate = [&](int f, int& home) -> const real_t* {
            if (f < sL) { home = f; return a.in + (long long)bz * sL + f; }
            int k = 0, rr = a.r[1], cc = a.c[1], pm = n1, start = sL, hb = n1, nxt = coarse0;
            while (f >= start + 3 * pm) {
                start += 3 * pm; ++k;
                rr = (rr + 1) >> 1; cc = (cc + 1) >> 1; pm = rr * cc;
                hb = nxt; nxt += 3 * pm;
            }
            const int g = f - start, b = (g >= pm) + (g >= 2 * pm);
            home = hb + g;
            return ptab[3 * k + b] + (long long)bz * pm + (g - b * pm);
        };
        const real_t* src[kTailTrips];
        int home[kTailTrips];
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int f = tid + t * NT;
            if (t * NT < total) src[t] = locate(f < total ? f : total - 1, home[t]);  // (a uniform test: tiny images skip trips)
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t)
            if (t * NT < total) v[t] = *src[t];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t)
            if (tid + t * NT < total) smem[home[t]] = v[t];
        for (int f = tid + kTailTrips * NT; f < total; f += NT) {  // odd sizes: a few values more than R0 x C0
            int h;
            const real_t* s = locate(f, h);
            smem[h] = *s;
        }
    }
    PDWT_SYNC();
    // A work item is a PAIR of outputs (2m, 2m + 1): with p = g + S their windows start at m - C and m - C + S, so the pair reads
    // H2 + S sources instead of 2 H2, the tap parities are compile-time constants (even taps for one output, odd for the other:
    // registers, where the one-output form read its taps from LDS by a run-time index) and there is one index split per pair.
    //   out[2m]     = sum_j s[j]     f[2j + 1 - S],   out[2m + 1] = sum_j s[j + S] f[2j + S],   s[i] = in[wrap(m - C + i)]
    for (int k = a.K - 1; k >= 0; --k) {
        const int ri = a.r[k + 1], ci = a.c[k + 1], ro = a.r[k], co = a.c[k];
        const TailAxis ari = tail_axis(ri), aci = tail_axis(ci);
        const int n_in = ri * ci, n_u = ro * ci, mp = (ro + 1) >> 1, cp = (co + 1) >> 1;
        const real_t* dH = smem + a.doff[k];
        const real_t* dV = dH + n_in;
        const real_t* dD = dV + n_in;
        real_t* u1 = U;
        real_t* u2 = U + n_u;
        PDWT_FOR_THREADS(tid, NT) {  // column synthesis: (A, H) -> u1, (V, D) -> u2, (ro x ci) each
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            for (int idx = tid; idx < mp * ci; idx += NT) {
                const int m = idx / ci, x = idx - m * ci;
                real_t e1 = 0, e2 = 0, o1 = 0, o2 = 0;
#pragma unroll
                for (int i = 0; i < H2 + S; ++i) {
                    const int src = ari.wrap(m - C + i) * ci + x;
                    const real_t vA = X[src], vH = dH[src], vV = dV[src], vD = dD[src];
                    if (i < H2) {
                        const int t = 2 * i + 1 - S;
                        e1 = pdwt_fma(vA, f.l(t), e1);
                        e1 = pdwt_fma(vH, f.h(t), e1);
                        e2 = pdwt_fma(vV, f.l(t), e2);
                        e2 = pdwt_fma(vD, f.h(t), e2);
                    }
                    if (i >= S) {
                        const int t = 2 * (i - S) + S;
                        o1 = pdwt_fma(vA, f.l(t), o1);
                        o1 = pdwt_fma(vH, f.h(t), o1);
                        o2 = pdwt_fma(vV, f.l(t), o2);
                        o2 = pdwt_fma(vD, f.h(t), o2);
                    }
                }
                const int o = 2 * m * ci + x;
                u1[o] = e1;
                u2[o] = e2;
                if (2 * m + 1 < ro) { u1[o + ci] = o1; u2[o + ci] = o2; }
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // row synthesis: (u1, u2) -> (ro x co), to LDS or (the last step) global
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            real_t* PDWT_RESTRICT gout = a.out + (long long)bz * ro * co;
            for (int idx = tid; idx < ro * cp; idx += NT) {
                const int q = idx / cp, m = idx - q * cp;
                real_t e = 0, od = 0;
#pragma unroll
                for (int i = 0; i < H2 + S; ++i) {
                    const int src = q * ci + aci.wrap(m - C + i);
                    const real_t v1 = u1[src], v2 = u2[src];
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
                real_t* dst = (k == 0 ? gout : X) + q * co + 2 * m;
                dst[0] = e;
                if (2 * m + 1 < co) dst[1] = od;
            }
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_fwd_tail_image<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
}
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    dwt2_inv_tail_image<HLEN, NT>(a, blockIdx.x, reinterpret_cast<real_t*>(tail_smem));
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
            if (t * NT < n0) v[t] = in[idx < n0 ? idx : n0 - 1];  // (a uniform test: tiny images skip the trips they do not need)
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
    real_t** ptab = tail_ptr_table(smem, 2 * n0 + 2 * kMaxTaps);
    PDWT_FOR_THREADS(tid, NT) {
        if (tid == 0)
            for (int k = 0; k < a.K; ++k)
                for (int b = 0; b < 3; ++b) ptab[3 * k + b] = a.det[k][b];
        for (int j = tid; j < hlen; j += NT) {
            fLo[j] = a.fb.lo[hlen - 1 - j];
            fHi[j] = a.fb.hi[hlen - 1 - j];
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        // three unrolled passes so that a thread's loads are in flight TOGETHER: the addresses (plane pointers out of the LDS
        // table), then the values, then the LDS stores.
        const real_t* src[kTailTrips];
        real_t v[kTailTrips];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            if (t * NT >= n0) continue;  // (a uniform test: tiny images skip the trips they do not need)
            int f = tid + t * NT;
            f = f < n0 ? f : n0 - 1;
            if (f < sL) {
                src[t] = a.in + (long long)bz * sL + f;
            } else {
                const int m = tail_log4((unsigned)f >> lgsL), lgp = lgsL + 2 * m;  // plane size 2^lgp
                const int g = f - (1 << lgp), b = g >> lgp, idx = g & ((1 << lgp) - 1);
                src[t] = ptab[3 * (a.K - 1 - m) + b] + ((long long)bz << lgp) + idx;
            }
        }
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t)
            if (t * NT < n0) v[t] = *src[t];
#pragma unroll
        for (int t = 0; t < kTailTrips; ++t) {
            const int f = tid + t * NT;
            if (f < n0) {
                if (f < sL || f >= q0) X[f] = v[t];
                else U[q0 + f] = v[t];
            }
        }
    }
    PDWT_SYNC();
    // pairs of outputs per work item, taps in registers: see dwt2_inv_tail_image
    for (int k = a.K - 1; k >= 0; --k) {
        const int lgri = a.lgR - k - 1, lgci = a.lgC - k - 1, ri = 1 << lgri, ci = 1 << lgci, lgco = lgci + 1;
        const int n_in = 1 << (lgri + lgci);  // = 4^(K-1-k) sL
        const real_t* dH = k == 0 ? X + q0 : U + q0 + n_in;
        const real_t* dV = dH + n_in;
        const real_t* dD = dV + n_in;
        real_t* u1 = U;
        real_t* u2 = U + 2 * n_in;
        PDWT_FOR_THREADS(tid, NT) {  // column synthesis: (A, H) -> u1, (V, D) -> u2, (2 ri x ci) each
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            for (int idx = tid; idx < n_in; idx += NT) {
                const int m = idx >> lgci, x = idx & (ci - 1);
                real_t e1 = 0, e2 = 0, o1 = 0, o2 = 0;
#pragma unroll
                for (int i = 0; i < H2 + S; ++i) {
                    const int src = (((m - C + i) & (ri - 1)) << lgci) + x;
                    const real_t vA = X[src], vH = dH[src], vV = dV[src], vD = dD[src];
                    if (i < H2) {
                        const int t = 2 * i + 1 - S;
                        e1 = pdwt_fma(vA, f.l(t), e1);
                        e1 = pdwt_fma(vH, f.h(t), e1);
                        e2 = pdwt_fma(vV, f.l(t), e2);
                        e2 = pdwt_fma(vD, f.h(t), e2);
                    }
                    if (i >= S) {
                        const int t = 2 * (i - S) + S;
                        o1 = pdwt_fma(vA, f.l(t), o1);
                        o1 = pdwt_fma(vH, f.h(t), o1);
                        o2 = pdwt_fma(vV, f.l(t), o2);
                        o2 = pdwt_fma(vD, f.h(t), o2);
                    }
                }
                const int o = (m << lgco) + x;  // row 2m of ci columns
                u1[o] = e1;
                u2[o] = e2;
                u1[o + ci] = o1;
                u2[o + ci] = o2;
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // row synthesis: (u1, u2) -> (2 ri x 2 ci), to LDS or (the last step) global
            TailTaps<HLEN> f;
            f.load(fLo, fHi);
            real_t* PDWT_RESTRICT gout = a.out + (long long)bz * 4 * n_in;
            for (int idx = tid; idx < 2 * n_in; idx += NT) {
                const int q = idx >> lgci, m = idx & (ci - 1);
                real_t e = 0, od = 0;
#pragma unroll
                for (int i = 0; i < H2 + S; ++i) {
                    const int src = (q << lgci) + ((m - C + i) & (ci - 1));
                    const real_t v1 = u1[src], v2 = u2[src];
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
                real_t* dst = (k == 0 ? gout : X) + 2 * idx;  // (q, 2m) of 2 ci columns: an aligned pair
                dst[0] = e;
                dst[1] = od;
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
