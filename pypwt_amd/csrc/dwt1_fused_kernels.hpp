// dwt1_fused_kernels.hpp -- ALL levels of a (batched) 1D DWT in one launch per direction (gfx950).
//
// In 1D the halo a workgroup needs for K fused levels is (2^K - 1)(hlen - 2) samples against a
// segment of 2^K * TF samples: 11 % for sym8 (hlen 16), K = 6, TF = 128.  So one workgroup stages
// its input segment once, runs the whole analysis pyramid out of LDS (two ping-pong buffers) and
// writes every detail band and the final approximation once: the transform moves its COMPULSORY
// bytes only (8 B/sample per direction) instead of re-reading and re-writing the approximation at
// every level (the per-level kernels of dwt1_kernels.hpp move 15.75 B/sample at K = 6, and the
// deep levels are launch-latency bound).  The inverse runs the pyramid backwards; its halo does not
// compound (about 2*(hlen/2) coefficients per level).
//
// Exactness: the fused kernels extend each level's approximation periodically in place, which
// equals the reference's per-level periodization (pdwt/src/separable.cu:114-121) only when every
// level length is even, i.e. 2^K divides the row length; the host uses them only then and falls
// back to the per-level kernels otherwise (odd lengths repeat the last sample at each level).
//
// Arithmetic is the same as everywhere else (restated in oracle/pdwt_oracle.c):
//   analysis   out[o] = sum_j x[2o - c + j] * f[hlen-1-j]              (separable.cu:91-131)
//   synthesis  polyphase form of separable.cu:293-328, see dwt2_kernels.hpp.
#pragma once

#include "dwt2_fast_kernels.hpp"  // inv_row_synth4
#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

constexpr int kMaxFusedLevels = 10;

struct Fwd1DFusedArgs {
    const float* in;                // (rows, N0)
    float* det[kMaxFusedLevels];    // det[k-1] = D_k, (rows, N0 >> k)
    float* app;                     // A_K, (rows, N0 >> K)
    int rows, N0, K;
    FilterBankI fb;                 // (dec_lo, dec_hi)
};

struct Inv1DFusedArgs {
    const float* app;                     // A_K
    const float* det[kMaxFusedLevels];    // D_1 .. D_K
    float* out;                           // (rows, N0)
    int rows, N0, K;
    FilterBankI fb;                       // (rec_lo, rec_hi)
};

// Stage the periodic segment [start, start+count) of a row of length N (N % 4 == 0, 16-B aligned) into
// LDS with 16-B global loads: groups are read from the 4-aligned origin below `start`, all of them
// issued before the first LDS write (UN independent loads in flight per thread), and scattered
// with the `pad = start & 3` shift so that LDS index 0 is sample `start`.  STRIDE = 1 writes floats,
// STRIDE = 2 writes the .x/.y lane `lane` of (a,d) pairs.
template <int NT, int UN, int STRIDE>
PDWT_DEVICE void stage_periodic_f4(int tid, const float* PDWT_RESTRICT row, int N, int start, int count,
                                   float* dst, int lane) {
    const int w0 = true_mod(start, N);
    const int pad = w0 & 3;
    const int o4 = w0 - pad;
    const int ngroups = (pad + count + 3) >> 2;
    const bool simple = (pad + count) <= N;  // at most one wrap: conditional subtract per group
    for (int base = tid; base < ngroups; base += NT * UN) {
        v4f v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int g = base + u * NT;
            if (g >= ngroups) g = ngroups - 1;
            int pos = o4 + 4 * g;
            if (simple) {
                if (pos >= N) pos -= N;
            } else {
                pos = true_mod(pos, N);
            }
            v[u] = *reinterpret_cast<const v4f*>(row + pos);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int g = base + u * NT;
            if (g < ngroups) {
                const int i0 = 4 * g - pad;
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (i0 + c >= 0 && i0 + c < count) dst[(i0 + c) * STRIDE + lane] = e[c];
            }
        }
    }
}

// staged sizes of the analysis pyramid: n[K] = TF, n[k-1] = 2 n[k] + hlen - 2
PDWT_DEVICE int fwd1d_fused_n(int TF, int hlen, int K, int k) {
    int n = TF;
    for (int i = K; i > k; --i) n = 2 * n + hlen - 2;
    return n;
}

// Buffer layout of the forward pyramid (K >= 2).  Every buffer has 4 floats of front padding and a
// per-level shift: local sample i of level k sits at buf[4 + sh_k + i], with
//   sh_0 = s_0 & 3 = C & 3   (s_0 = 2^K bx TF - C (2^K - 1), 4 | 2^K)   -> the staged input is copied in
//                            whole 16-B groups from its 4-aligned origin, no per-sample scatter;
//   sh_k = e_k = s_k & 1     (= C & 1 for k < K, 0 for k = K)           -> a work item owns the outputs
//                            q0 = 2t - e_k, q0 + 1, whose GLOBAL positions s_k + q0 are (even, odd): one
//                            8-B global store per band and an 8-B aligned LDS write at buf[4 + 2t].
// Output q of level k+1 reads the level-k samples 2q .. 2q + hlen - 1, i.e. buffer positions
// 4 + sh_k + 2q + j = 4t + OFF + j with OFF = 4 + sh_k - 2 e_{k+1}: three compile-time cases
// (first, middle, last level), each fully unrolled.
constexpr int fwd1d_fused_lds_floats(int TF, int hlen, int K) {
    int n1 = TF;
    for (int i = K; i > 1; --i) n1 = 2 * n1 + hlen - 2;
    const int n0 = 2 * n1 + hlen - 2;
    return ((n0 + 16 + 3) & ~3) + ((n1 + 16 + 3) & ~3);
}

// one level of the pyramid: nk outputs of level k from `src` (layout above) -> details (and A_K) to HBM,
// A_k to `dst`.  OFF as above, E = e_k, LAST = (k == K).
template <int HLEN, int OFF, int E, bool LAST, int NT>
PDWT_DEVICE void fwd1d_fused_level(int tid, const float* src, float* dst, int nk, int sk, int own_lo, int own_hi, int Nk,
                                   float* PDWT_RESTRICT outD, float* PDWT_RESTRICT outA, const FilterBankI& fb) {
    constexpr int B4 = OFF & ~3, O = OFF & 3;
    constexpr int NV = (O + HLEN + 2 + 3) & ~3;  // LDS floats read per work item (two adjacent outputs)
    for (int t = tid; 2 * t - E < nk; t += NT) {
        float v[NV];
        v4f w[NV / 4];
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) w[q] = lds_load16(src + 4 * t + B4 + 4 * q);
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            lds_pin(w[q]);
            v[4 * q + 0] = w[q].x;
            v[4 * q + 1] = w[q].y;
            v[4 * q + 2] = w[q].z;
            v[4 * q + 3] = w[q].w;
        }
        v2f acc0 = mk2(0.f, 0.f), acc1 = mk2(0.f, 0.f);  // (A, D) of outputs q0 = 2t - E and q0 + 1
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const v2f tap = fb.t[HLEN - 1 - j];
            acc0 = fma2(bc(v[O + j]), tap, acc0);
            acc1 = fma2(bc(v[O + 2 + j]), tap, acc1);
        }
        f32x2 pr;
        if (!LAST) {  // A_k for the next level (outputs outside [0, nk) land in slots nothing valid reads)
            pr.x = acc0.x;
            pr.y = acc1.x;
            *reinterpret_cast<f32x2*>(dst + 4 + 2 * t) = pr;
        }
        const int p0 = sk + 2 * t - E;  // even; the owned range has even bounds: both outputs or neither
        if (p0 >= own_lo && p0 < own_hi && p0 < Nk) {
            pr.x = acc0.y;
            pr.y = acc1.y;
            *reinterpret_cast<f32x2*>(outD + p0) = pr;
            if (LAST) {
                pr.x = acc0.x;
                pr.y = acc1.x;
                *reinterpret_cast<f32x2*>(outA + p0) = pr;
            }
        }
    }
}

template <int HLEN, int TF, int NT>
PDWT_DEVICE void dwt1_fwd_fused_tile(const Fwd1DFusedArgs& a, int bx, int row, float* smem) {
    constexpr int C = HLEN / 2 - 1, E = C & 1, SH0 = C & 3;
    const int K = a.K;  // >= 2

    // region of level k: positions [s_k, s_k + n_k);  s_K = bx*TF, s_{k-1} = 2 s_k - C
    const int n1 = fwd1d_fused_n(TF, HLEN, K, 1);
    const int n0 = 2 * n1 + HLEN - 2;
    float* buf0 = smem;                                  // holds level 0, 2, 4, ...
    float* buf1 = smem + ((n0 + 16 + 3) & ~3);           // holds level 1, 3, 5, ... (16-B aligned)

    // positions fit in 32 bits (the host only uses these kernels for rows shorter than 2^30)
    int s0 = bx * TF;
    for (int k = K; k > 0; --k) s0 = 2 * s0 - C;

    // ---- stage the input segment (periodic): whole 16-B groups from the 4-aligned origin below s0,
    //      all of a thread's loads issued before its first LDS write
    PDWT_FOR_SUBTHREADS(tid, NT) {
        const float* PDWT_RESTRICT in = a.in + (long long)row * a.N0;
        const int o4 = true_mod(s0, a.N0) - SH0;  // 4-aligned (s0 = SH0 mod 4, 4 | N0), may be -SH0 < 0 .. wraps below
        const int ngroups = (SH0 + n0 + 3) >> 2;
        constexpr int UN = 6;
        for (int base = tid; base < ngroups; base += NT * UN) {
            v4f v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                int g = base + u * NT;
                if (g >= ngroups) g = ngroups - 1;
                v[u] = *reinterpret_cast<const v4f*>(in + true_mod(o4 + 4 * g, a.N0));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int g = base + u * NT;
                if (g < ngroups) *reinterpret_cast<v4f*>(buf0 + 4 + 4 * g) = v[u];
            }
        }
    }
    PDWT_SYNC();

    float* src = buf0;
    float* dst = buf1;
    int nk = n0;
    for (int k = 1; k <= K; ++k) {
        nk = (nk - (HLEN - 2)) / 2;  // n_k
        int sk = bx * TF;             // s_k = 2^(K-k) s_K - C (2^(K-k) - 1)
        for (int i = K; i > k; --i) sk = 2 * sk - C;
        const int Nk = a.N0 >> k;
        const int own_lo = (bx * TF) << (K - k);
        const int own_hi = own_lo + (TF << (K - k));
        PDWT_FOR_SUBTHREADS(tid, NT) {
            float* PDWT_RESTRICT outD = a.det[k - 1] + (long long)row * Nk;
            float* PDWT_RESTRICT outA = a.app + (long long)row * Nk;
            if (k == 1)
                fwd1d_fused_level<HLEN, 4 + SH0 - 2 * E, E, false, NT>(tid, src, dst, nk, sk, own_lo, own_hi, Nk, outD, outA, a.fb);
            else if (k < K)
                fwd1d_fused_level<HLEN, 4 + E - 2 * E, E, false, NT>(tid, src, dst, nk, sk, own_lo, own_hi, Nk, outD, outA, a.fb);
            else
                fwd1d_fused_level<HLEN, 4 + E, 0, true, NT>(tid, src, dst, nk, sk, own_lo, own_hi, Nk, outD, outA, a.fb);
        }
        PDWT_SYNC();
        float* sw = src;
        src = dst;
        dst = sw;
    }
}

// ---------------------------------------------------------------------------
// inverse pyramid.  Level-(k-1) samples [lo, hi) need level-k coefficients
//   [ (lo+S)/2 - C2 , (hi-1+S)/2 - C2 + H2 )          (H2 = hlen/2, C2 = H2/2, S = H2 even)
// ---------------------------------------------------------------------------
PDWT_DEVICE void inv1d_fused_range(int lo, int hi, int H2, int& clo, int& chi) {
    const int C2 = H2 / 2, S = (H2 & 1) ? 0 : 1;
    clo = ((lo + S) >> 1) - C2;  // arithmetic shift = floor division, operands may be negative
    chi = ((hi - 1 + S) >> 1) - C2 + H2;
}

// ---- planar layout of the inverse pyramid -------------------------------------------------------
// Level k keeps its approximation and detail coefficients in two FLOAT planes; coefficient c of level k
// sits at plane[kInvFront + c - lok4], lok4 = the 4-aligned floor of the first needed coefficient, so
// plane positions are congruent to global indices mod 4:
//   * the detail segment is staged in whole 16-B groups (global 16-B loads, ds_write_b128, no scatter);
//   * a work item synthesises EIGHT consecutive samples 8m .. 8m+7 of level k-1 from the coefficients
//     4m .. 4m+3: its window starts at coefficient 4m - C2, always at offset O = (-C2) mod 4 inside an
//     aligned group (a compile-time constant), is read with ds_read_b128 from both planes, and the eight
//     results leave as two 16-B writes (into the next approximation plane, or to HBM at level 0).
// The earlier (a,d)-pair layout wrote every intermediate sample with a 4-B LDS store at a 32-B lane
// stride: rocprofv3 counted 7.2 M LDS bank-conflict cycles per launch against 2.5 M LDS-active cycles.
constexpr int kInvFront = 8;

constexpr int inv1d_fused_cap(int T0, int hlen, int parity) {  // plane capacity: odd levels (1), even levels (0)
    const int H2 = hlen / 2;
    return parity ? ((T0 / 2 + 2 * H2 + 32 + 3) & ~3) : ((T0 / 4 + 3 * H2 + 32 + 3) & ~3);
}

constexpr int inv1d_fused_lds_floats(int T0, int hlen, int K) {
    (void)K;
    return 2 * (inv1d_fused_cap(T0, hlen, 1) + inv1d_fused_cap(T0, hlen, 0)) + 32;  // + the range table
}

// 16-B groups [first4, first4 + 4 ngroups) of a periodic row (4 | N, first4 a multiple of 4, possibly
// negative) -> dst (16-B aligned); UN loads in flight per thread before the first LDS write
template <int NT, int UN>
PDWT_DEVICE void stage_groups(int tid, const float* PDWT_RESTRICT row, int N, int first4, int ngroups, float* dst) {
    const int w0 = true_mod(first4, N);
    const bool simple = w0 + 4 * ngroups <= 2 * N;  // at most one wrap
    for (int base = tid; base < ngroups; base += NT * UN) {
        v4f v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int g = base + u * NT;
            if (g >= ngroups) g = ngroups - 1;
            int pos = w0 + 4 * g;
            if (simple) {
                if (pos >= N) pos -= N;
            } else {
                pos = true_mod(pos, N);
            }
            v[u] = *reinterpret_cast<const v4f*>(row + pos);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int g = base + u * NT;
            if (g < ngroups) *reinterpret_cast<v4f*>(dst + 4 * g) = v[u];
        }
    }
}

// the same in PAIRS, for a row of N % 4 == 2 samples (the deepest level of a pyramid on rows of 2^(K+1) but not 2^(K+2) samples, round 5):
// first2 and N are even, a pair never straddles the periodic wrap; dst holds the logical window like stage_groups'
template <int NT, int UN>
PDWT_DEVICE void stage_pairs(int tid, const float* PDWT_RESTRICT row, int N, int first2, int npairs, float* dst) {
    const int w0 = true_mod(first2, N);
    for (int base = tid; base < npairs; base += NT * UN) {
        f32x2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int g = base + u * NT;
            if (g >= npairs) g = npairs - 1;
            v[u] = *reinterpret_cast<const f32x2*>(row + true_mod(w0 + 2 * g, N));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int g = base + u * NT;
            if (g < npairs) *reinterpret_cast<f32x2*>(dst + 2 * g) = v[u];
        }
    }
}

PDWT_DEVICE int floor4(int x) { return x - true_mod(x, 4); }

template <int HLEN, int T0, int NT>
PDWT_DEVICE void dwt1_inv_fused_tile(const Inv1DFusedArgs& a, int bx, int row, float* smem) {
    constexpr int H2 = HLEN / 2, C2 = H2 / 2, S = (H2 & 1) ? 0 : 1;
    constexpr int O = (4 - (C2 & 3)) & 3;       // offset of the window's first element in its aligned group
    constexpr int NW = H2 + 3 + S;              // window elements of four consecutive coefficients
    constexpr int NVI = (O + NW + 3) & ~3;      // floats read per plane and work item
    constexpr int NJ = H2 + S;                  // taps pairs per coefficient
    const int K = a.K;
    constexpr int CAP1 = inv1d_fused_cap(T0, HLEN, 1), CAP0 = inv1d_fused_cap(T0, HLEN, 0);
    float* aP = smem;            // odd levels 1,3,5..: approximation plane
    float* dP = aP + CAP1;       //                     detail plane
    float* aQ = dP + CAP1;       // even levels 2,4,..
    float* dQ = aQ + CAP0;
    int* rng = reinterpret_cast<int*>(dQ + CAP0);  // [2k], [2k+1] = coefficient range of level k

    const int lo0 = bx * T0;
    const int hi0 = (lo0 + T0 > a.N0) ? a.N0 : lo0 + T0;
    PDWT_FOR_SUBTHREADS(tid, NT) {
        if (tid == 0) {  // ranges once per workgroup (a per-level recomputation is O(K^2) scalar work per wave)
            int l = lo0, h = hi0;
            rng[0] = l;
            rng[1] = h;
            for (int k = 1; k <= K; ++k) {
                int cl, ch;
                inv1d_fused_range(l, h, H2, cl, ch);
                l = cl;
                h = ch;
                rng[2 * k] = l;
                rng[2 * k + 1] = h;
            }
        }
    }
    PDWT_SYNC();

    // packed tap pairs: (sample 2c, sample 2c+1) of coefficient c accumulate bc(a_e) * PA[j] + bc(d_e) * PD[j],
    // e = c - C2 + j.  S = 0: both samples use base c, taps (f[H-2-2j], f[H-1-2j]).  S = 1: sample 2c has base c
    // and tap f[H-1-2j] (j < H2), sample 2c+1 has base c+1, i.e. element j with tap f[H-2j] (j >= 1).
    v2f PA[NJ], PD[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (S == 0) {
            PA[j] = mk2(a.fb.t[HLEN - 2 - 2 * j].x, a.fb.t[HLEN - 1 - 2 * j].x);
            PD[j] = mk2(a.fb.t[HLEN - 2 - 2 * j].y, a.fb.t[HLEN - 1 - 2 * j].y);
        } else {
            const float l0 = (j < H2) ? a.fb.t[(j < H2) ? HLEN - 1 - 2 * j : 0].x : 0.f;
            const float h0 = (j < H2) ? a.fb.t[(j < H2) ? HLEN - 1 - 2 * j : 0].y : 0.f;
            const float l1 = (j >= 1) ? a.fb.t[(j >= 1) ? HLEN - 2 * j : 0].x : 0.f;
            const float h1 = (j >= 1) ? a.fb.t[(j >= 1) ? HLEN - 2 * j : 0].y : 0.f;
            PA[j] = mk2(l0, l1);
            PD[j] = mk2(h0, h1);
        }
    }

    // ---- A_K and D_K -> planes of level K
    {
        const int NK = a.N0 >> K;
        const int loK = rng[2 * K], hiK = rng[2 * K + 1];
        const int lo4 = floor4(loK);
        const int ng = (hiK - lo4 + 3) >> 2;
        float* pa = ((K & 1) ? aP : aQ) + kInvFront;
        float* pd = ((K & 1) ? dP : dQ) + kInvFront;
        PDWT_FOR_SUBTHREADS(tid, NT) {
            if (NK & 2) {  // (uniform) rows of the deepest level that are not whole quads
                stage_pairs<NT, 4>(tid, a.app + (long long)row * NK, NK, lo4, 2 * ng, pa);
                stage_pairs<NT, 4>(tid, a.det[K - 1] + (long long)row * NK, NK, lo4, 2 * ng, pd);
            } else {
                stage_groups<NT, 2>(tid, a.app + (long long)row * NK, NK, lo4, ng, pa);
                stage_groups<NT, 2>(tid, a.det[K - 1] + (long long)row * NK, NK, lo4, ng, pd);
            }
        }
    }
    PDWT_SYNC();

    for (int k = K; k >= 1; --k) {
        const float* ca = ((k & 1) ? aP : aQ) + kInvFront;   // level k planes
        const float* cd = ((k & 1) ? dP : dQ) + kInvFront;
        float* na = ((k & 1) ? aQ : aP) + kInvFront;         // level k-1 planes
        float* nd = ((k & 1) ? dQ : dP) + kInvFront;
        const int lok4 = floor4(rng[2 * k]);
        const int lom = rng[2 * k - 2], him = rng[2 * k - 1];
        const int lom4 = floor4(lom);
        PDWT_FOR_SUBTHREADS(tid, NT) {
            float* PDWT_RESTRICT out = a.out + (long long)row * a.N0;
            const int m_lo = lom >> 3, m_hi = (him - 1) >> 3;  // arithmetic shifts: floor
            for (int i = tid; i <= m_hi - m_lo; i += NT) {
                const int m = m_lo + i;
                const int base = 4 * m - C2 - O - lok4;  // plane position of the window's aligned group
                float va[NVI], vd[NVI];
                v4f wa[NVI / 4], wd[NVI / 4];
#pragma unroll
                for (int q = 0; q < NVI / 4; ++q) {
                    wa[q] = lds_load16(ca + base + 4 * q);
                    wd[q] = lds_load16(cd + base + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < NVI / 4; ++q) {
                    lds_pin(wa[q]);
                    lds_pin(wd[q]);
                    va[4 * q] = wa[q].x; va[4 * q + 1] = wa[q].y; va[4 * q + 2] = wa[q].z; va[4 * q + 3] = wa[q].w;
                    vd[4 * q] = wd[q].x; vd[4 * q + 1] = wd[q].y; vd[4 * q + 2] = wd[q].z; vd[4 * q + 3] = wd[q].w;
                }
                v2f acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = mk2(0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        // S = 1: the first element feeds only the even sample and the last only the odd one.
                        // Those are done as scalar FMAs: the other lane's tap is a structural zero, and the
                        // element may lie outside the staged range (0 x uninitialised LDS would be NaN).
                        if (S == 1 && j == 0) {
                            acc[c].x = pdwt_fma(va[O + c], PA[0].x, acc[c].x);
                            acc[c].x = pdwt_fma(vd[O + c], PD[0].x, acc[c].x);
                        } else if (S == 1 && j == NJ - 1) {
                            acc[c].y = pdwt_fma(va[O + c + j], PA[j].y, acc[c].y);
                            acc[c].y = pdwt_fma(vd[O + c + j], PD[j].y, acc[c].y);
                        } else {
                            acc[c] = fma2(bc(va[O + c + j]), PA[j], acc[c]);
                            acc[c] = fma2(bc(vd[O + c + j]), PD[j], acc[c]);
                        }
                    }
                }
                f32x4 r0, r1;
                r0.x = acc[0].x; r0.y = acc[0].y; r0.z = acc[1].x; r0.w = acc[1].y;
                r1.x = acc[2].x; r1.y = acc[2].y; r1.z = acc[3].x; r1.w = acc[3].y;
                if (k > 1) {  // samples outside [lom, him) land in slots nothing valid reads
                    float* dst = na + 8 * m - lom4;
                    *reinterpret_cast<f32x4*>(dst) = r0;
                    *reinterpret_cast<f32x4*>(dst + 4) = r1;
                } else {      // level 0: [lom, him) = [bx T0, min(.., N0)) is a union of whole 8-sample blocks
                    *reinterpret_cast<f32x4*>(out + 8 * m) = r0;
                    *reinterpret_cast<f32x4*>(out + 8 * m + 4) = r1;
                }
            }
            if (k > 1) {  // the details of level k-1, into the planes this phase does not read
                const int Nm = a.N0 >> (k - 1);
                stage_groups<NT, 4>(tid, a.det[k - 2] + (long long)row * Nm, Nm, lom4, (him - lom4 + 3) >> 2, nd);
            }
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
// Grid = 8 * chunk workgroups; ids b and b+8 share an XCD.  XCD x gets the contiguous range
// [x*chunk, (x+1)*chunk) of (row, segment) work items, so neighbouring segments -- which re-read each
// other's halo -- meet in one L2 (the plain order fetched 1.22x the algorithmic bytes).
__device__ __forceinline__ bool fused1d_item(int block, long long total, int tiles_x, int& row, int& bx) {
    const long long chunk = (total + 7) >> 3;
    const long long item = (long long)(block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || item >= total) return false;
    row = (int)(item / tiles_x);
    bx = (int)(item - (long long)row * tiles_x);
    return true;
}

template <int HLEN, int TF, int NT>
__global__ void __launch_bounds__(NT) dwt1_fwd_fused_kernel(const Fwd1DFusedArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int row, bx;
    if (!fused1d_item(blockIdx.x, (long long)tiles_x * a.rows, tiles_x, row, bx)) return;
    dwt1_fwd_fused_tile<HLEN, TF, NT>(a, bx, row, pdwt_smem);
}

template <int HLEN, int T0, int NT>
__global__ void __launch_bounds__(NT) dwt1_inv_fused_kernel(const Inv1DFusedArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int row, bx;
    if (!fused1d_item(blockIdx.x, (long long)tiles_x * a.rows, tiles_x, row, bx)) return;
    dwt1_inv_fused_tile<HLEN, T0, NT>(a, bx, row, pdwt_smem);
}

// SHORT rows (at most 512 samples): SUBS row tiles per workgroup, each run by its own sub-group of NT threads with its own slice
// of LDS -- a workgroup of 256 threads per 64-sample row left 3/4 of every wavefront slot and every barrier idle (65536 rows of
// 64 samples: 100 us forward for 16 MiB).  All sub-groups run the same phases (same K, same row length), so the workgroup
// barriers match; surplus sub-groups of the last workgroup redo the last item (identical values to identical addresses).
template <int HLEN, int TF, int NT, int SUBS>
__global__ void __launch_bounds__(NT * SUBS) dwt1_fwd_fused_rows_kernel(const Fwd1DFusedArgs a, int tiles_x, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const long long total = (long long)tiles_x * a.rows;
    const int sub = (int)(threadIdx.x / NT);
    long long item = (long long)blockIdx.x * SUBS + sub;
    if (item >= total) item = total - 1;
    const int row = (int)(item / tiles_x), bx = (int)(item - (long long)row * tiles_x);
    dwt1_fwd_fused_tile<HLEN, TF, NT>(a, bx, row, pdwt_smem + (size_t)sub * lds_floats);
}
template <int HLEN, int T0, int NT, int SUBS>
__global__ void __launch_bounds__(NT * SUBS) dwt1_inv_fused_rows_kernel(const Inv1DFusedArgs a, int tiles_x, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const long long total = (long long)tiles_x * a.rows;
    const int sub = (int)(threadIdx.x / NT);
    long long item = (long long)blockIdx.x * SUBS + sub;
    if (item >= total) item = total - 1;
    const int row = (int)(item / tiles_x), bx = (int)(item - (long long)row * tiles_x);
    dwt1_inv_fused_tile<HLEN, T0, NT>(a, bx, row, pdwt_smem + (size_t)sub * lds_floats);
}
#endif

}  // namespace pdwt
