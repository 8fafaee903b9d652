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
    PDWT_FOR_THREADS(tid, NT) {
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
        PDWT_FOR_THREADS(tid, NT) {
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

// LDS floats: two (a,d) pair buffers sized for the two largest consecutive levels
constexpr int inv1d_fused_lds_floats(int T0, int hlen, int K) {
    (void)K;
    const int H2 = hlen / 2;
    const int m1 = T0 / 2 + 2 * H2 + 8;   // generous bound on the level-1 coefficient count
    const int m2 = T0 / 4 + 3 * H2 + 8;
    // + 8 pairs in front (the 4-sample synthesis blocks read a little outside their range) + the range table
    return 2 * (m1 + m2) + 32 + 16 + 32;
}

template <int HLEN, int T0, int NT>
PDWT_DEVICE void dwt1_inv_fused_tile(const Inv1DFusedArgs& a, int bx, int row, float* smem) {
    constexpr int H2 = HLEN / 2, C2 = H2 / 2;
    const int K = a.K;
    constexpr int M1 = T0 / 2 + 2 * H2 + 8;
    constexpr int M2 = T0 / 4 + 3 * H2 + 8;
    v2f* bufP = reinterpret_cast<v2f*>(smem) + 8;  // (a,d) pairs of the odd levels 1,3,5.. (8 pairs of front slack)
    v2f* bufQ = bufP + M1 + 8;                     // even levels 2,4,6..
    int* rng = reinterpret_cast<int*>(smem + 2 * (M1 + M2) + 32 + 16);  // [2k], [2k+1] = range of level k

    // coefficient range of every level, from the owned output range: computed once (a per-level
    // recomputation is O(K^2) scalar work per wave and showed up as 2x more SALU than VALU instructions),
    // kept in LDS because a runtime-indexed register array would live in scratch
    const int lo0 = bx * T0;
    const int hi0 = (lo0 + T0 > a.N0) ? a.N0 : lo0 + T0;
    PDWT_FOR_THREADS(tid, NT) {
        if (tid == 0) {
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
    auto range_of = [&](int level, int& l, int& h) {
        l = rng[2 * level];
        h = rng[2 * level + 1];
    };

    // ---- approximation A_K -> .x of level K's buffer
    {
        v2f* cur = (K & 1) ? bufP : bufQ;
        const int NK = a.N0 >> K;
        int loK, hiK;
        range_of(K, loK, hiK);
        PDWT_FOR_THREADS(tid, NT) {
            const float* PDWT_RESTRICT gA = a.app + (long long)row * NK;
            stage_periodic_f4<NT, 2, 2>(tid, gA, NK, loK, hiK - loK, reinterpret_cast<float*>(cur), 0);
        }
    }
    for (int k = K; k >= 1; --k) {
        v2f* cur = (k & 1) ? bufP : bufQ;  // level k pairs: .x filled (A_K or previous synthesis)
        v2f* nxt = (k & 1) ? bufQ : bufP;  // level k-1
        const int Nk = a.N0 >> k;
        int lok, hik, lom, him;  // level k coefficients, level k-1 samples
        range_of(k, lok, hik);
        range_of(k - 1, lom, him);
        const int m = hik - lok;
        // details D_k -> .y
        PDWT_FOR_THREADS(tid, NT) {
            const float* PDWT_RESTRICT gD = a.det[k - 1] + (long long)row * Nk;
            stage_periodic_f4<NT, 4, 2>(tid, gD, Nk, lok, m, reinterpret_cast<float*>(cur), 1);
        }
        PDWT_SYNC();
        // synthesis of level k-1 samples [lom, him): work item = the 4-aligned block of samples 4b .. 4b+3,
        // i.e. coefficient indices 2b and 2b+1 (inv_row_synth4: 16-B LDS reads, all four results from
        // one window of pairs, 16-B stores at the last level); samples outside [lom, him) are dropped
        PDWT_FOR_THREADS(tid, NT) {
            float* PDWT_RESTRICT out = a.out + (long long)row * a.N0;
            const int b_lo = lom >> 2, b_hi = (him - 1) >> 2;  // arithmetic shifts: floor, lom may be negative
            const int pe = (C2 + lok) & 1;                      // parity that makes the window origin 16-B aligned
            for (int i = tid; i <= b_hi - b_lo; i += NT) {
                const int kk = 2 * (b_lo + i);
                const v2f* base = cur + (kk - C2 - lok - pe);
                float res[4];
                if (pe) inv_row_synth4<HLEN, 1>(base, a.fb, res);
                else inv_row_synth4<HLEN, 0>(base, a.fb, res);
                const int g0 = 2 * kk;
                if (k > 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (g0 + c >= lom && g0 + c < him) nxt[g0 + c - lom].x = res[c];
                } else {
                    // level 0: [lom, him) = [bx T0, min(.., N0)) is a union of whole blocks inside [0, N0)
                    f32x4 v;
                    v.x = res[0]; v.y = res[1]; v.z = res[2]; v.w = res[3];
                    *reinterpret_cast<f32x4*>(out + g0) = v;
                }
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
#endif

}  // namespace pdwt
