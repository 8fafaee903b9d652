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

constexpr int fwd1d_fused_lds_floats(int TF, int hlen, int K) {
    int n1 = TF;
    for (int i = K; i > 1; --i) n1 = 2 * n1 + hlen - 2;
    const int n0 = 2 * n1 + hlen - 2;
    return ((n0 + 8 + 3) & ~3) + (n1 + 8);
}

template <int HLEN, int TF, int NT>
PDWT_DEVICE void dwt1_fwd_fused_tile(const Fwd1DFusedArgs& a, int bx, int row, float* smem) {
    constexpr int C = HLEN / 2 - 1;
    constexpr int NV = (HLEN + 2 + 3) & ~3;  // LDS floats read per thread for two adjacent outputs
    const int K = a.K;

    // region of level k: positions [s_k, s_k + n_k);  s_K = bx*TF, s_{k-1} = 2 s_k - C
    const int n1 = fwd1d_fused_n(TF, HLEN, K, 1);
    const int n0 = 2 * n1 + HLEN - 2;
    float* buf0 = smem;             // holds level 0, 2, 4, ...
    float* buf1 = smem + ((n0 + 8 + 3) & ~3);  // holds level 1, 3, 5, ... (16-B aligned)

    // positions fit in 32 bits (the host only uses these kernels for rows shorter than 2^30)
    int s0 = bx * TF;
    for (int k = K; k > 0; --k) s0 = 2 * s0 - C;

    // ---- stage the input segment (periodic), coalesced
    PDWT_FOR_THREADS(tid, NT) {
        const float* PDWT_RESTRICT in = a.in + (long long)row * a.N0;
        stage_periodic_f4<NT, 6, 1>(tid, in, a.N0, s0, n0, buf0, 0);
        if (tid < 8) buf0[n0 + tid] = 0.f;  // slack read (never used) by the last ds_read_b128
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
            for (int t = tid; 2 * t < nk; t += NT) {
                float v[NV];
                v4f w[NV / 4];
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) w[q] = lds_load16(src + 4 * t + 4 * q);
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) {
                    lds_pin(w[q]);
                    v[4 * q + 0] = w[q].x;
                    v[4 * q + 1] = w[q].y;
                    v[4 * q + 2] = w[q].z;
                    v[4 * q + 3] = w[q].w;
                }
                v2f acc0 = mk2(0.f, 0.f), acc1 = mk2(0.f, 0.f);  // (A, D) of outputs 2t and 2t+1
#pragma unroll
                for (int j = 0; j < HLEN; ++j) {
                    const v2f tap = a.fb.t[HLEN - 1 - j];
                    acc0 = fma2(bc(v[j]), tap, acc0);
                    acc1 = fma2(bc(v[2 + j]), tap, acc1);
                }
                if (k < K) {
                    f32x2 w;
                    w.x = acc0.x;
                    w.y = acc1.x;
                    *reinterpret_cast<f32x2*>(dst + 2 * t) = w;
                }
                const int p0 = sk + 2 * t, p1 = p0 + 1;
                if (p0 >= own_lo && p0 < own_hi && p0 < Nk) {
                    outD[p0] = acc0.y;
                    if (k == K) outA[p0] = acc0.x;
                }
                if (p1 >= own_lo && p1 < own_hi && p1 < Nk) {
                    outD[p1] = acc1.y;
                    if (k == K) outA[p1] = acc1.x;
                }
            }
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
template <int HLEN, int TF, int NT>
__global__ void __launch_bounds__(NT) dwt1_fwd_fused_kernel(const Fwd1DFusedArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const int row = blockIdx.x / tiles_x;
    dwt1_fwd_fused_tile<HLEN, TF, NT>(a, blockIdx.x - row * tiles_x, row, pdwt_smem);
}

template <int HLEN, int T0, int NT>
__global__ void __launch_bounds__(NT) dwt1_inv_fused_kernel(const Inv1DFusedArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const int row = blockIdx.x / tiles_x;
    dwt1_inv_fused_tile<HLEN, T0, NT>(a, blockIdx.x - row * tiles_x, row, pdwt_smem);
}
#endif

}  // namespace pdwt
