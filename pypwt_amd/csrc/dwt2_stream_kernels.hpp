// dwt2_stream_kernels.hpp -- one DECIMATED pass as a register-blocked stream, any (even) filter length in one instantiation
// (gfx950): the decimated twin of swt_stream_kernels.hpp.  The fp64 library's 2D levels of more than 20 taps (round 5).
//
// Why.  Over doubles the LDS tiles of a 22-40-tap level hold 32 x 32 outputs with 38 halo rows and columns each -- db20 4096^2,
// three levels forward + inverse, 938 us against 235 for the fp32 library (profiles/r05_f64scan.txt); the packed-fp32 two-launch
// kernels of dwt2_split_kernels.hpp (lab library) are unrolled per filter length and need > 256 VGPRs over doubles.
//
// How.  As in swt_stream_kernels.hpp a work item owns R outputs (NC adjacent columns each) along the filtered axis and streams the
// inputs they share past R stationary accumulators in chunks of R, the taps of a chunk fetched by wave-uniform index out of a
// zero-padded table in the kernel-argument segment.  Decimation changes the index arithmetic only:
//   analysis   out[k] = sum_j x[per(2 k - c + j)] f[hlen-1-j]      (c = hlen/2 - 1; pdwt/src/separable.cu:91-176): outputs
//              k0 .. k0 + R - 1 share the 2 (R - 1) + hlen inputs from 2 k0 - c on; input r meets output m with tap j = r - 2 m.
//   synthesis  out[g], p = g + s (s = 1 for even hlen/2), i = p / 2:  out[g] = sum_(j < hlen/2) a[(i - c2 + j) mod n] rlo[t] +
//              d[...] rhi[t],  t = hlen - 1 - (2 j + 1 - (p & 1)),  c2 = hlen/4   (polyphase form of separable.cu:246-328): a work
//              item owns the R coefficient positions i0 .. i0 + R - 1, i.e. the 2 R outputs g = (2 i + par - s) mod 2n, and streams
//              the R + hlen/2 - 1 coefficient pairs (a, d) from i0 - c2 on; input r meets position m with j = r - m and feeds BOTH
//              parities, each with its own sub-filter (two tap tables per filter).
// Even sizes along the filtered axis, even filter lengths (the host declines otherwise).  A launch runs one or two independent
// problems (the column pass of a 2D level: lo -> A, H and hi -> V, D; (A, H) -> L' and (V, D) -> H').
#pragma once

#include "swt_stream_kernels.hpp"

namespace pdwt {

struct DwtStreamArgs {
    const real_t* in[2][2];   // [problem][operand]: analysis reads [q][0]; synthesis [q][0] with lo, [q][1] with hi
    real_t* out[2][2];        // analysis: [q][0] = lo, [q][1] = hi; synthesis: [q][0]
    int in_rows, in_cols, out_rows, out_cols;  // plane shapes (the filtered axis halves or doubles, the other one stays)
    int batch, hlen, problems;
    long long in_bstride, out_bstride;  // elements between the images of a batch
    // analysis: t[0] = lo, t[1] = hi with t[.][kStreamPadL + j] = filter[hlen - 1 - j]; synthesis: t[0] / t[1] = lo for output parity
    // 0 / 1, t[2] / t[3] = hi, t[.][kStreamPadL + j] = filter[hlen - 1 - (2 j + 1 - par)]; zero elsewhere
    real_t t[4][kStreamTaps];
};

constexpr long long dwt_stream_waves(bool along_y, int problems, int batch, int rows, int cols, int positions, int R, int NC) {
    // positions: outputs (analysis) or coefficient positions (synthesis) along the filtered axis
    return along_y ? (long long)problems * batch * ((positions + R - 1) / R) * (((cols / NC) + 63) >> 6)
                   : (long long)problems * batch * rows * ((((positions + R - 1) / R) + 63) >> 6);
}

// ALONG_Y: lanes = adjacent column groups of NC columns, the row walk is wave-uniform.  Along x (NC = 1): a wavefront = 64 work
// items of one row, each streaming single elements.
template <bool SYN, bool ALONG_Y, int NC, int R, int NT>
PDWT_DEVICE void dwt_stream_tile(const DwtStreamArgs& a, long long block) {
    static_assert(2 * (R - 1) <= kStreamPadL && 3 * R - 2 <= kStreamPadR, "tap table padding");
    static_assert(ALONG_Y || NC == 1, "along x a work item streams single elements");
    constexpr int P = SYN ? 2 : 1, Q = 2;  // operands per input position; accumulators per owned position (lo / hi, or the two parities)
    constexpr int S = SYN ? 1 : 2;         // stream positions between two owned positions
    const int hlen = a.hlen, h2 = hlen / 2;
    const int n_in = ALONG_Y ? a.in_rows : a.in_cols;      // length of the filtered axis, input side
    const int n_out = ALONG_Y ? a.out_rows : a.out_cols;   // ... output side
    const int positions = SYN ? n_in : n_out;
    const int c = SYN ? h2 / 2 : hlen / 2 - 1;
    const int taps = SYN ? h2 : hlen;                      // stream positions one owned position spans
    const int nchunks = (S * (R - 1) + taps + R - 1) / R;
    const int shift = SYN ? ((h2 & 1) ? 0 : 1) : 0;        // synthesis: g = 2 i + par - shift
    PDWT_FOR_THREADS(tid, NT) {
        const long long wave = block * (NT / 64) + PDWT_STREAM_UNIFORM(tid >> 6);
        const int lane = tid & 63;
        const int blocks = (positions + R - 1) / R;
        int q, m0;
        long long in_base, out_base, bz, in_pitch, out_pitch;
        bool active;
        if constexpr (ALONG_Y) {
            const int QW = a.in_cols / NC, QG = (QW + 63) >> 6;
            if (wave >= (long long)a.problems * a.batch * blocks * QG) continue;
            const int blk = (int)(wave % blocks);  // the blocks of R positions are the FASTEST index: neighbours share their rows in L1 / L2
            long long t = wave / blocks;
            const int qg = (int)(t % QG);
            t /= QG;
            bz = t % a.batch;
            q = (int)(t / a.batch);
            int col = qg * 64 + lane;
            active = col < QW;
            if (!active) col = QW - 1;
            in_base = out_base = (long long)NC * col;
            in_pitch = a.in_cols;
            out_pitch = a.out_cols;
            m0 = blk * R;
        } else {
            const int TG = (blocks + 63) >> 6;
            if (wave >= (long long)a.problems * a.batch * a.in_rows * TG) continue;
            const int tg = (int)(wave % TG);
            long long t = wave / TG;
            const int y = (int)(t % a.in_rows);
            t /= a.in_rows;
            bz = t % a.batch;
            q = (int)(t / a.batch);
            int tr = tg * 64 + lane;
            active = tr < blocks;
            if (!active) tr = blocks - 1;
            in_base = (long long)y * a.in_cols;
            out_base = (long long)y * a.out_cols;
            in_pitch = out_pitch = 1;
            m0 = tr * R;
        }
        StreamPos<ALONG_Y> pos;
        pos.n = (unsigned)n_in;
        pos.step = 1u;
        pos.p = (unsigned)true_mod(S * m0 - c, n_in);
        const real_t* PDWT_RESTRICT src[P];
#pragma unroll
        for (int k = 0; k < P; ++k) src[k] = a.in[q][k] + bz * a.in_bstride + in_base;
        const real_t zero = 0;
        svec<NC> acc[R][Q];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int o = 0; o < Q; ++o)
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[m][o].v[i] = zero;
        svec<NC> b0[R][P], b1[R][P];
        auto fetch = [&](svec<NC>(&b)[R][P]) {
            if constexpr (!ALONG_Y && R % 2 == 0) {  // along x the stream is R consecutive samples: wide loads (swt_stream_kernels.hpp)
                if (pos.p + R <= pos.n) {
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        real_t run[R];
                        stream_ld_run<R>(src[k] + pos.p, run);
#pragma unroll
                        for (int u = 0; u < R; ++u) b[u][k].v[0] = run[u];
                    }
                    pos.p += R;
                    if (pos.p >= pos.n) pos.p -= pos.n;
                    return;
                }
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const long long o = (long long)pos.p * in_pitch;
#pragma unroll
                for (int k = 0; k < P; ++k) b[u][k] = stream_ld<NC>(src[k] + o);
                pos.next();
            }
        };
        // chunk k: stream positions r = k R + u (u < R) meet owned positions m (< R) with tap index j = r - S m
        auto consume = [&](svec<NC>(&b)[R][P], int k) {
            const real_t* t0 = a.t[0] + kStreamPadL + k * R;
            const real_t* t1 = a.t[1] + kStreamPadL + k * R;
            const real_t* t2 = a.t[2] + kStreamPadL + k * R;
            const real_t* t3 = a.t[3] + kStreamPadL + k * R;
#pragma unroll
            for (int u = 0; u < R; ++u) {
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int d = u - S * m;
                    if (SYN) {
                        const real_t l0 = t0[d], l1 = t1[d], g0 = t2[d], g1 = t3[d];
#pragma unroll
                        for (int i = 0; i < NC; ++i) {
                            acc[m][0].v[i] = pdwt_fma(b[u][0].v[i], l0, acc[m][0].v[i]);
                            acc[m][0].v[i] = pdwt_fma(b[u][P - 1].v[i], g0, acc[m][0].v[i]);
                            acc[m][1].v[i] = pdwt_fma(b[u][0].v[i], l1, acc[m][1].v[i]);
                            acc[m][1].v[i] = pdwt_fma(b[u][P - 1].v[i], g1, acc[m][1].v[i]);
                        }
                    } else {
                        const real_t l = t0[d], h = t1[d];
#pragma unroll
                        for (int i = 0; i < NC; ++i) {
                            acc[m][0].v[i] = pdwt_fma(b[u][0].v[i], l, acc[m][0].v[i]);
                            acc[m][1].v[i] = pdwt_fma(b[u][0].v[i], h, acc[m][1].v[i]);
                        }
                    }
                }
            }
        };
        fetch(b0);
        int k = 0;
        for (; k + 2 < nchunks; k += 2) {  // two chunks per trip, both fetches unconditional (swt_stream_kernels.hpp)
            fetch(b1);
            PDWT_STREAM_FENCE();
            consume(b0, k);
            PDWT_STREAM_FENCE();
            fetch(b0);
            PDWT_STREAM_FENCE();
            consume(b1, k + 1);
            PDWT_STREAM_FENCE();
        }
        if (k + 1 < nchunks) {
            fetch(b1);
            PDWT_STREAM_FENCE();
            consume(b0, k);
            PDWT_STREAM_FENCE();
            consume(b1, k + 1);
        } else {
            consume(b0, k);
        }
        if (!active) continue;
        const long long ob = bz * a.out_bstride + out_base;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int pm = m0 + m;
            if (pm < positions) {
                if (SYN) {
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        int g = 2 * pm + par - shift;
                        if (g < 0) g += n_out;  // position 0, parity 0 with the shifted filter: the last output
                        if (g < n_out) stream_st<NC>(a.out[q][0] + ob + (long long)g * out_pitch, acc[m][par]);
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < 2; ++o) stream_st<NC>(a.out[q][o] + ob + (long long)pm * out_pitch, acc[m][o]);
                }
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <bool SYN, bool ALONG_Y, int NC, int R, int NT>
__global__ void __launch_bounds__(NT) dwt_stream_kernel(const DwtStreamArgs a) {
    dwt_stream_tile<SYN, ALONG_Y, NC, R, NT>(a, blockIdx.x);
}
#endif

}  // namespace pdwt
