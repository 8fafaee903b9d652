// swt_stream_kernels.hpp -- one undecimated (a-trous) pass as a register-blocked STREAM, any filter length in one
// instantiation (gfx950).  The fp64 library's path for filters of 10 taps and more (round 5).
//
// Why.  The fp64 library has the LDS-tiled level kernels of swt_kernels.hpp only: a 128 x 16 tile of doubles with its halo
// rows is 112-145 KB of LDS (one workgroup of four wavefronts per CU), every multiply-add has a 16-B LDS or global operand of
// its own, and the tile's hlen - 1 halo rows are filtered again by every tile: db20 2048^2, three levels forward + inverse,
// 3.3 ms against 0.32 ms for the fp32 library (profiles/r05e_f64_ab.txt).  The packed-fp32 two-launch kernels of
// swt_split_kernels.hpp do not carry over: unrolled over 40 taps of doubles their column kernels need more than 256 VGPRs.
//
// How.  The same observation as there -- outputs ONE DILATION STEP apart share all but one of their inputs -- but as a LOOP:
// a work item owns R outputs spaced f apart along the filtered axis (NC adjacent columns each) and streams the R + hlen - 1
// inputs past its R stationary accumulators in chunks of R.  Inside a chunk the (input, output) pairs are unrolled (R x R
// multiply-adds per operand), the taps they need -- 2R - 1 consecutive entries of a zero-padded table in the kernel-argument
// segment -- are fetched by wave-uniform index (scalar loads), and the next chunk's loads are in flight while this one is
// consumed.  Nothing depends on the filter length at compile time: one kernel per (analysis | synthesis, axis, NC).
//   analysis  (1 operand -> 2 outputs):  lo[y] = sum_j in[y + (j - c) f] dlo[hlen-1-j],  hi likewise        c = hlen/2 - 1
//   synthesis (2 operands -> 1 output):  out[y] = 1/2 sum_j (a[y + (j - c) f] rlo[hlen-1-j] + d[..] rhi[hlen-1-j])   c = hlen/2
// (the semantics of swt_kernels.hpp; pdwt/src/separable.cu:409-493,553-626).  A launch runs one or two independent PROBLEMS:
// the column pass of a 2D level is two of them (lo -> A, H and hi -> V, D; (A, H) -> L' and (V, D) -> H').
// Any row count, any row length: indices walk by f and wrap per load (f < Nr, f < Nc); NC = 2 needs even rows and 16-B
// aligned planes, NC = 1 nothing.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

constexpr int kStreamPadL = 8, kStreamPadR = 16;  // zero entries before / behind the taps: R <= 8
constexpr int kStreamTaps = kStreamPadL + kMaxTaps + kStreamPadR;

struct SwtStreamArgs {
    const real_t* in[2][2];   // [problem][operand]: analysis reads [q][0]; synthesis [q][0] with lo, [q][1] with hi
    real_t* out[2][2];        // analysis: [q][0] = lo, [q][1] = hi; synthesis: [q][0]
    int Nr, Nc, f, batch, hlen, problems;
    long long in_bstride, out_bstride;  // elements between the images of a batch
    real_t soft[2][2];        // synthesis: soft threshold applied to operand [q][k] as it is loaded (0: none)
    real_t scale;             // synthesis: factor of the result (1/2)
    real_t tl[kStreamTaps], th[kStreamTaps];  // t[kStreamPadL + j] = filter[hlen - 1 - j], zero elsewhere
};

#ifdef PDWT_CPU_EMU
#define PDWT_STREAM_UNIFORM(x) (x)
#define PDWT_STREAM_FENCE() ((void)0)
#else
#define PDWT_STREAM_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define PDWT_STREAM_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

template <int NC>
struct svec {
    real_t v[NC];
};
template <int NC>
PDWT_DEVICE svec<NC> stream_ld(const real_t* p) {
    svec<NC> r;
    if constexpr (NC == 2) {
        const real2_t t = *reinterpret_cast<const real2_t*>(p);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
        r.v[0] = *p;
    }
    return r;
}
template <int NC>
PDWT_DEVICE void stream_st(real_t* p, const svec<NC>& s) {
    if constexpr (NC == 2) {
        real2_t t;
        t.x = s.v[0];
        t.y = s.v[1];
        *reinterpret_cast<real2_t*>(p) = t;
    } else {
        *p = s.v[0];
    }
}

// R CONSECUTIVE samples of a row with wide loads at element alignment (a work item that streams single elements along x at
// dilation 1 reads R of them per chunk: one 16-B load of floats, two of doubles, instead of R loads of one element at a lane stride
// of R elements -- rocprofv3 on a 2047-column image showed the synthesis row launch, all 4-B loads, as the slowest of the four)
template <int R>
PDWT_DEVICE void stream_ld_run(const real_t* p, real_t (&out)[R]) {
#ifdef PDWT_CPU_EMU
    for (int i = 0; i < R; ++i) out[i] = p[i];
#else
    static_assert(R % 2 == 0, "pairs");
    if constexpr (sizeof(real_t) == 4 && R % 4 == 0) {
        typedef real_t v4u __attribute__((ext_vector_type(4), aligned(sizeof(real_t))));
#pragma unroll
        for (int i = 0; i < R; i += 4) {
            const v4u v = *reinterpret_cast<const v4u*>(p + i);
            out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
        }
    } else {
        typedef real_t v2u __attribute__((ext_vector_type(2), aligned(sizeof(real_t))));
#pragma unroll
        for (int i = 0; i < R; i += 2) {
            const v2u v = *reinterpret_cast<const v2u*>(p + i);
            out[i] = v.x; out[i + 1] = v.y;
        }
    }
#endif
}

// where a work item stands: the element offset of its next input inside the plane (ALONG_Y: row index, wave-uniform; along x:
// column index, per lane) and how it advances
template <bool ALONG_Y>
struct StreamPos {
    unsigned p, step, n;
    PDWT_DEVICE void next() {
        const unsigned q = p + step, r = q - n;
        p = q < r ? q : r;  // 0 < step < n
    }
};

constexpr long long stream_waves_y(int problems, int batch, int Nr, int Nc, int f, int R, int NC) {
    return (long long)problems * batch * f * (((Nr + f - 1) / f + R - 1) / R) * (((Nc / NC) + 63) >> 6);
}
constexpr int stream_items_x(int Nc, int f, int R, int NC) { return ((Nc + R * f - 1) / (R * f)) * (f / NC); }
constexpr long long stream_waves_x(int problems, int batch, int Nr, int Nc, int f, int R, int NC) {
    return (long long)problems * batch * Nr * ((stream_items_x(Nc, f, R, NC) + 63) >> 6);
}

// SYN: synthesis (else analysis).  ALONG_Y: the filtered axis is y (lanes = adjacent column groups of one row: the row walk is
// wave-uniform), else x (a wavefront = 64 work items of one row; f a multiple of NC).
template <bool SYN, bool ALONG_Y, int NC, int R, int NT>
PDWT_DEVICE void swt_stream_tile(const SwtStreamArgs& a, long long block) {
    static_assert(R <= kStreamPadL && 2 * R - 2 <= kStreamPadR, "tap table padding");
    constexpr int P = SYN ? 2 : 1, Q = SYN ? 1 : 2;
    const int f = a.f, hlen = a.hlen;
    const int c = SYN ? hlen / 2 : analysis_centre(hlen);
    const int nchunks = (R + hlen - 1 + R - 1) / R;
    PDWT_FOR_THREADS(tid, NT) {
        const long long wave = block * (NT / 64) + PDWT_STREAM_UNIFORM(tid >> 6);
        const int lane = tid & 63;
        int q, m0;        // problem; position of output 0 along the filtered axis
        long long base;   // element offset of the work item's fixed coordinates (image excluded)
        long long bz;
        bool active;
        StreamPos<ALONG_Y> pos;
        long long pitch;  // elements per step of pos.p
        if constexpr (ALONG_Y) {
            // the blocks of R rows are the FASTEST index: the wavefronts of a workgroup walk consecutive blocks of one column
            // group, so of the R + hlen - 1 rows a wavefront reads all but R were just read by its neighbour on the same CU
            const int QW = a.Nc / NC, QG = (QW + 63) >> 6;
            const int blocks = ((a.Nr + f - 1) / f + R - 1) / R;
            if (wave >= (long long)a.problems * a.batch * f * blocks * QG) continue;
            const int blk = (int)(wave % blocks);
            long long t = wave / blocks;
            const int qg = (int)(t % QG);
            t /= QG;
            const int ph = (int)(t % f);
            t /= f;
            bz = t % a.batch;
            q = (int)(t / a.batch);
            int col = qg * 64 + lane;
            active = col < QW;
            if (!active) col = QW - 1;
            base = (long long)NC * col;
            m0 = ph + f * blk * R;
            pos.n = (unsigned)a.Nr;
            pitch = a.Nc;
        } else {
            const int G = f / NC;
            const int trow = stream_items_x(a.Nc, f, R, NC), TG = (trow + 63) >> 6;
            if (wave >= (long long)a.problems * a.batch * a.Nr * TG) continue;
            const int tg = (int)(wave % TG);
            long long t = wave / TG;
            const int y = (int)(t % a.Nr);
            t /= a.Nr;
            bz = t % a.batch;
            q = (int)(t / a.batch);
            int tr = tg * 64 + lane;
            active = tr < trow;
            if (!active) tr = trow - 1;
            m0 = (tr / G) * (R * f) + NC * (tr % G);
            base = (long long)y * a.Nc;
            pos.n = (unsigned)a.Nc;
            pitch = 1;
        }
        pos.step = (unsigned)f;
        pos.p = (unsigned)true_mod(m0 - c * f, (int)pos.n);
        const real_t* PDWT_RESTRICT src[P];
        real_t beta[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            src[k] = a.in[q][k] + bz * a.in_bstride + base;
            beta[k] = SYN ? a.soft[q][k] : (real_t)0;
        }
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
            if constexpr (!ALONG_Y && NC == 1 && R % 2 == 0) {
                if (f == 1) {  // (uniform) R consecutive samples: wide loads, unless this lane's run crosses the periodic wrap
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
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const long long o = (long long)pos.p * pitch;
#pragma unroll
                for (int k = 0; k < P; ++k) b[u][k] = stream_ld<NC>(src[k] + o);
                pos.next();
            }
        };
        // chunk k: inputs r = k R + u (u < R) meet outputs m (< R) with tap j = r - m; table entry kStreamPadL + j
        auto consume = [&](svec<NC>(&b)[R][P], int k) {
            const real_t* tl = a.tl + kStreamPadL + k * R;
            const real_t* th = a.th + kStreamPadL + k * R;
            if (SYN) {
#pragma unroll
                for (int kk = 0; kk < P; ++kk) {
                    if (beta[kk] != zero) {
#pragma unroll
                        for (int u = 0; u < R; ++u)
#pragma unroll
                            for (int i = 0; i < NC; ++i) b[u][kk].v[i] = soft_shrink(b[u][kk].v[i], beta[kk]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const real_t l = tl[u - m], h = th[u - m];
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        if (SYN) {
                            acc[m][0].v[i] = pdwt_fma(b[u][0].v[i], l, acc[m][0].v[i]);
                            acc[m][0].v[i] = pdwt_fma(b[u][P - 1].v[i], h, acc[m][0].v[i]);
                        } else {
                            acc[m][0].v[i] = pdwt_fma(b[u][0].v[i], l, acc[m][0].v[i]);
                            acc[m][Q - 1].v[i] = pdwt_fma(b[u][0].v[i], h, acc[m][Q - 1].v[i]);
                        }
                    }
                }
            }
        };
        // two chunks per trip, both fetches unconditional (a fetch behind a branch makes the compiler wait for EVERY load in
        // flight before it consumes the other buffer); the last one or two chunks are peeled
        fetch(b0);
        int k = 0;
        for (; k + 2 < nchunks; k += 2) {
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
        const long long ob = bz * a.out_bstride + base;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int pm = m0 + m * f;
            if (pm < (int)pos.n) {
#pragma unroll
                for (int o = 0; o < Q; ++o) {
                    svec<NC> r = acc[m][o];
                    if (SYN) {
#pragma unroll
                        for (int i = 0; i < NC; ++i) r.v[i] *= a.scale;
                    }
                    stream_st<NC>(a.out[q][o] + ob + (long long)pm * pitch, r);
                }
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <bool SYN, bool ALONG_Y, int NC, int R, int NT>
__global__ void __launch_bounds__(NT) swt_stream_kernel(const SwtStreamArgs a) {
    swt_stream_tile<SYN, ALONG_Y, NC, R, NT>(a, blockIdx.x);
}
#endif

}  // namespace pdwt
