// swt_invstream_kernels.hpp -- one INVERSE level of the undecimated (a-trous) 2D transform in ONE launch for filters of 6-40
// taps: the synthesis along x and along y streamed down column strips, nothing but the reconstruction leaves the chip
// (gfx950, fp32).  The twin of swt_fwdstream_kernels.hpp.
//
// The reference (pdwt/src/separable.cu:553-672) and the two-launch path here (swt_split_kernels.hpp, swt_colstream_kernels.hpp) run
// the column synthesis first; the two passes commute, and row-first is what lets a strip walk do both: the x halo of a strip is
// only LOADED (column-first, it would have to be filtered along y as well).  With P, Q the row-synthesised halves
//     P = (Lx A + Hx V) / 2      (the bands that are low along y)         Q = (Lx H + Hx D) / 2      (high along y)
//     out = (Ly P + Hy Q) / 2
// (fp32 rounding differs from column-first, nothing else).  20 B per sample are compulsory (four planes in, one out); the tiles
// of swt_kernels.hpp filter their halo rows again, the two launches move 28 B through scratch.
//
// Walk (per workgroup: a strip of TXC columns of one chain of rows, steps of TY rows), four LDS-only barriers per step:
//   A  carry the history to the top of the (P, Q) buffer; row pass P out of the staged (A, V) pairs
//   B  stage the step's (H, D) pairs over them (their loads were issued during A), issue the next step's (A, V) loads
//   C  row pass Q
//   D  column pass out of the buffer (M output rows per work item), stage the next step's (A, V), issue its (H, D) loads
// A pending soft threshold is applied to H, V, D as they are staged.  The staged rows are de-interleaved by column phase of the
// dilation, as in the forward.  Packed arithmetic: a pair (A, V) / (H, D) / (P, Q) times the tap pair (rlo, rhi) element by
// element, the two halves added at the end -- 80 packed FMAs per sample in the row pass, 40 in the column pass.
// Index convention of swt_split_kernels.hpp: out[i] = 1/2 sum_j in[i + (j - c) f] t[j], c = hlen / 2, t[j] = (rlo, rhi)[hlen - 1 - j].
// Preconditions (the launcher checks them): even hlen 6-40, f = F in {1, 2, 4, 8}, chains of at least TY rows, planes below 4 GiB; any
// width (swt_stage_pad in swt_fwdstream_kernels.hpp).
#pragma once

#include "swt_fwdstream_kernels.hpp"

namespace pdwt {

struct SwtInvStreamArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nr, Nc;
    long long bstride;
    int strips, segs, seg;
    SwtWalk wk;
    real_t soft_beta;      // soft threshold of H, V, D (0: none)
    FilterBankI t;         // t[j] = (rlo[hlen - 1 - j], rhi[hlen - 1 - j])
};

template <int HLEN, int F, int TXC, int TY>
struct SwtInvStreamGeom {
    using GF = SwtFwdStreamGeom<HLEN, F, TXC, TY>;
    static constexpr int C = HLEN / 2;
    static constexpr int D = HLEN - 1;
    static constexpr int W = (D + TY - 1) / TY;
    static constexpr int SKIP = W * TY - D;
    static constexpr int BR = D + TY;
    static constexpr int XS = GF::XS;
    static constexpr int NQ = (3 + XS + 3) / 4;
    static constexpr int PWA = GF::PWA;                          // pairs of one phase, padded
    static constexpr int RXA = (F * PWA + 30) / 32 * 32 + 1;     // staged pairs per row: a pitch of 2 words mod 64 (8-B reads of lanes on consecutive rows)
    static constexpr int BP = GF::BP;                            // (P, Q) pairs per buffer row
    static constexpr int LDS_REALS = 2 * TY * RXA + 2 * (BR * BP + 1);
};

template <int HLEN, int F, int TXC, int TY, int NT, int KB, int M>
PDWT_DEVICE void swt_invstream_wg(const SwtInvStreamArgs& a, int strip, int py, int seg, int bz, real_t* smem) {
    using G = SwtInvStreamGeom<HLEN, F, TXC, TY>;
    constexpr int C = G::C, D = G::D, W = G::W, NQ = G::NQ, PWA = G::PWA, RXA = G::RXA, XS = G::XS, BP = G::BP;
    static_assert(TXC % (F * KB) == 0, "row-pass items tile the phases");
    static_assert((TY / M) * TXC == NT && TXC % 64 == 0 && NT % 64 == 0, "one column-pass item per thread, one block of M rows per wavefront");
    constexpr int TOTAL = TY * NQ, TRIPS = (TOTAL + NT - 1) / NT;
    constexpr int CARRY = (D * BP + 1) / 2, CTRIPS = (CARRY + NT - 1) / NT;

    v2f* sIn = reinterpret_cast<v2f*>(smem);                    // TY x RXA pairs: (A, V), then (H, D), phases de-interleaved
    v2f* buf = reinterpret_cast<v2f*>(smem + 2 * TY * RXA);    // BR x BP (P, Q) pairs

    const int rows_phase = a.wk.rows_phase;
    const int k0 = strip * TXC, i0 = seg * a.seg;
    const int nm = rows_phase - i0 < a.seg ? rows_phase - i0 : a.seg;
    if (nm <= 0) return;
    const int T = W + (nm + TY - 1) / TY;
    const int pbase = i0 - C + D - W * TY;
    const int padl = swt_stage_pad(k0 - C * F, XS, a.Nc);
    const int xa = k0 - C * F - padl;
    const long long boff = (long long)bz * a.bstride;
    const real_t* PDWT_RESTRICT pA = a.A + boff;
    const real_t* PDWT_RESTRICT pH = a.H + boff;
    const real_t* PDWT_RESTRICT pV = a.V + boff;
    const real_t* PDWT_RESTRICT pD = a.D + boff;
    const LanePlane pO = lane_plane(a.out + boff);
    v2f tv[HLEN];
#pragma unroll
    for (int j = 0; j < HLEN; ++j) tv[j] = in_vgprs(a.t.t[j]);
    const real_t beta = a.soft_beta;
    const bool soft = beta != real_t(0);
    const real_t half = real_t(0.5);

    // per staged group: source column, chain position of the NEXT (A, V) rows, element offset of the rows the pending loads came from
    // ((H, D) follow (A, V) of the same rows; planes are below 2^30 elements).  The LDS position is recomputed from the group's number
    // when it is written: at 40 taps the kernel needs every register it can get (tap table 80, loads in flight 32-48)
    PDWT_PER_THREAD(int, plan, 2 * TRIPS, NT);
    PDWT_PER_THREAD(int, src, TRIPS, NT);
    PDWT_PER_THREAD(v4f, pre, 2 * TRIPS, NT);
    PDWT_PER_THREAD(v4f, car, CTRIPS > 0 ? CTRIPS : 1, NT);
    auto make_plan = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int r = idx / NQ;
            const int g = idx - r * NQ;
            pl[2 * q + 0] = wrap_periodic(xa + 4 * g, a.Nc);
            pl[2 * q + 1] = true_mod(pbase + r, rows_phase);
        }
    };
    auto issue_av = [&](int tid) {  // the next step's rows: (A, V); remembers where, advances the plan
        int* pl = PDWT_MINE(plan, tid);
        int* so = PDWT_MINE(src, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int pos = pl[2 * q + 1];
            const int o = swt_walk_row<true, 1>(a.wk, a.Nr, py, pos * F) * a.Nc + pl[2 * q];
            so[q] = o;
            p[2 * q] = swt_ld16<true>(pA, kRealBytes * (unsigned)o);
            p[2 * q + 1] = swt_ld16<true>(pV, kRealBytes * (unsigned)o);
            const int np = pos + TY;
            pl[2 * q + 1] = np >= rows_phase ? np - rows_phase : np;
        }
    };
    auto issue_hd = [&](int tid) {  // (H, D) of the rows whose (A, V) were issued last
        const int* so = PDWT_MINE(src, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            p[2 * q] = swt_ld16<true>(pH, kRealBytes * (unsigned)so[q]);
            p[2 * q + 1] = swt_ld16<true>(pD, kRealBytes * (unsigned)so[q]);
        }
    };
    // the loaded quads of two planes -> (first, second) pairs, phase u mod F, index u / F; `first_soft`: the first plane is a detail band
    auto commit = [&](int tid, bool first_soft) {
        const v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int r = idx / NQ;
            v2f* row = sIn + r * RXA;
            const int u0 = 4 * (idx - r * NQ) - padl;
            real_t x[4] = {p[2 * q].x, p[2 * q].y, p[2 * q].z, p[2 * q].w};
            real_t y[4] = {p[2 * q + 1].x, p[2 * q + 1].y, p[2 * q + 1].z, p[2 * q + 1].w};
            if (soft) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (first_soft) x[e] = soft_shrink(x[e], beta);
                    y[e] = soft_shrink(y[e], beta);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int u = u0 + e;
                if (u >= 0 && u < XS) row[(u % F) * PWA + u / F] = mk2(x[e], y[e]);
            }
        }
    };
    auto carry_read = [&](int tid) {
        v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            c[q] = lds_load16(buf + TY * BP + 2 * idx);
        }
    };
    auto carry_write = [&](int tid) {
        const v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            if (((D * BP) & 1) && idx == CARRY - 1) buf[2 * idx] = mk2(c[q].x, c[q].y);  // an odd number of pairs: the last group is half a group (its other half is row D, which this step writes)
            else *reinterpret_cast<v4f*>(buf + 2 * idx) = c[q];
        }
    };

    // ---- row pass of the staged pairs into half `second` (0: P, 1: Q) of buffer rows [D, D + TY): item = (row, phase, block of KB
    // outputs of the phase); output i of phase ph (column ph + F i) reads the phase's pairs i .. i + hlen - 1
    auto row_pass = [&](int tid, int first_row, int second) {
        constexpr int NB = TXC / (F * KB), ITEMS = TY * F * NB;
        constexpr int NWIN = KB - 1 + HLEN;
        constexpr int GB = 4, NG = (NWIN + GB - 1) / GB;
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            // (the rows are the fastest index, step 0 numbers its items over the rows it filters: swt_fwdstream_kernels.hpp)
            const int nrows = first_row ? TY - first_row : TY;
            const int rem = first_row ? it / nrows : it / TY;
            const int r = first_row + it - rem * nrows;
            const int ph = rem / NB;
            const int b = rem - ph * NB;
            if (it >= nrows * F * NB) continue;
            const v2f* p2 = sIn + r * RXA + ph * PWA + KB * b;
            v2f acc[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[kk] = mk2(real_t(0), real_t(0));
            v2f w[2][GB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int e = 0; e < GB; ++e)
                    if (g * GB + e < NWIN) w[g & 1][e] = p2[g * GB + e];
            };
            load_group(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1);
#pragma unroll
                for (int e = 0; e < GB; ++e) {
                    const int wi = g * GB + e;
                    if (wi < NWIN) {
#pragma unroll
                        for (int kk = 0; kk < KB; ++kk) {
                            const int j = wi - kk;
                            if (j >= 0 && j < HLEN) acc[kk] = fma2(w[g & 1][e], tv[j < 0 || j >= HLEN ? 0 : j], acc[kk]);
                        }
                    }
                }
            }
            real_t* dst = reinterpret_cast<real_t*>(buf + (D + r) * BP + ph + F * KB * b) + second;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) dst[2 * F * kk] = half * (acc[kk].x + acc[kk].y);
        }
    };

    // ---- column pass out of the buffer: thread = (block ch of M rows, column x); output row mm reads buffer rows ch M + mm + j
    auto col_pass = [&](int tid, int t) {
        constexpr int NWIN = M - 1 + HLEN;
        constexpr int GB = 6, NG = (NWIN + GB - 1) / GB;
        const int ch = wave_uniform(tid / TXC);
        const int x = tid - (tid / TXC) * TXC;
        const v2f* base = buf + ch * M * BP + x;
        v2f acc[M];
#pragma unroll
        for (int mm = 0; mm < M; ++mm) acc[mm] = mk2(real_t(0), real_t(0));
        v2f w[2][GB];
        auto load_group = [&](int g) {
#pragma unroll
            for (int e = 0; e < GB; ++e)
                if (g * GB + e < NWIN) w[g & 1][e] = base[(g * GB + e) * BP];
        };
        load_group(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1);
#pragma unroll
            for (int e = 0; e < GB; ++e) {
                const int i = g * GB + e;
                if (i < NWIN) {
#pragma unroll
                    for (int mm = 0; mm < M; ++mm) {
                        const int j = i - mm;
                        if (j >= 0 && j < HLEN) acc[mm] = fma2(w[g & 1][e], tv[j < 0 || j >= HLEN ? 0 : j], acc[mm]);
                    }
                }
            }
        }
        const int p0 = i0 + (t - W) * TY + ch * M;
        const int ox = k0 + x;
        if (ox < a.Nc) {
#pragma unroll
            for (int mm = 0; mm < M; ++mm) {
                if (p0 + mm < i0 + nm) {
                    const unsigned ro = (unsigned)swt_walk_row<true, 1>(a.wk, a.Nr, py, (p0 + mm) * F) * (unsigned)a.Nc;
                    st_lane(pO, ro, ox, half * (acc[mm].x + acc[mm].y));
                }
            }
        }
    };

    PDWT_FOR_THREADS(tid, NT) {
        make_plan(tid);
        issue_av(tid);
        commit(tid, false);
        issue_hd(tid);
    }
    PDWT_LONG_SYNC();
    for (int t = 0; t < T; ++t) {
        const int first = t == 0 ? G::SKIP : 0;
        PDWT_FOR_THREADS(tid, NT) {  // A
            if (t > 0) carry_write(tid);
            row_pass(tid, first, 0);
        }
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // B
            commit(tid, true);
            if (t + 1 < T) issue_av(tid);
        }
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) row_pass(tid, first, 1);  // C
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) {  // D
            if (t >= W) col_pass(tid, t);
            if (t + 1 < T) {
                carry_read(tid);
                commit(tid, false);
                issue_hd(tid);
            }
        }
        if (t + 1 < T) PDWT_LONG_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int F, int TXC, int TY, int NT, int KB, int M, int MINB>
__global__ void __launch_bounds__(NT, MINB) swt_invstream_kernel(const SwtInvStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, cs;
    if (!xcd_tile(blockIdx.x, a.strips, a.segs * a.wk.phases, strip, cs)) return;
    const int py = cs / a.segs;
    swt_invstream_wg<HLEN, F, TXC, TY, NT, KB, M>(a, strip, py, cs - py * a.segs, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
