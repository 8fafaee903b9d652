// swt_fwdstream_kernels.hpp -- one FORWARD level of the undecimated (a-trous) 2D transform in ONE launch for filters of 6-40
// taps: row pass and column pass streamed down column strips, the row-filtered (lo, hi) rows never leave LDS (gfx950, fp32).
//
// Why (round 6).  An SWT level writes four full-size planes: 20 B per sample are compulsory.  The LDS tiles of swt_kernels.hpp
// filter every tile's hlen - 1 halo rows again and sit at 0.34-0.4 of the HBM rate on 2048^2 ... 4096^2 planes (8 taps: 26 / 125 us
// per level); from 14 taps on the level runs as a row launch + a column launch with the (lo, hi) planes in scratch
// (swt_split_kernels.hpp, swt_colstream_kernels.hpp): 36 B per sample and two launches.  The reference's own benchmark is this
// transform with haar and db20 (test/benchmark.py:24-38; kernels pdwt/src/separable.cu:409-493, one load per tap and output).
//
// How: dwt2_fwd_long_wg (dwt2_long_kernels.hpp) without the decimation.  A workgroup owns a STRIP of TXC columns of one chain of
// rows (SwtWalk: rows r, r + f, r + 2 f ...) and walks down a segment of it in steps of TY rows:
//   stage   the TY new rows, TXC + (hlen - 1) f columns of each (global loads issued a step ahead), into LDS with the f column
//           PHASES of the dilation de-interleaved: phase p of a row holds the samples p, p + f, p + 2 f ... contiguously, so that
//           the dilated row filter reads a contiguous window;
//   row     a work item filters KB consecutive outputs of one phase (columns p + f i): KB + hlen - 1 samples slide past KB
//           stationary (lo, hi) accumulator pairs (16-B LDS reads, the sample broadcast against the tap pair); the results go
//           under the D = hlen - 1 rows of history in the (lo, hi) buffer, which a copy through registers moved to its top;
//   column  swt_colstream_kernels.hpp's forward pass: M output rows of a column, M + hlen - 1 buffer rows slide past them.
// Nothing is filtered twice along y except the D warm-up rows in front of a segment (row pass only); the x halo is only loaded.
// Index convention of swt_split_kernels.hpp: out[i] = sum_j in[i + (j - c) f] t[j], c = hlen / 2 - 1, t[j] = (lo, hi)[hlen - 1 - j].
// Preconditions (the launcher checks them): even hlen 6-40, f = F in {1, 2, 4, 8, 16}, chains of at least TY rows, planes below 4 GiB;
// any width (rows that are not whole 16-B groups: at least TXC + (hlen - 1) f + 4 columns, see swt_stage_pad).
#pragma once

#include "swt_colstream_kernels.hpp"

namespace pdwt {

struct SwtFwdStreamArgs {
    const real_t* in;
    real_t *A, *H, *V, *D;
    int Nr, Nc;
    long long bstride;     // elements between the images of a batch (input and output planes alike)
    int strips, segs, seg; // column strips, segments per chain, rows of a chain per segment (a multiple of TY)
    SwtWalk wk;            // swt_walk(Nr, Nc, f, 4)
    FilterBankI t;         // t[j] = (lo[hlen - 1 - j], hi[hlen - 1 - j]) of the analysis bank
};

// The 16-B groups a strip stages start `pad` samples in front of its first column x0 (which may be negative: the periodic halo).  Rows
// of whole groups (Nc % 4 == 0): the groups are aligned, pad = x0 mod 4.  Any other width: the groups are loaded at 4-B alignment, and
// none may straddle the row end -- where the staged window [x0, x0 + xs) runs past the END of the row the groups are laid out from
// there backwards (pad = (x0 - Nc) mod 4), where it starts before column 0 from 0 (the same formula as the aligned case); a window
// that does both needs Nc % 4 == 0 (the launcher declines narrower images of other widths).
PDWT_DEVICE int swt_stage_pad(int x0, int xs, int Nc) {
    const bool right_only = x0 >= 0 && x0 + xs > Nc;
    return true_mod(right_only ? x0 - Nc : x0, 4);
}

template <int HLEN, int F, int TXC, int TY>
struct SwtFwdStreamGeom {
    static_assert(F == 1 || F == 2 || F == 4 || F == 8 || F == 16, "dilations whose phases tile a strip");
    static constexpr int C = HLEN / 2 - 1;
    static constexpr int D = HLEN - 1;                           // rows of history an output row needs
    static constexpr int W = (D + TY - 1) / TY;                  // warm-up steps: staging and row pass only
    static constexpr int SKIP = W * TY - D;                      // rows of step 0 nobody needs
    static constexpr int BR = D + TY;                            // rows of the (lo, hi) buffer
    static constexpr int XS = TXC + (HLEN - 1) * F;              // staged columns of a row: column k0 - C F onwards
    static constexpr int NQ = (3 + XS + 3) / 4;                  // 16-B groups loaded per row (up to 3 samples in front of column k0 - C F)
    static constexpr int PW = TXC / F + HLEN - 1;                // samples of one phase
    static constexpr int PWA = ((PW + 3) & ~3) + 4;              // ... padded: whole 16-B groups, phases start in different banks
    // LDS banks (64 of 4 B).  The row pass has the lanes of a wavefront on consecutive ROWS: a 16-B read of 16 lanes is conflict-free
    // when the row pitch is 4 words mod 64, and so are their 8-B writes into the buffer when ITS pitch is 2 words mod 64 (rocprofv3,
    // 8 taps, lanes on consecutive blocks of one row: 1.7 M conflict cycles per 2048^2 level forward, 7.8 M inverse = half its time)
    // From 30 taps on the first layout stays -- lanes on consecutive blocks of a row, dense pitches: the conflicts are a small share of a
    // 40-tap pass and the padded layout measured 4 % SLOWER there (db20 512^2 L3 forward 32.7 -> 34.4 us; 6-20 taps: 7-12 % faster)
    static constexpr bool ROWS_FAST = HLEN < 30;
    static constexpr int RXA = ROWS_FAST ? (F * PWA + 59) / 64 * 64 + 4 : F * PWA;  // staged samples per row
    static constexpr int BP = ROWS_FAST ? TXC + 1 : TXC;                            // (lo, hi) pairs per buffer row
    static constexpr int LDS_REALS = TY * RXA + 2 * (BR * BP + 1);
};

//   KB outputs of one phase per row-pass item;  M output rows per column-pass item
template <int HLEN, int F, int TXC, int TY, int NT, int KB, int M>
PDWT_DEVICE void swt_fwdstream_wg(const SwtFwdStreamArgs& a, int strip, int py, int seg, int bz, real_t* smem) {
    using G = SwtFwdStreamGeom<HLEN, F, TXC, TY>;
    constexpr int C = G::C, D = G::D, W = G::W, NQ = G::NQ, PWA = G::PWA, RXA = G::RXA, XS = G::XS, BP = G::BP;
    static_assert(TXC % (F * KB) == 0 && (KB % 4 == 0 || TXC == F * KB), "row-pass items tile the phases in whole 16-B groups");
    static_assert((TY / M) * TXC == NT && TXC % 64 == 0 && NT % 64 == 0, "one column-pass item per thread, one block of M rows per wavefront");
    constexpr int TOTAL = TY * NQ, TRIPS = (TOTAL + NT - 1) / NT;
    constexpr int CARRY = (D * BP + 1) / 2, CTRIPS = (CARRY + NT - 1) / NT;  // 16-B groups of the D carried rows of (lo, hi) pairs

    real_t* sIn = smem;                                      // TY x RXA: the step's input rows, phases de-interleaved
    v2f* buf = reinterpret_cast<v2f*>(smem + TY * RXA);     // BR x BP (lo, hi) pairs

    const int rows_phase = a.wk.rows_phase;
    const int k0 = strip * TXC, i0 = seg * a.seg;
    const int nm = rows_phase - i0 < a.seg ? rows_phase - i0 : a.seg;  // output rows of this segment
    if (nm <= 0) return;
    const int T = W + (nm + TY - 1) / TY;
    const int pbase = i0 - C + D - W * TY;   // chain position of the first row of step 0
    const int padl = swt_stage_pad(k0 - C * F, XS, a.Nc);
    const int xa = k0 - C * F - padl;        // the origin of the row's 16-B groups
    const long long boff = (long long)bz * a.bstride;
    const real_t* PDWT_RESTRICT in = a.in + boff;
    const LanePlane pA = lane_plane(a.A + boff), pH = lane_plane(a.H + boff), pV = lane_plane(a.V + boff), pD = lane_plane(a.D + boff);
    v2f tv[HLEN];  // the tap pairs in vector registers, see dwt2_long_kernels.hpp
#pragma unroll
    for (int j = 0; j < HLEN; ++j) tv[j] = in_vgprs(a.t.t[j]);

    PDWT_PER_THREAD(int, plan, 4 * TRIPS, NT);  // LDS row base, 4 g - PADL, source column, chain position (advanced by TY per step)
    PDWT_PER_THREAD(v4f, pre, TRIPS, NT);
    PDWT_PER_THREAD(v4f, car, CTRIPS > 0 ? CTRIPS : 1, NT);
    auto make_plan = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int r = idx / NQ;
            const int g = idx - r * NQ;
            pl[4 * q + 0] = r * RXA;
            pl[4 * q + 1] = 4 * g - padl;
            pl[4 * q + 2] = wrap_periodic(xa + 4 * g, a.Nc);  // a group never straddles the row end (swt_stage_pad)
            pl[4 * q + 3] = true_mod(pbase + r, rows_phase);
        }
    };
    auto issue = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int pos = pl[4 * q + 3];
            p[q] = swt_ld16<true>(in, kRealBytes * (unsigned)(swt_walk_row<true, 1>(a.wk, a.Nr, py, pos * F) * a.Nc + pl[4 * q + 2]));
            const int np = pos + TY;
            pl[4 * q + 3] = np >= rows_phase ? np - rows_phase : np;  // rows_phase >= TY
        }
    };
    // sample u of the staged row (column k0 - C F + u) lies in phase u mod F at index u / F
    auto commit = [&](int tid) {
        const int* pl = PDWT_MINE(plan, tid);
        const v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            real_t* row = sIn + pl[4 * q];
            const int u0 = pl[4 * q + 1];
            const real_t v[4] = {p[q].x, p[q].y, p[q].z, p[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int u = u0 + e;
                if (u >= 0 && u < XS) row[(u % F) * PWA + u / F] = v[e];
            }
        }
    };
    auto carry_read = [&](int tid) {  // buffer rows [TY, TY + D)
        v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            c[q] = lds_load16(buf + TY * BP + 2 * idx);
        }
    };
    auto carry_write = [&](int tid) {  // ... to rows [0, D)
        const v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            if (((D * BP) & 1) && idx == CARRY - 1) buf[2 * idx] = mk2(c[q].x, c[q].y);  // an odd number of pairs: the last group is half a group (its other half is row D, which this step writes)
            else *reinterpret_cast<v4f*>(buf + 2 * idx) = c[q];
        }
    };

    // ---- row pass of the step's TY rows into buffer rows [D, D + TY): item = (row, phase, block of KB outputs of the phase);
    // output i of phase ph (column ph + F i) reads the phase's samples i .. i + hlen - 1
    auto row_pass = [&](int tid, int first_row) {
        constexpr int NB = TXC / (F * KB), ITEMS = TY * F * NB;
        constexpr int NQW = (KB - 1 + HLEN + 3) / 4;  // 16-B groups of a window
        constexpr int GB = 3, NG = (NQW + GB - 1) / GB;
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            // the rows are the fastest index (the pitches of SwtFwdStreamGeom); step 0 filters rows first_row .. TY - 1 only: its
            // items are numbered over those rows, so that whole wavefronts drop out instead of most lanes of every wavefront
            int r, ph, b;
            if constexpr (G::ROWS_FAST) {
                const int nrows = first_row ? TY - first_row : TY;
                const int rem = first_row ? it / nrows : it / TY;
                r = first_row + it - rem * nrows;
                ph = rem / NB;
                b = rem - ph * NB;
                if (it >= nrows * F * NB) continue;
            } else {
                r = it / (F * NB);
                const int rem = it - r * (F * NB);
                ph = rem / NB;
                b = rem - ph * NB;
                if (it >= ITEMS || r < first_row) continue;
            }
            const real_t* p4 = sIn + r * RXA + ph * PWA + KB * b;
            v2f acc[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[kk] = mk2(real_t(0), real_t(0));
            v4f w[2][GB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int e = 0; e < GB; ++e)
                    if (g * GB + e < NQW) w[g & 1][e] = lds_load16(p4 + 4 * (g * GB + e));
            };
            load_group(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1);
#pragma unroll
                for (int e = 0; e < GB; ++e) {
                    if (g * GB + e < NQW) {
                        v4f& cur = w[g & 1][e];
                        lds_pin(cur);
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const int wi = 4 * (g * GB + e) + h;  // window sample
                            const v2f pr = h < 2 ? mk2(cur.x, cur.y) : mk2(cur.z, cur.w);
#pragma unroll
                            for (int kk = 0; kk < KB; ++kk) {
                                const int j = wi - kk;
                                if (j >= 0 && j < HLEN) {
                                    const v2f tap = tv[j < 0 || j >= HLEN ? 0 : j];
                                    acc[kk] = (h & 1) ? fma2_by_v(pr, tap, acc[kk]) : fma2_bx_v(pr, tap, acc[kk]);
                                }
                            }
                        }
                    }
                }
            }
            v2f* dst = buf + (D + r) * BP + ph + F * KB * b;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) dst[F * kk] = acc[kk];
        }
    };

    // ---- column pass out of the buffer (swt_colstream_kernels.hpp, forward): thread = (block ch of M rows, column x)
    auto col_pass = [&](int tid, int t) {
        constexpr int NWIN = M - 1 + HLEN;
        constexpr int GB = 6, NG = (NWIN + GB - 1) / GB;
        const int ch = wave_uniform(tid / TXC);
        const int x = tid - (tid / TXC) * TXC;
        const v2f* base = buf + ch * M * BP + x;
        v2f accAH[M], accVD[M];
#pragma unroll
        for (int mm = 0; mm < M; ++mm) accAH[mm] = accVD[mm] = mk2(real_t(0), real_t(0));
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
                    const v2f lh = w[g & 1][e];
#pragma unroll
                    for (int mm = 0; mm < M; ++mm) {
                        const int j = i - mm;
                        if (j >= 0 && j < HLEN) {
                            const v2f tap = tv[j < 0 || j >= HLEN ? 0 : j];
                            accAH[mm] = fma2_bx_v(lh, tap, accAH[mm]);
                            accVD[mm] = fma2_by_v(lh, tap, accVD[mm]);
                        }
                    }
                }
            }
        }
        const int p0 = i0 + (t - W) * TY + ch * M;  // chain position of the block's first output row (uniform)
        const int ox = k0 + x;
        if (ox < a.Nc) {
#pragma unroll
            for (int mm = 0; mm < M; ++mm) {
                if (p0 + mm < i0 + nm) {
                    const unsigned ro = (unsigned)swt_walk_row<true, 1>(a.wk, a.Nr, py, (p0 + mm) * F) * (unsigned)a.Nc;
                    st_lane(pA, ro, ox, accAH[mm].x);
                    st_lane(pH, ro, ox, accAH[mm].y);
                    st_lane(pV, ro, ox, accVD[mm].x);
                    st_lane(pD, ro, ox, accVD[mm].y);
                }
            }
        }
    };

    // two barriers per step (dwt2_long_kernels.hpp): [carry_write, row pass(t)] | [column pass(t), carry_read, commit(t + 1), issue(t + 2)] |
    PDWT_FOR_THREADS(tid, NT) {
        make_plan(tid);
        issue(tid);
        commit(tid);
        if (T > 1) issue(tid);
    }
    PDWT_LONG_SYNC();
    for (int t = 0; t < T; ++t) {
        PDWT_FOR_THREADS(tid, NT) {
            if (t > 0) carry_write(tid);
            row_pass(tid, t == 0 ? G::SKIP : 0);
        }
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) {
            if (t >= W) col_pass(tid, t);
            if (t + 1 < T) {
                carry_read(tid);
                commit(tid);
                if (t + 2 < T) issue(tid);
            }
        }
        if (t + 1 < T) PDWT_LONG_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int F, int TXC, int TY, int NT, int KB, int M, int MINB>
__global__ void __launch_bounds__(NT, MINB) swt_fwdstream_kernel(const SwtFwdStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, cs;
    if (!xcd_tile(blockIdx.x, a.strips, a.segs * a.wk.phases, strip, cs)) return;
    const int py = cs / a.segs;
    swt_fwdstream_wg<HLEN, F, TXC, TY, NT, KB, M>(a, strip, py, cs - py * a.segs, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
