// swt_colstream_kernels.hpp -- the COLUMN pass of an undecimated (a-trous) 2D level for long filters, streamed down the image
// with the filter's history in LDS (gfx950, fp32).
//
// Why (round 6; rocprofv3 on 2048^2 db20, profiles/r06_swt_colstream.txt).  The column kernels of swt_split_kernels.hpp keep
// everything in registers: a work item owns R = 4 outputs one dilation step apart and streams the R + hlen - 1 input rows
// past them -- 43 loads for 4 output rows at 40 taps, relying on the neighbouring wavefronts of its workgroup to have just
// pulled the same rows into L1.  The inverse (four input planes) issues 705 k wavefront loads of 1 KiB per 2048^2 level: 21 us of
// the texture path alone, 44 us for the launch where its packed multiply-adds need 16; the forward 26 us for 15.  The
// reference runs the same pass with one load per tap and output (pdwt/src/separable.cu:409-493, :553-626).
//
// How: the strip walk of dwt2_long_kernels.hpp with one pass instead of two.  A workgroup owns a STRIP of TXC columns of one
// chain of rows (SwtWalk: the rows r, r + f, r + 2 f ... -- f phases when f divides the row count) and walks down a segment of it
// in steps of TY rows.  The first step stages the whole buffer (D = hlen - 1 rows of history + TY rows); per later step the TY new
// rows (their global loads were issued a step ahead) are written under the history, which a copy through registers moved to the
// top of the buffer; then every wavefront filters M output rows of
// its column: the M + hlen - 1 buffer rows slide past M stationary accumulator pairs, one LDS read and 2 M packed multiply-adds
// per row.  Every input element is loaded from global memory ONCE per segment (+ the D warm-up rows), and read from LDS
// (M + hlen - 1) / M times per output row instead of coming hlen / R + 1 times through L1.
//   forward : (lo, hi) planes of the row launch -> A = Ly lo, H = Hy lo, V = Ly hi, D = Hy hi.  The buffer holds (lo, hi) pairs:
//             one 8-B read feeds (A, H) += lo x (tlo, thi) and (V, D) += hi x (tlo, thi) (the data half is broadcast).
//   inverse : A, H, V, D -> the interleaved (L', H') plane of the row launch, L' = (Ly A + Hy H) / 2, H' = (Ly V + Hy D) / 2.
//             The buffer holds (A, V, H, D) quadruples (a pending soft threshold applied as they are staged): one 16-B read feeds
//             (L', H') += (A, V) x rlo and += (H, D) x rhi (the tap half is broadcast).
// Index convention of swt_split_kernels.hpp: out[i] = sum_j in[i + j - c] t[j] along a chain, c = hlen / 2 - 1 (analysis) or
// hlen / 2 (synthesis), t[j] = (lo, hi)[hlen - 1 - j].
// Preconditions (the launcher checks them): even hlen 10-40, Nc % 4 == 0, 16-B aligned planes, chains of at least TY rows,
// planes below 4 GiB.
#pragma once

#include "dwt2_long_kernels.hpp"   // LanePlane / st_lane, wave_uniform, PDWT_LONG_ITEMS
#include "swt2_fused_kernels.hpp"  // SwtWalk
#include "packed_math.hpp"

namespace pdwt {

struct SwtColStreamArgs {
    const real_t* in[4];   // forward: lo, hi; inverse: A, H, V, D
    real_t* out[4];        // forward: A, H, V, D; inverse: the interleaved (L', H') plane (rows of 2 Nc)
    int Nr, Nc, f;
    long long in_bstride, out_bstride;
    int strips;            // ceil(Nc / TXC)
    int segs;              // segments per chain
    int seg;               // rows of a chain per segment (a multiple of TY)
    SwtWalk wk;            // swt_walk(Nr, Nc, f, 4)
    real_t soft_beta;      // inverse: soft threshold of H, V, D (0: none)
    FilterBankI t;         // t[j] = (lo[hlen - 1 - j], hi[hlen - 1 - j]) of the analysis / synthesis bank
};

template <int HLEN, bool INV, int TXC, int TY>
struct SwtColStreamGeom {
    static constexpr int C = INV ? HLEN / 2 : HLEN / 2 - 1;
    static constexpr int D = HLEN - 1;                     // rows of history an output row needs
    static constexpr int BR = D + TY;                      // buffer rows
    static constexpr int EL = INV ? 4 : 2;                 // reals per (row, column) of the buffer
    static constexpr int LDS_REALS = BR * TXC * EL;
};

#if !defined(PDWT_CPU_EMU) && !defined(PDWT_DOUBLE)
static __device__ __forceinline__ v2f fma2_tx_v(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(p), "v"(t));
    return acc;
}
static __device__ __forceinline__ v2f fma2_ty_v(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "v"(t));
    return acc;
}
#else
static PDWT_DEVICE v2f fma2_tx_v(v2f p, v2f t, v2f acc) { return fma2(p, bc(t.x), acc); }
static PDWT_DEVICE v2f fma2_ty_v(v2f p, v2f t, v2f acc) { return fma2(p, bc(t.y), acc); }
#endif

// one workgroup: strip `strip` of chain `py`, segment `seg`, image `bz`
template <int HLEN, bool INV, int TXC, int TY, int NT, int M>
PDWT_DEVICE void swt_colstream_wg(const SwtColStreamArgs& a, int strip, int py, int seg, int bz, real_t* smem) {
    using G = SwtColStreamGeom<HLEN, INV, TXC, TY>;
    constexpr int C = G::C, D = G::D, BR = G::BR, EL = G::EL, NPL = INV ? 4 : 2;
    static_assert(TXC % 4 == 0 && TY % M == 0 && (TY / M) * TXC == NT, "one column-pass item per thread");
    static_assert(NT % 64 == 0 && TXC % 64 == 0, "a wavefront filters ONE block of M rows: scalar row offsets");
    constexpr int QX = TXC / 4, TOTAL = TY * QX, TRIPS = (TOTAL + NT - 1) / NT;
    static_assert(TOTAL % NT == 0, "staging items tile the step");
    constexpr int TOTAL0 = BR * QX, TRIPS0 = (TOTAL0 + NT - 1) / NT;  // the first step stages the whole buffer: history and new rows
    constexpr int CARRY = D * TXC * EL / 4, CTRIPS = (CARRY + NT - 1) / NT;  // 16-B groups of the D carried rows

    const int rows_phase = a.wk.rows_phase;
    const int k0 = strip * TXC, i0 = seg * a.seg;
    const int nm = rows_phase - i0 < a.seg ? rows_phase - i0 : a.seg;  // output rows of this segment
    if (nm <= 0) return;
    const int T = (nm + TY - 1) / TY;
    const int pbase = i0 - C;  // chain position of buffer row 0 at step 0
    const long long ib = (long long)bz * a.in_bstride, ob = (long long)bz * a.out_bstride;
    v2f tv[HLEN];  // the tap pairs in vector registers (80 tap SGPRs beside the pointers spill: dwt2_long_kernels.hpp)
#pragma unroll
    for (int j = 0; j < HLEN; ++j) tv[j] = in_vgprs(a.t.t[j]);
    const real_t beta = a.soft_beta;
    const bool soft = INV && beta != real_t(0);

    PDWT_PER_THREAD(int, plan, 3 * TRIPS, NT);  // LDS offset (reals), source column, chain position (advanced by TY per step)
    PDWT_PER_THREAD(v4f, pre, NPL * TRIPS, NT);
    PDWT_PER_THREAD(v4f, car, CTRIPS, NT);
    PDWT_PER_THREAD(v4f, pre0, NPL * TRIPS0, NT);
    auto make_plan = [&](int tid) {  // steps 1 ...: the TY new rows under the history
        int* pl = PDWT_MINE(plan, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int idx = tid + q * NT;
            const int r = idx / QX;
            const int g = idx - r * QX;
            pl[3 * q + 0] = ((D + r) * TXC + 4 * g) * EL;
            const int x = k0 + 4 * g;
            pl[3 * q + 1] = x < a.Nc ? x : a.Nc - 4;  // Nc % 4 == 0: a group lies inside the row or beyond it (nothing of it is stored)
            pl[3 * q + 2] = true_mod(pbase + BR + r, rows_phase);
        }
    };
    auto issue = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int pos = pl[3 * q + 2];
            const long long o = ib + (long long)swt_walk_row<true, 1>(a.wk, a.Nr, py, pos * a.f) * a.Nc + pl[3 * q + 1];
#pragma unroll
            for (int k = 0; k < NPL; ++k) p[NPL * q + k] = *reinterpret_cast<const v4f*>(a.in[k] + o);
            const int np = pos + TY;
            pl[3 * q + 2] = np >= rows_phase ? np - rows_phase : np;  // rows_phase >= TY
        }
    };
    // one staged group of four columns: its NPL loaded quads -> the buffer's (lo, hi) pairs / (A, V, H, D) quadruples
    auto put = [&](real_t* dst, const v4f* p) {
        if constexpr (!INV) {
            const v4f lo = p[0], hi = p[1];
            v4f w0, w1;
            w0.x = lo.x; w0.y = hi.x; w0.z = lo.y; w0.w = hi.y;
            w1.x = lo.z; w1.y = hi.z; w1.z = lo.w; w1.w = hi.w;
            *reinterpret_cast<v4f*>(dst) = w0;
            *reinterpret_cast<v4f*>(dst + 4) = w1;
        } else {
            const v4f A = p[0];
            v4f H = p[1], V = p[2], Dd = p[3];
            if (soft) {
                H.x = soft_shrink(H.x, beta); H.y = soft_shrink(H.y, beta); H.z = soft_shrink(H.z, beta); H.w = soft_shrink(H.w, beta);
                V.x = soft_shrink(V.x, beta); V.y = soft_shrink(V.y, beta); V.z = soft_shrink(V.z, beta); V.w = soft_shrink(V.w, beta);
                Dd.x = soft_shrink(Dd.x, beta); Dd.y = soft_shrink(Dd.y, beta); Dd.z = soft_shrink(Dd.z, beta); Dd.w = soft_shrink(Dd.w, beta);
            }
            v4f w;
            w.x = A.x; w.y = V.x; w.z = H.x; w.w = Dd.x;
            *reinterpret_cast<v4f*>(dst) = w;
            w.x = A.y; w.y = V.y; w.z = H.y; w.w = Dd.y;
            *reinterpret_cast<v4f*>(dst + 4) = w;
            w.x = A.z; w.y = V.z; w.z = H.z; w.w = Dd.z;
            *reinterpret_cast<v4f*>(dst + 8) = w;
            w.x = A.w; w.y = V.w; w.z = H.w; w.w = Dd.w;
            *reinterpret_cast<v4f*>(dst + 12) = w;
        }
    };
    auto commit = [&](int tid) {
        const int* pl = PDWT_MINE(plan, tid);
        const v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) put(smem + pl[3 * q], p + NPL * q);
    };
    // step 0: ALL rows of the buffer, every load in flight before the first is consumed -- one memory latency in front of the first
    // output row instead of one per TY rows of history (2048^2, 40 taps: 512 workgroups of four compute steps each; two warm-up
    // steps in front of them were a third of the launch)
    auto stage_all = [&](int tid) {
        v4f* p = PDWT_MINE(pre0, tid);
#pragma unroll
        for (int q = 0; q < TRIPS0; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL0 ? idx : TOTAL0 - 1;
            const int r = idx / QX;
            const int x = k0 + 4 * (idx - r * QX);
            const int pos = true_mod(pbase + r, rows_phase);
            const long long o = ib + (long long)swt_walk_row<true, 1>(a.wk, a.Nr, py, pos * a.f) * a.Nc + (x < a.Nc ? x : a.Nc - 4);
#pragma unroll
            for (int k = 0; k < NPL; ++k) p[NPL * q + k] = *reinterpret_cast<const v4f*>(a.in[k] + o);
        }
#pragma unroll
        for (int q = 0; q < TRIPS0; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL0 ? idx : TOTAL0 - 1;  // (the surplus items of the last round rewrite the last group with the same values)
            put(smem + 4 * idx * EL, p + NPL * q);
        }
    };
    auto carry_read = [&](int tid) {  // buffer rows [TY, TY + D)
        v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            c[q] = *reinterpret_cast<const v4f*>(smem + TY * TXC * EL + 4 * idx);
        }
    };
    auto carry_write = [&](int tid) {  // ... to rows [0, D)
        const v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            *reinterpret_cast<v4f*>(smem + 4 * idx) = c[q];
        }
    };

    // ---- the column pass of step t: thread = (block ch of M rows, column x); output row mm reads buffer rows ch M + mm + j
    auto col_pass = [&](int tid, int t) {
        constexpr int NWIN = M - 1 + HLEN;
        constexpr int GB = INV ? 4 : 6, NG = (NWIN + GB - 1) / GB;
        const int ch = wave_uniform(tid / TXC);
        const int x = tid - (tid / TXC) * TXC;
        const real_t* base = smem + (ch * M * TXC + x) * EL;
        v2f acc0[M], acc1[M];  // forward: (A, H), (V, D); inverse: (L', H') from the low and the high taps
#pragma unroll
        for (int mm = 0; mm < M; ++mm) acc0[mm] = acc1[mm] = mk2(real_t(0), real_t(0));
        v4f w[2][GB];
        auto load_group = [&](int g) {
#pragma unroll
            for (int e = 0; e < GB; ++e)
                if (g * GB + e < NWIN) {
                    const real_t* src = base + (g * GB + e) * TXC * EL;
                    if constexpr (INV) {
                        w[g & 1][e] = *reinterpret_cast<const v4f*>(src);
                    } else {
                        const v2f one = *reinterpret_cast<const v2f*>(src);
                        w[g & 1][e].x = one.x;
                        w[g & 1][e].y = one.y;
                    }
                }
        };
        load_group(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1);
#pragma unroll
            for (int e = 0; e < GB; ++e) {
                const int i = g * GB + e;
                if (i < NWIN) {
                    if constexpr (INV) lds_pin(w[g & 1][e]);
                    const v2f lo = mk2(w[g & 1][e].x, w[g & 1][e].y);
#pragma unroll
                    for (int mm = 0; mm < M; ++mm) {
                        const int j = i - mm;
                        if (j >= 0 && j < HLEN) {
                            const v2f tap = tv[j < 0 || j >= HLEN ? 0 : j];
                            if constexpr (INV) {
                                const v2f hi = mk2(w[g & 1][e].z, w[g & 1][e].w);
                                acc0[mm] = fma2_tx_v(lo, tap, acc0[mm]);  // (A, V) rlo
                                acc1[mm] = fma2_ty_v(hi, tap, acc1[mm]);  // (H, D) rhi
                            } else {
                                acc0[mm] = fma2_bx_v(lo, tap, acc0[mm]);  // lo (tlo, thi)
                                acc1[mm] = fma2_by_v(lo, tap, acc1[mm]);  // hi (tlo, thi)
                            }
                        }
                    }
                }
            }
        }
        const int p0 = i0 + t * TY + ch * M;  // chain position of the block's first output row (uniform)
        const int ox = k0 + x;
        if (ox < a.Nc) {
            if constexpr (!INV) {
                const LanePlane pA = lane_plane(a.out[0] + ob), pH = lane_plane(a.out[1] + ob), pV = lane_plane(a.out[2] + ob),
                                pD = lane_plane(a.out[3] + ob);
#pragma unroll
                for (int mm = 0; mm < M; ++mm) {
                    if (p0 + mm < i0 + nm) {
                        const unsigned ro = (unsigned)swt_walk_row<true, 1>(a.wk, a.Nr, py, (p0 + mm) * a.f) * (unsigned)a.Nc;
                        st_lane(pA, ro, ox, acc0[mm].x);
                        st_lane(pH, ro, ox, acc0[mm].y);
                        st_lane(pV, ro, ox, acc1[mm].x);
                        st_lane(pD, ro, ox, acc1[mm].y);
                    }
                }
            } else {
                real_t* PDWT_RESTRICT o2 = a.out[0] + ob + 2 * ox;
                const real_t half = real_t(0.5);
#pragma unroll
                for (int mm = 0; mm < M; ++mm) {
                    if (p0 + mm < i0 + nm) {
                        const long long ro = (long long)swt_walk_row<true, 1>(a.wk, a.Nr, py, (p0 + mm) * a.f) * 2 * a.Nc;
                        real2_t v;
                        v.x = half * (acc0[mm].x + acc1[mm].x);
                        v.y = half * (acc0[mm].y + acc1[mm].y);
                        *reinterpret_cast<real2_t*>(o2 + ro) = v;
                    }
                }
            }
        }
    };

    // Two barriers per step: [carry_write, commit(t), issue(t + 1)] | [column pass(t), carry_read] |
    PDWT_FOR_THREADS(tid, NT) {
        stage_all(tid);
        make_plan(tid);
        if (T > 1) issue(tid);
    }
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            PDWT_FOR_THREADS(tid, NT) {
                carry_write(tid);
                commit(tid);
                if (t + 1 < T) issue(tid);
            }
        }
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) {
            col_pass(tid, t);
            if (t + 1 < T) carry_read(tid);
        }
        if (t + 1 < T) PDWT_LONG_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
// blockIdx.x -> (strip, chain x segment) through xcd_tile: an XCD gets a contiguous range, strips fastest
template <int HLEN, bool INV, int TXC, int TY, int NT, int M, int MINB>
__global__ void __launch_bounds__(NT, MINB) swt_colstream_kernel(const SwtColStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, cs;
    if (!xcd_tile(blockIdx.x, a.strips, a.segs * a.wk.phases, strip, cs)) return;
    const int py = cs / a.segs;
    swt_colstream_wg<HLEN, INV, TXC, TY, NT, M>(a, strip, py, cs - py * a.segs, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
