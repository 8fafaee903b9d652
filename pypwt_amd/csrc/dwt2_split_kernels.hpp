// dwt2_split_kernels.hpp -- one DECIMATED 2D level as a register-blocked ROW launch + COLUMN launch through scratch,
// for the filter lengths at which the fused LDS tiles of dwt2_fast_kernels.hpp stop paying (gfx950).
//
// Why.  A fused level kernel filters the rows of its tile's (2 TY + hlen - 2) input rows although it owns only 2 TY of
// them: at 40 taps and 64 x 32 tiles the row pass is done 1.6 times, the tile needs 67 KB of LDS (one or two workgroups
// of four wavefronts per CU) and one tile is a serial chain of ~13 us whatever the image size -- dwt2 db20 took 13 us
// per forward level and 18 us per inverse level from 128^2 to 1024^2 (profiles/r04c_ktimes_long.txt), 0.07 of the
// streaming rate at 2048^2.  Making tiles small enough to fill the chip multiplies the recomputation (8 output rows:
// 3.4 x).  The reference's own structure -- a row pass and a column pass through a temporary (pdwt/src/separable.cu:
// 179-209, 332-364) -- has no such trade-off: no pass recomputes anything, every row segment / column block is
// independent work, and for one image the temporary stays in the Infinity Cache.  What the reference does not do is
// block the passes in registers (it issues one load per tap and output, separable.cu:112-128):
//   row pass     a wavefront stages 1024 samples of a row (+ the window halo) in LDS with coalesced 16-B loads; a lane
//                owns 8 consecutive outputs (forward; 16 in the inverse) and streams its window of 16 + hlen - 2
//                samples past 8 stationary (lo, hi) accumulator pairs; results return through LDS as whole-row stores;
//   column pass  a lane owns four adjacent columns and R output rows: the 2 R + hlen - 2 (inverse: R / 2 + hlen / 2)
//                input rows stream past R x 4 stationary accumulator pairs -- (2 R + hlen - 2) 16-B loads per plane for
//                R hlen packed multiply-adds per column.  The wavefronts of a workgroup walk consecutive row blocks of
//                one column group, so the rows two blocks share come from L1 / the XCD's L2.
// Every multiply-add is a v_pk_fma_f32 with one half broadcast (packed_math.hpp: fma2_bx/by/tx/ty).
//
// Semantics (oracle/pdwt_oracle.c, separable.cu:91-176, 246-328; even hlen, even row and column counts):
//   analysis   out[k] = sum_j x[(2 k - c + j) mod N] f[hlen-1-j],  c = hlen/2 - 1
//   synthesis  out[g] = sum_{j < h2} a[(q - c2 + j) mod Nin] rlo[t] + d[...] rhi[t],  h2 = hlen/2, c2 = h2/2,
//              p = g + (h2 even), q = p / 2, t = hlen - 1 - (2 j + 1 - (p & 1))
//   forward  rows first: in -> lo, hi planes; then columns: lo -> (A, H), hi -> (V, D)
//   inverse  columns first: (A, H) -> t1, (V, D) -> t2, written interleaved (t1, t2) per column; then rows.
// fp32 only.
#pragma once

#include "packed_math.hpp"
#include "swt_split_kernels.hpp"

namespace pdwt {

struct DwtSplitArgs {
    const real_t* in[4];  // row fwd: in ; col fwd: lo, hi ; col inv: A, H, V, D ; row inv: interleaved (t1, t2)
    real_t* out[4];       // row fwd: lo, hi ; col fwd: A, H, V, D ; col inv: interleaved ; row inv: out
    int rows, cols;       // of the INPUT plane(s) of this pass (row inv: cols = coefficient columns = pairs per row)
    int batch;
    long long in_bstride, out_bstride;  // elements between the images of a batch, per plane as laid out
    FilterBankI t;                      // t[j] = (f_lo[hlen-1-j], f_hi[hlen-1-j]) of the pass's bank (dec / rec)
};

// ------------------------------------------------------------------------------------------------------------------
// Column passes: a WAVEFRONT owns 64 adjacent quad columns of (image, block of R output rows).
struct DwtColWork {
    int q, blk;
    long long bz;
    bool active;  // lanes beyond the last quad of a row load (clamped) and do not store
};
constexpr long long dwt_col_waves(int batch, int out_rows, int cols, int R) {
    return (long long)batch * ((out_rows + R - 1) / R) * (((cols >> 2) + 63) >> 6);
}
template <int R, int NT>
PDWT_DEVICE bool dwt_col_work(const DwtSplitArgs& a, int out_rows, long long block, int tid, DwtColWork& w) {
    const int QW = a.cols >> 2, QG = (QW + 63) >> 6;
    const unsigned blocks = (unsigned)(out_rows + R - 1) / R;
    const unsigned wave = (unsigned)block * (NT / 64) + (unsigned)PDWT_UNIFORM(tid >> 6);  // < 2^31 (launcher)
    if (wave >= (unsigned)a.batch * blocks * (unsigned)QG) return false;
    const unsigned t = wave / blocks;
    w.blk = (int)(wave - t * blocks);  // fastest: the wavefronts of a workgroup share their halo rows in L1 / L2
    const unsigned bz = t / (unsigned)QG;
    const int qg = (int)(t - bz * (unsigned)QG);
    w.bz = bz;
    const int q = qg * 64 + (tid & 63);
    w.active = q < QW;
    w.q = w.active ? q : QW - 1;
    return true;
}

// forward, column pass: lo, hi planes [rows][cols] -> A = Ly lo, H = Hy lo, V = Ly hi, D = Hy hi  [rows / 2][cols]
// CH: rows per chunk of loads; two chunks are in flight (one being consumed), and a wavefront's time is its
// NIN / CH chunks x max(load latency, the chunk's arithmetic): at 2048^2 there is ONE wavefront per SIMD, nothing else hides it
template <int HLEN, int R, int NT, int CH>
PDWT_DEVICE void dwt_col_fwd_tile(const DwtSplitArgs& a, long long block) {
    constexpr int c = HLEN / 2 - 1, NIN = 2 * (R - 1) + HLEN, NCH = (NIN + CH - 1) / CH;
    const int out_rows = a.rows >> 1;
    PDWT_FOR_THREADS(tid, NT) {
        DwtColWork w;
        if (!dwt_col_work<R, NT>(a, out_rows, block, tid, w)) continue;
        const real_t* PDWT_RESTRICT lo = a.in[0] + w.bz * a.in_bstride;
        const real_t* PDWT_RESTRICT hi = a.in[1] + w.bz * a.in_bstride;
        const int xq = 4 * w.q;
        const real_t zero = 0;
        v2f accAH[R][4], accVD[R][4];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) accAH[m][i] = accVD[m][i] = mk2(zero, zero);
        int y = true_mod(2 * w.blk * R - c, a.rows);  // wave-uniform
        rv4 bl[2][CH], bh[2][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                bl[0][u] = load4(lo + (long long)y * a.cols + xq);
                bh[0][u] = load4(hi + (long long)y * a.cols + xq);
                y = step_wrap(y, 1, a.rows);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    bl[(ch + 1) & 1][u] = load4(lo + (long long)y * a.cols + xq);
                    bh[(ch + 1) & 1][u] = load4(hi + (long long)y * a.cols + xq);
                    y = step_wrap(y, 1, a.rows);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - 2 * m;
                        if (j >= 0 && j < HLEN) {
                            const rv4 &vl = bl[ch & 1][u], &vh = bh[ch & 1][u];
                            accAH[m][0] = fma2_bx(mk2(vl.x, vl.y), a.t.t[j], accAH[m][0]);
                            accAH[m][1] = fma2_by(mk2(vl.x, vl.y), a.t.t[j], accAH[m][1]);
                            accAH[m][2] = fma2_bx(mk2(vl.z, vl.w), a.t.t[j], accAH[m][2]);
                            accAH[m][3] = fma2_by(mk2(vl.z, vl.w), a.t.t[j], accAH[m][3]);
                            accVD[m][0] = fma2_bx(mk2(vh.x, vh.y), a.t.t[j], accVD[m][0]);
                            accVD[m][1] = fma2_by(mk2(vh.x, vh.y), a.t.t[j], accVD[m][1]);
                            accVD[m][2] = fma2_bx(mk2(vh.z, vh.w), a.t.t[j], accVD[m][2]);
                            accVD[m][3] = fma2_by(mk2(vh.z, vh.w), a.t.t[j], accVD[m][3]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER_S(y, accAH[0][0], accAH[R - 1][0], accVD[0][0], accVD[R - 1][0]);
        }
        if (!w.active) continue;
        const long long ob = w.bz * a.out_bstride + xq;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int yo = w.blk * R + m;
            if (yo < out_rows) {
                const long long o = ob + (long long)yo * a.cols;
                store4(a.out[0] + o, mk4(accAH[m][0].x, accAH[m][1].x, accAH[m][2].x, accAH[m][3].x));
                store4(a.out[1] + o, mk4(accAH[m][0].y, accAH[m][1].y, accAH[m][2].y, accAH[m][3].y));
                store4(a.out[2] + o, mk4(accVD[m][0].x, accVD[m][1].x, accVD[m][2].x, accVD[m][3].x));
                store4(a.out[3] + o, mk4(accVD[m][0].y, accVD[m][1].y, accVD[m][2].y, accVD[m][3].y));
            }
        }
    }
}

// Synthesis geometry of R consecutive outputs starting at an EVEN index: output m uses window q_m = (m + S) / 2 and the
// taps of parity par_m; source r (counted from window 0's first source) is tap j = r - q_m of output m.
template <int HLEN>
struct SynGeom {
    static constexpr int H2 = HLEN / 2, C2 = H2 / 2, S = (H2 & 1) ? 0 : 1;
    static constexpr int q(int m) { return (m + S) >> 1; }
    static constexpr int par(int m) { return 1 - ((m + S) & 1); }
    static constexpr int nin(int R) { return q(R - 1) + H2; }  // sources needed by R outputs
};

// inverse, column pass: t1 = Ly A + Hy H, t2 = Ly V + Hy D  [2 rows][cols], written interleaved (t1, t2) per column.
// Packed over two adjacent columns (the natural halves of a loaded quad) with the tap broadcast.
template <int HLEN, int R, int NT, int CH>
PDWT_DEVICE void dwt_col_inv_tile(const DwtSplitArgs& a, long long block) {
    using G = SynGeom<HLEN>;
    static_assert(R % 2 == 0, "blocks of output rows start at even rows");
    constexpr int NIN = G::nin(R), NCH = (NIN + CH - 1) / CH;
    const int out_rows = 2 * a.rows;
    PDWT_FOR_THREADS(tid, NT) {
        DwtColWork w;
        if (!dwt_col_work<R, NT>(a, out_rows, block, tid, w)) continue;
        const long long ib = w.bz * a.in_bstride;
        const real_t* PDWT_RESTRICT pA = a.in[0] + ib;
        const real_t* PDWT_RESTRICT pH = a.in[1] + ib;
        const real_t* PDWT_RESTRICT pV = a.in[2] + ib;
        const real_t* PDWT_RESTRICT pD = a.in[3] + ib;
        const int xq = 4 * w.q;
        const real_t zero = 0;
        v2f acc1[R][2], acc2[R][2];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int k = 0; k < 2; ++k) acc1[m][k] = acc2[m][k] = mk2(zero, zero);
        int y = true_mod(w.blk * (R / 2) - G::C2, a.rows);  // wave-uniform: first source row of output row blk * R
        rv4 b[2][CH][4];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < NIN) {
                const long long o = (long long)y * a.cols + xq;
                b[0][u][0] = load4(pA + o);
                b[0][u][1] = load4(pH + o);
                b[0][u][2] = load4(pV + o);
                b[0][u][3] = load4(pD + o);
                y = step_wrap(y, 1, a.rows);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if ((ch + 1) * CH + u < NIN) {
                    const long long o = (long long)y * a.cols + xq;
                    b[(ch + 1) & 1][u][0] = load4(pA + o);
                    b[(ch + 1) & 1][u][1] = load4(pH + o);
                    b[(ch + 1) & 1][u][2] = load4(pV + o);
                    b[(ch + 1) & 1][u][3] = load4(pD + o);
                    y = step_wrap(y, 1, a.rows);
                }
            }
            PDWT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = ch * CH + u;
                if (r < NIN) {
                    const rv4 &vA = b[ch & 1][u][0], &vH = b[ch & 1][u][1], &vV = b[ch & 1][u][2], &vD = b[ch & 1][u][3];
#pragma unroll
                    for (int m = 0; m < R; ++m) {
                        const int j = r - G::q(m);
                        if (j >= 0 && j < G::H2) {
                            const v2f tp = a.t.t[2 * j + G::par(m)];  // (rlo[t], rhi[t]), t = hlen - 1 - (2 j + par)
                            acc1[m][0] = fma2_tx(mk2(vA.x, vA.y), tp, acc1[m][0]);
                            acc1[m][1] = fma2_tx(mk2(vA.z, vA.w), tp, acc1[m][1]);
                            acc1[m][0] = fma2_ty(mk2(vH.x, vH.y), tp, acc1[m][0]);
                            acc1[m][1] = fma2_ty(mk2(vH.z, vH.w), tp, acc1[m][1]);
                            acc2[m][0] = fma2_tx(mk2(vV.x, vV.y), tp, acc2[m][0]);
                            acc2[m][1] = fma2_tx(mk2(vV.z, vV.w), tp, acc2[m][1]);
                            acc2[m][0] = fma2_ty(mk2(vD.x, vD.y), tp, acc2[m][0]);
                            acc2[m][1] = fma2_ty(mk2(vD.z, vD.w), tp, acc2[m][1]);
                        }
                    }
                }
            }
            PDWT_SCHED_FENCE();
            PDWT_ORDER_AFTER_S(y, acc1[0][0], acc1[R - 1][0], acc2[0][0], acc2[R - 1][0]);
        }
        if (!w.active) continue;
        real_t* PDWT_RESTRICT o2 = a.out[0] + w.bz * a.out_bstride + 2 * xq;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int yo = w.blk * R + m;
            if (yo < out_rows) {
                real_t* o = o2 + (long long)yo * 2 * a.cols;
                store4(o, mk4(acc1[m][0].x, acc2[m][0].x, acc1[m][0].y, acc2[m][0].y));
                store4(o + 4, mk4(acc1[m][1].x, acc2[m][1].x, acc1[m][1].y, acc2[m][1].y));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Row passes through LDS (the scheme of swt_row_fwd_lds_tile / swt_row_inv_lds_tile at dilation 1, with stride-2 index
// relations): a wavefront owns kRowSpan = 1024 SAMPLES of one row -- forward: 1024 inputs -> 512 (lo, hi) outputs, a lane
// its 16 inputs' 8 outputs; inverse: 512 (t1, t2) pairs -> 1024 outputs, a lane 16 of them.
template <int HLEN>
struct DwtRowFwdLds {
    static constexpr int c = HLEN / 2 - 1, LEAD = (c + 3) / 4 * 4, OFF = LEAD - c;
    static constexpr int NQ = (LEAD + kRowSpan + (HLEN - 1 - c) + 3) / 4;  // staged quads per wavefront
    static constexpr int NWQ = (OFF + 14 + HLEN + 3) / 4;                  // window quads per lane: floats [0, OFF + 14 + HLEN)
    static constexpr int WAVE_FLOATS = (NQ + 3) / 4 * 20;
};
template <int HLEN>
struct DwtRowInvLds {
    using G = SynGeom<HLEN>;
    // staged unit: one (t1, t2) pair; a lane's 16 outputs use windows q = 8 lane + (0 .. 7 + S), sources q - C2 + (0 .. H2 - 1)
    static constexpr int LEADP = (G::C2 + 1) / 2 * 2, OFFP = LEADP - G::C2;  // pairs; the staging origin is a whole quad
    static constexpr int NWP = OFFP + G::nin(16);                             // window pairs per lane
    static constexpr int NWQ = (NWP + 1) / 2;
    static constexpr int NP = LEADP + kRowSpan / 2 + G::nin(16) - 8;          // staged pairs per wavefront
    static constexpr int NQ = (NP + 1) / 2;
    static constexpr int WAVE_FLOATS = (NQ + 3) / 4 * 20;
};
template <int HLEN>
constexpr int dwt_row_lds_floats(bool inverse, int NT) {
    return (NT / 64) * (inverse ? (DwtRowInvLds<HLEN>::WAVE_FLOATS > 64 * 20 ? DwtRowInvLds<HLEN>::WAVE_FLOATS : 64 * 20)
                                : DwtRowFwdLds<HLEN>::WAVE_FLOATS);
}

struct DwtRowWork {
    int y, xs, lane;  // xs: first SAMPLE (forward: input column, inverse: output column) of the wavefront's segment
    long long bz;
    real_t* reg;
    bool valid;
};
constexpr long long dwt_row_waves(int batch, int rows, int samples) { return (long long)batch * rows * ((samples + kRowSpan - 1) / kRowSpan); }
template <int NT>
PDWT_DEVICE DwtRowWork dwt_row_work(const DwtSplitArgs& a, int samples, long long block, int tid, real_t* smem, int wave_floats) {
    // 32-bit arithmetic (the launcher keeps the wavefront count below 2^31): a 64-bit division is ~100 scalar instructions,
    // and the CU's ONE scalar unit serves all its wavefronts -- with the decomposition recomputed in every phase the
    // 2048^2 row launch of 40 taps took 15 us, two thirds of it scalar division
    DwtRowWork w;
    const unsigned TG = (unsigned)(samples + kRowSpan - 1) / kRowSpan;
    const int wv = PDWT_UNIFORM(tid >> 6);
    const unsigned wave = (unsigned)block * (NT / 64) + (unsigned)wv;
    w.valid = wave < (unsigned)a.batch * (unsigned)a.rows * TG;
    const unsigned ty = wave / TG;
    w.xs = (int)(wave - ty * TG) * kRowSpan;
    const unsigned bz = ty / (unsigned)a.rows;
    w.y = (int)(ty - bz * (unsigned)a.rows);
    w.bz = bz;
    w.lane = tid & 63;
    w.reg = smem + wv * wave_floats;
    return w;
}

// forward: in [rows][cols] -> lo, hi [rows][cols / 2]
template <int HLEN, int NT>
PDWT_DEVICE void dwt_row_fwd_tile(const DwtSplitArgs& a, long long block, real_t* smem) {
    using G = DwtRowFwdLds<HLEN>;
    constexpr int CH = kSplitChunk, NCH = (G::NWQ + CH - 1) / CH;
    const int oc = a.cols >> 1;
    PDWT_PER_THREAD(DwtRowWork, wk, 1, NT);
    PDWT_PER_THREAD(v2f, acc, 8, NT);
    PDWT_FOR_THREADS(tid, NT) {  // phase 1: the row segment, coalesced, into the wavefront's region
        DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        w = dwt_row_work<NT>(a, a.cols, block, tid, smem, G::WAVE_FLOATS);  // once: kept in registers across the phases
        if (w.valid) {
            const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * a.cols;
#pragma unroll
            for (int k = 0; k < (G::NQ + 63) / 64; ++k) {
                const int qq = w.lane + 64 * k;
                if (qq < G::NQ) store4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3), load4(row + true_mod(w.xs - G::LEAD + 4 * qq, a.cols)));
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {  // phase 2: 8 outputs per lane from its window of 16 + hlen - 2 samples
        const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        v2f* acc = PDWT_MINE(acc, tid);
        const real_t zero = 0;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) acc[pp] = mk2(zero, zero);
        if (w.valid) {
            unsigned lbo = 20u * (unsigned)w.lane;
            rv4 b[2][CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (u < G::NWQ) b[0][u] = load4(w.reg + lbo + 20 * (u >> 2) + 4 * (u & 3));
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int n = (ch + 1) * CH + u;
                    if (n < G::NWQ) b[(ch + 1) & 1][u] = load4(w.reg + lbo + 20 * (n >> 2) + 4 * (n & 3));
                }
                PDWT_SCHED_FENCE();
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int wq = ch * CH + u;
                    if (wq < G::NWQ) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = 4 * wq + i;
#pragma unroll
                            for (int pp = 0; pp < 8; ++pp) {
                                const int d = k - G::OFF - 2 * pp;  // tap index of window sample k in output pp
                                if (d >= 0 && d < HLEN) {
                                    const rv4& v = b[ch & 1][u];
                                    const v2f pr = i < 2 ? mk2(v.x, v.y) : mk2(v.z, v.w);
                                    acc[pp] = (i & 1) ? fma2_by(pr, a.t.t[d], acc[pp]) : fma2_bx(pr, a.t.t[d], acc[pp]);
                                }
                            }
                        }
                    }
                }
                PDWT_SCHED_FENCE();
                PDWT_ORDER_AFTER(lbo, acc[0], acc[1], acc[6], acc[7]);
            }
        }
    }
    PDWT_SYNC();
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {  // phase 3: lo, then hi, through the region to whole-row stores
        PDWT_FOR_THREADS(tid, NT) {
            const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
            const v2f* acc = PDWT_MINE(acc, tid);
            if (w.valid) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    store4(w.reg + 8 * w.lane + 4 * m,
                           plane == 0 ? mk4(acc[4 * m].x, acc[4 * m + 1].x, acc[4 * m + 2].x, acc[4 * m + 3].x)
                                      : mk4(acc[4 * m].y, acc[4 * m + 1].y, acc[4 * m + 2].y, acc[4 * m + 3].y));
            }
        }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) {
            const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
            if (w.valid) {
                real_t* PDWT_RESTRICT orow = a.out[plane] + w.bz * a.out_bstride + (long long)w.y * oc;
#pragma unroll
                for (int k = 0; k < kRowSpan / 512; ++k) {
                    const int qq = w.lane + 64 * k, col = (w.xs >> 1) + 4 * qq;
                    if (col < oc) store4(orow + col, load4(w.reg + 4 * qq));
                }
            }
        }
        PDWT_SYNC();
    }
}

// inverse: interleaved (t1, t2) [rows][cols pairs] -> out [rows][2 cols]
template <int HLEN, int NT>
PDWT_DEVICE void dwt_row_inv_tile(const DwtSplitArgs& a, long long block, real_t* smem) {
    using L = DwtRowInvLds<HLEN>;
    using G = SynGeom<HLEN>;
    constexpr int WF = L::WAVE_FLOATS > 64 * 20 ? L::WAVE_FLOATS : 64 * 20;
    constexpr int CH = kSplitChunk, NCH = (L::NWQ + CH - 1) / CH;
    const int oc = 2 * a.cols;
    PDWT_PER_THREAD(DwtRowWork, wk, 1, NT);
    PDWT_PER_THREAD(v2f, acc, 16, NT);
    PDWT_FOR_THREADS(tid, NT) {  // phase 1: the wavefront's 512 + halo source pairs, two pairs per 16-B load
        DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        w = dwt_row_work<NT>(a, oc, block, tid, smem, WF);  // once: kept in registers across the phases
        if (w.valid) {
            const real_t* PDWT_RESTRICT row = a.in[0] + w.bz * a.in_bstride + (long long)w.y * 2 * a.cols;
#pragma unroll
            for (int k = 0; k < (L::NQ + 63) / 64; ++k) {
                const int qq = w.lane + 64 * k;  // pairs 2 qq, 2 qq + 1 of the staged range (cols is even: a quad never straddles)
                if (qq < L::NQ)
                    store4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3), load4(row + 2 * true_mod((w.xs >> 1) - L::LEADP + 2 * qq, a.cols)));
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {  // phase 2: 16 outputs per lane; staged pair 8 lane + k' is source k' - OFFP of window 0
        const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        v2f* acc = PDWT_MINE(acc, tid);
        const real_t zero = 0;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) acc[pp] = mk2(zero, zero);
        if (w.valid) {
            unsigned lbo = 20u * (unsigned)w.lane;  // the lane's 8 pairs = 16 floats = one padded block
            rv4 b[2][CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (u < L::NWQ) b[0][u] = load4(w.reg + lbo + 20 * (u >> 2) + 4 * (u & 3));
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int n = (ch + 1) * CH + u;
                    if (n < L::NWQ) b[(ch + 1) & 1][u] = load4(w.reg + lbo + 20 * (n >> 2) + 4 * (n & 3));
                }
                PDWT_SCHED_FENCE();
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int wq = ch * CH + u;
                    if (wq < L::NWQ) {
                        const rv4& v = b[ch & 1][u];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int r = 2 * wq + i - L::OFFP;  // source index counted from window 0's first source
                            const v2f pr = i == 0 ? mk2(v.x, v.y) : mk2(v.z, v.w);
#pragma unroll
                            for (int pp = 0; pp < 16; ++pp) {
                                const int j = r - G::q(pp);
                                if (j >= 0 && j < G::H2) acc[pp] = fma2_s(pr, a.t.t[2 * j + G::par(pp)], acc[pp]);
                            }
                        }
                    }
                }
                PDWT_SCHED_FENCE();
                PDWT_ORDER_AFTER(lbo, acc[0], acc[1], acc[14], acc[15]);
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        const v2f* acc = PDWT_MINE(acc, tid);
        if (w.valid) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                store4(w.reg + 20 * w.lane + 4 * m, mk4(acc[4 * m].x + acc[4 * m].y, acc[4 * m + 1].x + acc[4 * m + 1].y,
                                                        acc[4 * m + 2].x + acc[4 * m + 2].y, acc[4 * m + 3].x + acc[4 * m + 3].y));
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        const DwtRowWork& w = PDWT_MINE(wk, tid)[0];
        if (w.valid) {
            real_t* PDWT_RESTRICT orow = a.out[0] + w.bz * a.out_bstride + (long long)w.y * oc;
#pragma unroll
            for (int k = 0; k < kRowSpan / 256; ++k) {
                const int qq = w.lane + 64 * k, col = w.xs + 4 * qq;
                if (col < oc) store4(orow + col, load4(w.reg + 20 * (qq >> 2) + 4 * (qq & 3)));
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int R, int NT, int CH>
__global__ void __launch_bounds__(NT) dwt_col_fwd_kernel(const DwtSplitArgs a) { dwt_col_fwd_tile<HLEN, R, NT, CH>(a, blockIdx.x); }
template <int HLEN, int R, int NT, int CH>
__global__ void __launch_bounds__(NT) dwt_col_inv_kernel(const DwtSplitArgs a) { dwt_col_inv_tile<HLEN, R, NT, CH>(a, blockIdx.x); }
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt_row_fwd_kernel(const DwtSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    dwt_row_fwd_tile<HLEN, NT>(a, blockIdx.x, pdwt_smem);
}
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt_row_inv_kernel(const DwtSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    dwt_row_inv_tile<HLEN, NT>(a, blockIdx.x, pdwt_smem);
}
#endif

}  // namespace pdwt
