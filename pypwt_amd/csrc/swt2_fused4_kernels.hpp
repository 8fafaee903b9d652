// swt2_fused4_kernels.hpp -- TWO levels of a 2D stationary (a-trous) transform with a 4-TAP filter bank (db2, sym2 and
// every custom 4-tap bank) in ONE launch, one wavefront per strip, everything in registers (gfx950).
//
// The reference's documentation example is exactly this transform (doc/denoising.rst:85: Wavelets(img, "db2", 3,
// do_swt=1)); level per launch it moves 5 planes per level and direction (reference: pdwt/src/separable.cu:409-537,
// :553-672).  swt2_fused_kernels.hpp fuses the 2-tap banks, whose windows are one-sided; a 4-tap window is two-sided
// (analysis: columns / rows g - d .. g + 2 d, synthesis: g - 2 d .. g + d), so the rings are 4 d rows deep instead of
// 2 d and a strip needs halo lanes on BOTH sides.  Two levels (dilations f0 and 2 f0) fit the register file:
// 8 planes instead of 10 per pair and direction, one launch instead of two.
//
// Scheme (forward), same walk as the 2-tap kernels: a wavefront owns a strip of columns (a lane owns 4: one 16-B load
// per row) and walks DOWN the rows of ONE dilation phase of f0.  Per input row (phase row p):
//   * level A (dilation 1 phase row, f0 columns): the row filter takes the lane's four columns and their copies
//     f0 to the left, f0 and 2 f0 to the right (DPP lane shifts); the filtered (L, H) row enters a ring of 4 rows; with
//     rows p - 3 .. p in the ring the column filter emits level A's row p - 2: H, V, D stored, A kept in registers;
//   * level B (2 phase rows, 2 f0 columns) does the same on that A row with a ring of 8 rows and emits its row
//     (p - 2) - 4: all four bands stored.
// A segment of S owned rows therefore walks S + 9 rows (3 before, 6 after).  Lanes: 3 f0 halo columns on the left,
// 6 f0 on the right; rows and lanes a wavefront does not own carry a store offset beyond the plane's buffer
// descriptor and are dropped by the hardware (no branch around a store).
//
// Arithmetic (oracle/pdwt_oracle.c): analysis out[g] = sum_j x[g + (j - 1) d] f[3 - j]; synthesis
// out[g] = 0.5 sum_j (a[g + (j - 2) d] rlo[3 - j] + b[g + (j - 2) d] rhi[3 - j]); periodic in both directions.
#pragma once

#include "swt2_fused_kernels.hpp"  // shifts, swt_plane, row offsets, wave numbering

namespace pdwt {

struct Swt4Args {
    const real_t* in;   // forward: A_{l0-1}; inverse: A_{l0+1}
    real_t* out;        // forward: A_{l0+1}; inverse: A_{l0-1}
    real_t* H[2];       // details of the pair's levels, finer level first
    real_t* V[2];
    real_t* D[2];
    int Nr, Nc;
    long long bstride;
    int strips, segs, seg_rows;  // seg_rows: phase rows a wavefront owns (multiple of 8)
    real_t beta[2];     // inverse: soft threshold of each level's details (0 = none)
    real_t lo[4], hi[4];
    SwtWalk wk;         // swt_walk(Nr, Nc, f0, 4): planes of any size (swt2_fused_kernels.hpp)
};

template <int F0>
struct Swt4Geom {
    static_assert(F0 == 1 || F0 == 4, "levels 1-2 or 3-4 (whole-lane dilations)");
    static constexpr int DA = F0, DB = 2 * F0;                 // column dilations of the two levels
    // forward: level A reaches DA left / 2 DA right, level B the same with DB
    static constexpr int fwd_left = (DA + DB + 3) / 4, fwd_right = (2 * (DA + DB) + 3) / 4;
    // inverse: both levels reach 2 D left / D right
    static constexpr int inv_left = (2 * (DA + DB) + 3) / 4, inv_right = (DA + DB + 3) / 4;
    static constexpr int Vf = 64 - fwd_left - fwd_right;
    static constexpr int Vi = 64 - inv_left - inv_right;
    static constexpr int P = 8;   // rows per unrolled group: both ring periods (4, 8) and the load slots divide it
    static constexpr int NR = 4;  // input rows in flight
    static constexpr int Wf_before = 3, Wf_after = 6;          // forward: rows walked in front of / behind the owned ones
    static constexpr int Wi_before = 6, Wi_after = 3;
};

// ---------------------------------------------------------------------------------------------- forward
template <int F0>
struct Swt4FwdState {
    using G = Swt4Geom<F0>;
    WaveReg<real_t, 4 * G::NR> ld;
    WaveReg<real_t, 8 * 4> ringA;   // (L, H) rows of level A: [slot][L0..3 H0..3]
    WaveReg<real_t, 8 * 8> ringB;
    WaveReg<unsigned, 2> off;       // byte offsets in a row: load (wrapped), store (or kSwtLaneDropped)
    RowBuf bH[2], bV[2], bD[2], bA;
};

// one 4-tap analysis level on input row `ain` (ring slot SLOT of RD = 4 x the level's row dilation): emits the level's
// row 2 dilations earlier: details through the descriptors at `rowoff`, the approximation into `anext`
template <int D, int RD, int SLOT>
PDWT_DEVICE void swt4_fwd_level(const Swt4Args& a, WaveReg<real_t, 4>& ain, WaveReg<real_t, 8 * RD>& ring, WaveReg<real_t, 4>& anext,
                                WaveReg<unsigned, 2>& off, const RowBuf& bH, const RowBuf& bV, const RowBuf& bD, unsigned rowoff) {
    constexpr int RDIL = RD / 4;
    constexpr int S3 = (SLOT - 3 * RDIL + 4 * RD) % RD, S2 = (SLOT - 2 * RDIL + 4 * RD) % RD, S1 = (SLOT - RDIL + 4 * RD) % RD;
    WaveReg<real_t, 4> xl, xr1, xr2;
    swt_shift_left<D, 4>(ain, xl);
    swt_shift_right<D>(ain, xr1);
    swt_shift_right<2 * D>(ain, xr2);
    PDWT_WAVE_LANES(lane) {
        const real_t* x = ain.mine(lane);
        const real_t *l1 = xl.mine(lane), *r1 = xr1.mine(lane), *r2 = xr2.mine(lane);
        real_t* cur = ring.mine(lane) + 8 * SLOT;
        // row filter: out[g] = x[g - d] f[3] + x[g] f[2] + x[g + d] f[1] + x[g + 2 d] f[0]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            real_t L = l1[c] * a.lo[3], H = l1[c] * a.hi[3];
            L = pdwt_fma(x[c], a.lo[2], L);   H = pdwt_fma(x[c], a.hi[2], H);
            L = pdwt_fma(r1[c], a.lo[1], L);  H = pdwt_fma(r1[c], a.hi[1], H);
            L = pdwt_fma(r2[c], a.lo[0], L);  H = pdwt_fma(r2[c], a.hi[0], H);
            cur[c] = L;
            cur[4 + c] = H;
        }
        // column filter over ring rows t - 3 d (tap f[3]) .. t (tap f[0]): the level's row t - 2 d
        const real_t* q3 = ring.mine(lane) + 8 * S3;
        const real_t* q2 = ring.mine(lane) + 8 * S2;
        const real_t* q1 = ring.mine(lane) + 8 * S1;
        real_t* an = anext.mine(lane);
        real_t h[4], v[4], d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            real_t A = q3[c] * a.lo[3], Hh = q3[c] * a.hi[3], Vv = q3[4 + c] * a.lo[3], Dd = q3[4 + c] * a.hi[3];
            A = pdwt_fma(q2[c], a.lo[2], A);   Hh = pdwt_fma(q2[c], a.hi[2], Hh);
            Vv = pdwt_fma(q2[4 + c], a.lo[2], Vv); Dd = pdwt_fma(q2[4 + c], a.hi[2], Dd);
            A = pdwt_fma(q1[c], a.lo[1], A);   Hh = pdwt_fma(q1[c], a.hi[1], Hh);
            Vv = pdwt_fma(q1[4 + c], a.lo[1], Vv); Dd = pdwt_fma(q1[4 + c], a.hi[1], Dd);
            A = pdwt_fma(cur[c], a.lo[0], A);  Hh = pdwt_fma(cur[c], a.hi[0], Hh);
            Vv = pdwt_fma(cur[4 + c], a.lo[0], Vv); Dd = pdwt_fma(cur[4 + c], a.hi[0], Dd);
            an[c] = A; h[c] = Hh; v[c] = Vv; d[c] = Dd;
        }
        const unsigned o = off.mine(lane)[1] + rowoff;
        row_st16(bH, o, h[0], h[1], h[2], h[3]);
        row_st16(bV, o, v[0], v[1], v[2], v[3]);
        row_st16(bD, o, d[0], d[1], d[2], d[3]);
    }
}

// step R of a group of 8 rows: walk row w = g0 + R is phase row i0 - 3 + w
template <int F0, bool GEN, int R>
PDWT_DEVICE void swt4_fwd_step(const Swt4Args& a, Swt4FwdState<F0>& st, const real_t* in, int g0, int i0, int rows_phase, int py) {
    using G = Swt4Geom<F0>;
    const int w = g0 + R;
    const int last = a.seg_rows + G::Wf_before + G::Wf_after - 1;  // last walk row anybody needs
    {   // request walk row w + NR - 1 (its slot was consumed at the previous step); never past the last needed row
        int ww = w + G::NR - 1;
        ww = ww < last ? ww : last;
        int p = i0 - G::Wf_before + ww;
        p = ((p % rows_phase) + rows_phase) % rows_phase;
        const real_t* row = in + (long long)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, p) * a.Nc;
        PDWT_WAVE_LANES(lane) {
            const v4f q = swt_ld16<GEN>(row, st.off.mine(lane)[0]);
            real_t* v = st.ld.mine(lane) + 4 * ((R + G::NR - 1) % G::NR);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
    }
    PDWT_ROW_FENCE();
    WaveReg<real_t, 4> a0, a1, a2;
    PDWT_WAVE_LANES(lane) {
        const real_t* v = st.ld.mine(lane) + 4 * (R % G::NR);
#pragma unroll
        for (int c = 0; c < 4; ++c) a0.mine(lane)[c] = v[c];
    }
    auto rowoff = [&](int rel) -> unsigned {  // owned row `rel` of the segment -> byte offset of that row in a plane
        const bool ow = rel >= 0 && rel < a.seg_rows && i0 + rel < rows_phase;
        return ow ? kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, i0 + rel) * (unsigned)a.Nc : kSwtRowDropped;
    };
    // level A: input row w (ring slot w mod 4) emits its row w - 2, i.e. owned row w - 2 - 3
    swt4_fwd_level<G::DA, 4, R % 4>(a, a0, st.ringA, a1, st.off, st.bH[0], st.bV[0], st.bD[0], rowoff(w - 5));
    // level B: input = level A's row (walk index w - 2: ring slot (R + 6) mod 8) emits its row w - 6, owned row w - 9
    {
        const unsigned ro = rowoff(w - 9);
        swt4_fwd_level<G::DB, 8, (R + 6) % 8>(a, a1, st.ringB, a2, st.off, st.bH[1], st.bV[1], st.bD[1], ro);
        PDWT_WAVE_LANES(lane) { const real_t* v = a2.mine(lane); row_st16(st.bA, st.off.mine(lane)[1] + ro, v[0], v[1], v[2], v[3]); }
    }
}

template <int F0, bool GEN, int R>
PDWT_DEVICE void swt4_fwd_group(const Swt4Args& a, Swt4FwdState<F0>& st, const real_t* in, int g0, int i0, int rows_phase, int py) {
    if constexpr (R < Swt4Geom<F0>::P) {
        swt4_fwd_step<F0, GEN, R>(a, st, in, g0, i0, rows_phase, py);
        swt4_fwd_group<F0, GEN, R + 1>(a, st, in, g0, i0, rows_phase, py);
    }
}

// wavefront `w` of the launch: (image, phase, segment, strip)
template <int F0, bool GEN = false>
PDWT_DEVICE void swt4_fwd_fused(const Swt4Args& a, long long w) {
    using G = Swt4Geom<F0>;
    const int strip = (int)(w % a.strips);
    long long t = w / a.strips;
    const int seg = (int)(t % a.segs);
    t /= a.segs;
    const int phases = GEN ? a.wk.phases : F0;
    const int py = (int)(t % phases);
    const long long img = t / phases;
    const int rows_phase = GEN ? a.wk.rows_phase : a.Nr / F0;
    const int i0 = seg * a.seg_rows;
    const long long boff = img * a.bstride;
    const real_t* in = a.in + boff;
    Swt4FwdState<F0> st;
    PDWT_WAVE_LANES(lane) {
        const int x = (GEN ? swt_strip_x0(a.wk, strip, 4 * G::Vf) : strip * 4 * G::Vf) + 4 * (lane - G::fwd_left);
        // halo columns wrap periodically (a group never straddles: Nc % 4 == 0 or SwtWalk::pad); lanes far right of the row end re-read
        // the row's last group (nothing of theirs is used)
        int xl = x < 0 ? x + a.Nc : (x >= a.Nc ? x - a.Nc : x);
        if (x >= a.Nc + 4 * G::fwd_right) xl = a.Nc - 4;
        st.off.mine(lane)[0] = kRealBytes * (unsigned)xl;
        st.off.mine(lane)[1] = (lane >= G::fwd_left && lane < G::fwd_left + G::Vf && x < a.Nc) ? kRealBytes * (unsigned)x : kSwtLaneDropped;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        st.bH[k] = swt_plane(a.H[k], boff, a.Nr, a.Nc);
        st.bV[k] = swt_plane(a.V[k], boff, a.Nr, a.Nc);
        st.bD[k] = swt_plane(a.D[k], boff, a.Nr, a.Nc);
    }
    st.bA = swt_plane(a.out, boff, a.Nr, a.Nc);
    // walk rows 0 .. NR-2 in flight before the first step
#pragma unroll
    for (int p = 0; p < G::NR - 1; ++p) {
        int pr = i0 - G::Wf_before + p;
        pr = ((pr % rows_phase) + rows_phase) % rows_phase;
        const real_t* row = in + (long long)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, pr) * a.Nc;
        PDWT_WAVE_LANES(lane) {
            const v4f q = swt_ld16<GEN>(row, st.off.mine(lane)[0]);
            real_t* v = st.ld.mine(lane) + 4 * p;
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
    }
    PDWT_WAIT_VMEM();
    // seg_rows + 9 walk rows in groups of 8 (seg_rows is a multiple of 8: two extra groups cover the 9)
    const int ngroups = a.seg_rows / G::P + 2;
    for (int g = 0; g < ngroups; ++g) swt4_fwd_group<F0, GEN, 0>(a, st, in, g * G::P, i0, rows_phase, py);
}


// ---------------------------------------------------------------------------------------------- inverse
// The pair undone in one launch: level B (dilation 2 phase rows / 2 f0 columns) first, level A on its output.  Per walk
// step (phase row p of the level-B planes) a wavefront loads A, H, V, D of level B at row p and H, V, D of level A at
// row p - 2 (the row of level A's approximation that level B emits in the same step), NRI - 1 steps ahead; level B's
// row synthesis goes into a ring of 8 (u1, u2) rows, its column synthesis emits the approximation row p - 2; level A
// does the same with a ring of 4 and emits output row p - 3.  A segment of S owned rows walks S + 9 rows (6 before,
// 3 after); level A's details are only used from walk row 6 on: before that their loads are redirected to the row of
// the approximation plane the same step loads anyway (pointer select, no branch around a load).  A pending soft
// threshold is applied to the details as they are loaded.
template <int F0, int NRI_>
struct Swt4InvState {
    static constexpr int NRI = NRI_;
    WaveReg<real_t, 4 * NRI_ * 7> ld;   // [slot][plane][4]: 0 = A_B, 1..3 = H, V, D of level B, 4..6 = H, V, D of level A
    WaveReg<real_t, 8 * 8> ringB;       // (u1, u2) rows: [slot][u1 0..3 | u2 0..3]
    WaveReg<real_t, 8 * 4> ringA;
    WaveReg<unsigned, 2> off;
    RowBuf bo;
};

template <int F0, int NRI, bool GEN, int SLOT>
PDWT_DEVICE void swt4_inv_load(const Swt4Args& a, Swt4InvState<F0, NRI>& st, long long boff, int ww, int i0, int rows_phase, int py) {
    using G = Swt4Geom<F0>;
    int pB = i0 - G::Wi_before + ww, pA = pB - 2;
    pB = ((pB % rows_phase) + rows_phase) % rows_phase;
    pA = ((pA % rows_phase) + rows_phase) % rows_phase;
    const unsigned roB = kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, pB) * (unsigned)a.Nc;
    const bool usedA = ww >= G::Wi_before;  // uniform
    const unsigned roA = usedA ? kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, pA) * (unsigned)a.Nc : roB;
    const real_t* pl[7] = {a.in, a.H[1], a.V[1], a.D[1], usedA ? a.H[0] : a.in, usedA ? a.V[0] : a.in, usedA ? a.D[0] : a.in};
    PDWT_WAVE_LANES(lane) {
        real_t* base = st.ld.mine(lane) + 4 * 7 * SLOT;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const v4f w4 = swt_ld16<GEN>(pl[q] + boff, st.off.mine(lane)[0] + (q < 4 ? roB : roA));
            base[4 * q] = w4.x; base[4 * q + 1] = w4.y; base[4 * q + 2] = w4.z; base[4 * q + 3] = w4.w;
        }
    }
}

// one 4-tap synthesis level on the current row: ain (approximation row) + det = [H | V | D] (thresholded here) ->
// (u1, u2) into ring slot SLOT of RD = 4 x the level's row dilation; emits the level's row one dilation earlier into aout
template <int D, int RD, int SLOT>
PDWT_DEVICE void swt4_inv_level(const Swt4Args& a, WaveReg<real_t, 4>& ain, WaveReg<real_t, 12>& det, real_t beta,
                                WaveReg<real_t, 8 * RD>& ring, WaveReg<real_t, 4>& aout) {
    constexpr int RDIL = RD / 4;
    constexpr int S3 = (SLOT - 3 * RDIL + 4 * RD) % RD, S2 = (SLOT - 2 * RDIL + 4 * RD) % RD, S1 = (SLOT - RDIL + 4 * RD) % RD;
    // contribution of source column s to output column s + 2 d (tap [3]), s + d ([2]), s ([1]), s - d ([0])
    WaveReg<real_t, 4> p3, p2, p0, q3, q2, q0, sp3, sp2, sp0, sq3, sq2, sq0;
    WaveReg<real_t, 8> mid;  // tap [1] terms of u1 | u2
    PDWT_WAVE_LANES(lane) {
        const real_t* x = ain.mine(lane);
        real_t* dd = det.mine(lane);
#pragma unroll
        for (int i = 0; i < 12; ++i) dd[i] = soft_shrink(dd[i], beta);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const real_t A = x[c], Hh = dd[c], Vv = dd[4 + c], Dd = dd[8 + c];
            p3.mine(lane)[c] = pdwt_fma(Vv, a.hi[3], A * a.lo[3]);  q3.mine(lane)[c] = pdwt_fma(Dd, a.hi[3], Hh * a.lo[3]);
            p2.mine(lane)[c] = pdwt_fma(Vv, a.hi[2], A * a.lo[2]);  q2.mine(lane)[c] = pdwt_fma(Dd, a.hi[2], Hh * a.lo[2]);
            mid.mine(lane)[c] = pdwt_fma(Vv, a.hi[1], A * a.lo[1]); mid.mine(lane)[4 + c] = pdwt_fma(Dd, a.hi[1], Hh * a.lo[1]);
            p0.mine(lane)[c] = pdwt_fma(Vv, a.hi[0], A * a.lo[0]);  q0.mine(lane)[c] = pdwt_fma(Dd, a.hi[0], Hh * a.lo[0]);
        }
    }
    swt_shift_left<2 * D, 4>(p3, sp3);
    swt_shift_left<2 * D, 4>(q3, sq3);
    swt_shift_left<D, 4>(p2, sp2);
    swt_shift_left<D, 4>(q2, sq2);
    swt_shift_right<D>(p0, sp0);
    swt_shift_right<D>(q0, sq0);
    PDWT_WAVE_LANES(lane) {
        real_t* cur = ring.mine(lane) + 8 * SLOT;
        const real_t* r3 = ring.mine(lane) + 8 * S3;
        const real_t* r2 = ring.mine(lane) + 8 * S2;
        const real_t* r1 = ring.mine(lane) + 8 * S1;
        real_t* o = aout.mine(lane);
        const real_t half = real_t(0.5);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cur[c] = half * ((sp3.mine(lane)[c] + sp2.mine(lane)[c]) + (mid.mine(lane)[c] + sp0.mine(lane)[c]));
            cur[4 + c] = half * ((sq3.mine(lane)[c] + sq2.mine(lane)[c]) + (mid.mine(lane)[4 + c] + sq0.mine(lane)[c]));
            real_t r = r3[c] * a.lo[3];
            r = pdwt_fma(r3[4 + c], a.hi[3], r);
            r = pdwt_fma(r2[c], a.lo[2], r);
            r = pdwt_fma(r2[4 + c], a.hi[2], r);
            r = pdwt_fma(r1[c], a.lo[1], r);
            r = pdwt_fma(r1[4 + c], a.hi[1], r);
            r = pdwt_fma(cur[c], a.lo[0], r);
            r = pdwt_fma(cur[4 + c], a.hi[0], r);
            o[c] = half * r;
        }
    }
}

template <int F0, int NRI, bool GEN, int R>
PDWT_DEVICE void swt4_inv_step(const Swt4Args& a, Swt4InvState<F0, NRI>& st, int g0, int i0, int rows_phase, int py, long long boff) {
    using G = Swt4Geom<F0>;
    const int w = g0 + R;
    const int last = a.seg_rows + G::Wi_before + G::Wi_after - 1;
    {
        int ww = w + NRI - 1;
        ww = ww < last ? ww : last;
        swt4_inv_load<F0, NRI, GEN, (R + NRI - 1) % NRI>(a, st, boff, ww, i0, rows_phase, py);
    }
    PDWT_ROW_FENCE();
    WaveReg<real_t, 4> cur, mid, res;
    WaveReg<real_t, 12> det;
    PDWT_WAVE_LANES(lane) {
        const real_t* v = st.ld.mine(lane) + 4 * 7 * (R % NRI);
#pragma unroll
        for (int c = 0; c < 4; ++c) cur.mine(lane)[c] = v[c];
#pragma unroll
        for (int i = 0; i < 12; ++i) det.mine(lane)[i] = v[4 + i];
    }
    // level B on walk row w (ring slot w mod 8) emits the approximation of level A at walk row w - 2
    swt4_inv_level<G::DB, 8, R % 8>(a, cur, det, a.beta[1], st.ringB, mid);
    PDWT_WAVE_LANES(lane) {
        const real_t* v = st.ld.mine(lane) + 4 * 7 * (R % NRI) + 16;
#pragma unroll
        for (int i = 0; i < 12; ++i) det.mine(lane)[i] = v[i];
    }
    // level A on that row (ring slot (w - 2) mod 4) emits output row (w - 2) - 1: owned row w - 9
    swt4_inv_level<G::DA, 4, (R + 2) % 4>(a, mid, det, a.beta[0], st.ringA, res);
    const int rel = w - (G::Wi_before + 3);
    const bool ow = rel >= 0 && rel < a.seg_rows && i0 + rel < rows_phase;
    const unsigned ro = ow ? kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, i0 + rel) * (unsigned)a.Nc : kSwtRowDropped;
    PDWT_WAVE_LANES(lane) {
        const real_t* v = res.mine(lane);
        row_st16(st.bo, st.off.mine(lane)[1] + ro, v[0], v[1], v[2], v[3]);
    }
}

template <int F0, int NRI, bool GEN, int R>
PDWT_DEVICE void swt4_inv_group(const Swt4Args& a, Swt4InvState<F0, NRI>& st, int g0, int i0, int rows_phase, int py, long long boff) {
    if constexpr (R < Swt4Geom<F0>::P) {
        swt4_inv_step<F0, NRI, GEN, R>(a, st, g0, i0, rows_phase, py, boff);
        swt4_inv_group<F0, NRI, GEN, R + 1>(a, st, g0, i0, rows_phase, py, boff);
    }
}

template <int F0, int NRI, bool GEN, int I>
PDWT_DEVICE void swt4_inv_preload(const Swt4Args& a, Swt4InvState<F0, NRI>& st, long long boff, int i0, int rows_phase, int py) {
    if constexpr (I < NRI - 1) {
        swt4_inv_load<F0, NRI, GEN, I>(a, st, boff, I, i0, rows_phase, py);
        swt4_inv_preload<F0, NRI, GEN, I + 1>(a, st, boff, i0, rows_phase, py);
    }
}

template <int F0, int NRI, bool GEN = false>
PDWT_DEVICE void swt4_inv_fused(const Swt4Args& a, long long w) {
    using G = Swt4Geom<F0>;
    const int strip = (int)(w % a.strips);
    long long t = w / a.strips;
    const int seg = (int)(t % a.segs);
    t /= a.segs;
    const int phases = GEN ? a.wk.phases : F0;
    const int py = (int)(t % phases);
    const long long img = t / phases;
    const int rows_phase = GEN ? a.wk.rows_phase : a.Nr / F0;
    const int i0 = seg * a.seg_rows;
    const long long boff = img * a.bstride;
    Swt4InvState<F0, NRI> st;
    PDWT_WAVE_LANES(lane) {
        const int x = (GEN ? swt_strip_x0(a.wk, strip, 4 * G::Vi) : strip * 4 * G::Vi) + 4 * (lane - G::inv_left);
        int xl = x < 0 ? x + a.Nc : (x >= a.Nc ? x - a.Nc : x);
        if (x >= a.Nc + 4 * G::inv_right) xl = a.Nc - 4;
        st.off.mine(lane)[0] = kRealBytes * (unsigned)xl;
        st.off.mine(lane)[1] = (lane >= G::inv_left && lane < G::inv_left + G::Vi && x < a.Nc) ? kRealBytes * (unsigned)x : kSwtLaneDropped;
    }
    st.bo = swt_plane(a.out, boff, a.Nr, a.Nc);
    swt4_inv_preload<F0, NRI, GEN, 0>(a, st, boff, i0, rows_phase, py);
    PDWT_WAIT_VMEM();
    const int ngroups = a.seg_rows / G::P + 2;
    for (int g = 0; g < ngroups; ++g) swt4_inv_group<F0, NRI, GEN, 0>(a, st, g * G::P, i0, rows_phase, py, boff);
}

#ifndef PDWT_CPU_EMU
template <int F0, int NT, bool GEN>
__global__ void __launch_bounds__(NT, 1) swt4_fwd_fused_kernel(const Swt4Args a, long long waves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = swt_fused_wave(blockIdx.x, NT / 64, wave, waves);
    if (w < waves) swt4_fwd_fused<F0, GEN>(a, w);
}
template <int F0, int NRI, int NT, bool GEN>
__global__ void __launch_bounds__(NT, 1) swt4_inv_fused_kernel(const Swt4Args a, long long waves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = swt_fused_wave(blockIdx.x, NT / 64, wave, waves);
    if (w < waves) swt4_inv_fused<F0, NRI, GEN>(a, w);
}
#endif

}  // namespace pdwt
