// dwt2_ring_kernels.hpp -- 2D DWT level kernels for LONG filters (10 taps and more) on gfx950: one wavefront per
// tile, the per-row halo staged in LDS, the column filter a register ring of running sums.
//
// Replaces, for large levels, the LDS-tiled level kernels of dwt2_fast_kernels.hpp (reference: w_kern_forward_pass1/2,
// pdwt/src/separable.cu:91-176, and w_kern_inverse_pass1/2, :246-328, which treat every hlen <= 40 alike).  What
// rocprofv3 said about the 16-tap tile (profiles/r04c_planprof_sym8.txt): 338 vector instructions per wavefront and tile of
// which 156 are the packed multiply-adds -- the rest is staging index arithmetic, and the (hlen-2)-row halo of every tile
// is filtered along x again by the tile below; 60 % VALU-busy at 0.56 of the HBM peak with 1.0x traffic.  Here
//   * a wavefront walks DOWN a strip of 64 CPL image columns (CPL = 2 or 4 columns per lane): one coalesced load per image
//     row, NR-1 rows ahead, into a register ring (dwt2_wave_kernels.hpp); one extra 4-B load per lane fetches the
//     2 HL halo columns left and right of the strip (periodic wrap = a per-lane offset computed once);
//   * the ROW filter needs hlen-2 samples of the neighbouring lanes: the row goes through a wavefront-private LDS row
//     (one wide ds_write + one 4-B halo write, then NV/CPL wide ds_reads per lane) -- no workgroup barrier, LDS
//     instructions of one wavefront execute in order; the reads of row r+1 are issued between the row pass and the
//     column pass of row r, whose arithmetic covers their latency;
//   * the COLUMN filter is the running sum of dwt2_wave_kernels.hpp: a filtered row is added, with the matching tap,
//     to the hlen/2 output rows it contributes to (packed multiply-adds on interleaved (L,H) pairs, taps in SGPRs,
//     accumulators rotate through static register slots in a group of hlen rows); a finished output row leaves
//     through range-checked buffer stores (columns right of the image and rows below the segment are dropped by the
//     descriptor, so no store sits under a branch and the s_waitcnt counts of the load ring stay exact).
// Every staged row is filtered along x exactly once per segment; the only recomputation is the hlen-2 rows two
// vertically adjacent segments share.  Segments ALTERNATE their direction (even ones walk down, odd ones up, with the
// column taps taken in reverse order): two neighbours then meet at their common border -- they read the rows they share
// at the same time, and the second reader finds them in the XCD's L2 (rocprofv3, 16 taps, 4096^2, all walking down:
// FETCH_SIZE 92 MB for a 64 MB image with 16-row segments -- the shared rows were read 20 us apart).  Vector instructions per image row and wavefront: 2 hlen OPL packed multiply-adds
// (row + column pass) + the loads / LDS traffic above; no per-element index arithmetic.
//
// CPU emulation (tests/cpu_emu): as dwt2_wave_kernels.hpp, a phase ends where lanes read what other lanes wrote.
#pragma once

#include "dwt2_wave_kernels.hpp"

namespace pdwt {

// Diagnostic builds only (tools/ringbench.hip): PDWT_RING_DIAG is a mask of parts of the FORWARD kernel to leave out --
// 1: the multiply-adds, 2: the stores, 4: the LDS round trip, 8: the global loads after the prologue.  Results are wrong.
#ifndef PDWT_RING_DIAG
#define PDWT_RING_DIAG 0
#endif

constexpr int kRingMinHlen = 10;  // shorter filters: dwt2_wave_kernels.hpp (DPP, no LDS) and the 64 x 8 LDS tiles
constexpr int kRingMaxHlen = 20;  // the unrolled group is hlen rows long: longer filters stay on the LDS tiles

// Orders the LDS accesses of ONE wavefront for the compiler (lanes read what other lanes wrote).  The hardware
// executes a wavefront's DS instructions in order, so no wait is needed: a wavefront-scope fence emits nothing.
#ifdef PDWT_CPU_EMU
#define PDWT_WAVE_SYNC() ((void)0)
#else
#define PDWT_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#endif

// N reals moved by one memory instruction
#ifdef PDWT_CPU_EMU
template <int N>
struct rvec {
    real_t e[N];
    real_t& operator[](int i) { return e[i]; }
    real_t operator[](int i) const { return e[i]; }
};
#else
template <int N>
using rvec = real_t __attribute__((ext_vector_type(N)));
#endif
template <int N>
PDWT_DEVICE rvec<N> vec_ld(const void* p) { return *reinterpret_cast<const rvec<N>*>(p); }
template <int N>
PDWT_DEVICE void vec_st(void* p, const rvec<N>& v) { *reinterpret_cast<rvec<N>*>(p) = v; }
// (uniform base) + (per-lane unsigned byte offset): global_load ... v_offset, s[base:base+1]
template <int N>
PDWT_DEVICE rvec<N> wave_ldn(const real_t* base, unsigned byte_off) {
    return vec_ld<N>(reinterpret_cast<const char*>(base) + byte_off);
}
PDWT_DEVICE real_t wave_ld1(const real_t* base, unsigned byte_off) {
    return *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Packed multiply-adds with one HALF of a register pair broadcast, written as plain vector expressions: hipcc (ROCm 7.2)
// selects the half through op_sel / op_sel_hi, and the (uniform) tap pair is used element-wise straight from its SGPR pair.
// The ring kernels pair their data so that ONLY these forms occur: broadcasting a half of an SGPR pair instead makes hipcc
// build a second pair per tap (twice the scalar registers: they spill into VGPR lanes), and the inline-asm forms of
// packed_math.hpp count as zero wait states for hipcc's hazard recogniser, which then pads every dependent pair with an s_nop.
//   bfma_x / bfma_y (p, t, acc):  acc + (p.x, p.x) * t   /   acc + (p.y, p.y) * t
static PDWT_DEVICE v2f bfma_x(v2f p, v2f t, v2f acc) { return fma2(mk2(p.x, p.x), t, acc); }
static PDWT_DEVICE v2f bfma_y(v2f p, v2f t, v2f acc) { return fma2(mk2(p.y, p.y), t, acc); }
static PDWT_DEVICE v2f bmul_x(v2f p, v2f t) { return mk2(p.x * t.x, p.x * t.y); }
static PDWT_DEVICE v2f bmul_y(v2f p, v2f t) { return mk2(p.y * t.x, p.y * t.y); }
static PDWT_DEVICE v2f mul2(v2f p, v2f t) { return mk2(p.x * t.x, p.y * t.y); }

// Walks over the rows of a plane in either direction.  analysis = true: the plane is extended to an even number of rows by
// repeating the last one, and that is periodized (pdwt/src/separable.cu:114-121); false: plain periodic (coefficient rows of
// the inverse).  The walk never ends: rows requested beyond a segment are valid rows nobody uses.
struct RingWalk {
    int yp, np, last, dir;
    PDWT_DEVICE void start(int y, int n, bool analysis, bool up) {
        np = analysis ? n + (n & 1) : n;
        yp = true_mod(y, np);
        last = n - 1;
        dir = up ? -1 : 1;
    }
    PDWT_DEVICE int next() {
        const int sy = yp < last ? yp : last;
        yp += dir;
        if (yp == np) yp = 0;
        if (yp < 0) yp = np - 1;
        return sy;
    }
};

// ------------------------------------------------------------------------------------------------
// forward level: in (Nr, Nc) -> A, H, V, D (Nr2, Nc2);  Nc % 4 == 0, 16-B aligned rows.  Arguments: FwdWaveArgs with
// strips = ceil(Nc / (64 CPL)).
// ------------------------------------------------------------------------------------------------
#ifndef PDWT_RING_NR16
#define PDWT_RING_NR16 4  // load-ring slots of the 16-tap kernels (4 or 8; A/B builds)
#endif
constexpr int ring_slots(int hlen) {  // a divisor of the group length (hlen rows)
    return hlen == 10 ? 5 : hlen == 12 ? 4 : hlen == 14 ? 7 : hlen == 16 ? PDWT_RING_NR16 : hlen == 18 ? 6 : hlen == 20 ? 5 : 4;
}

template <int HLEN, int CPL>
struct FwdRingGeom {
    static_assert(CPL == 2 || CPL == 4, "image columns per lane");
    static constexpr int C = HLEN / 2 - 1;  // analysis centre: output k reads x[2k - C .. 2k - C + hlen - 1]
    static constexpr int NS = HLEN / 2;     // output rows a filtered row contributes to
    static constexpr int OPL = CPL / 2;     // output columns per lane
    static constexpr int W = 64 * CPL;      // image columns per strip
    static constexpr int HL = (C + CPL - 1) / CPL * CPL;  // halo samples staged on each side (lane windows stay aligned)
    static constexpr int PADL = HL - C;                   // samples between a lane's aligned window start and its first tap
    static constexpr int NV = (PADL + HLEN + 2 * (OPL - 1) + CPL - 1) / CPL * CPL;  // window of a lane, in samples
    static constexpr int ROWF = W + 2 * HL;  // samples of an LDS row: [left halo | strip | right halo]
    static constexpr int LDS_REALS = ROWF;   // per wavefront: one row
    static constexpr int GR = HLEN;  // rows per unrolled group: the accumulator slots' period
    static constexpr int NR = ring_slots(HLEN);
    static_assert(HLEN >= kRingMinHlen && HLEN <= kRingMaxHlen && (HLEN & 1) == 0, "even filters of 10-20 taps");
    static_assert(2 * HL <= 64 && 63 * CPL + NV <= ROWF, "halo lanes / window inside the staged row");
    static_assert(GR % NR == 0 && (GR / 2) % NS == 0, "slot indices are static inside a group");
};

template <int HLEN, int CPL>
struct FwdRingState {
    using G = FwdRingGeom<HLEN, CPL>;
    WaveReg<real_t, CPL * G::NR> ring;       // [slot][CPL]: own columns of the rows in flight
    WaveReg<real_t, G::NR> hring;            // [slot]: this lane's halo sample
    WaveReg<v2f, G::NV / 2> vbuf;            // the lane's window of the row that went through LDS last, as pairs
    WaveReg<v2f, G::OPL> lh;                 // the row-filtered (L, H) of the lane's output columns
    WaveReg<v2f, 2 * G::OPL * G::NS> acc;    // per slot: (A,H) of the OPL output columns, then (V,D)
    WaveReg<unsigned, 3> off;                // byte offsets: own columns, halo sample, own outputs
    WaveReg<int, 1> lidx;                    // LDS sample index of the halo sample (-1: none)
};

template <int HLEN, int CPL, bool PROLOGUE = false>
PDWT_DEVICE void fwd_ring_load(FwdRingState<HLEN, CPL>& st, int slot, const real_t* in, int Nc, RingWalk& walk) {
    if ((PDWT_RING_DIAG & 8) && !PROLOGUE) return;
    const real_t* row = in + (long long)walk.next() * Nc;
    PDWT_WAVE_LANES(lane) {
        const rvec<CPL> x = wave_ldn<CPL>(row, st.off.mine(lane)[0]);
        real_t* r = st.ring.mine(lane) + CPL * slot;
#pragma unroll
        for (int q = 0; q < CPL; ++q) r[q] = x[q];
        st.hring.mine(lane)[slot] = wave_ld1(row, st.off.mine(lane)[1]);
    }
}

// the row in ring slot SLOT goes through the wavefront's LDS row and comes back as every lane's window
template <int HLEN, int CPL, int SLOT>
PDWT_DEVICE void fwd_ring_stage(FwdRingState<HLEN, CPL>& st, real_t* row) {
    using G = FwdRingGeom<HLEN, CPL>;
    if (PDWT_RING_DIAG & 4) {
        PDWT_WAVE_LANES(lane) {
            v2f* v = st.vbuf.mine(lane);
            const real_t* r = st.ring.mine(lane) + CPL * SLOT;
            for (int q = 0; q < G::NV / 2; ++q) v[q] = mk2(r[q % CPL] + (real_t)q, st.hring.mine(lane)[SLOT]);
        }
        return;
    }
    PDWT_WAVE_SYNC();
    PDWT_WAVE_LANES(lane) {
        const real_t* r = st.ring.mine(lane) + CPL * SLOT;
        rvec<CPL> x;
#pragma unroll
        for (int q = 0; q < CPL; ++q) x[q] = r[q];
        vec_st<CPL>(row + G::HL + CPL * lane, x);
        const int hi = st.lidx.mine(lane)[0];
        if (hi >= 0) row[hi] = st.hring.mine(lane)[SLOT];
    }
    PDWT_WAVE_SYNC();
    PDWT_WAVE_LANES(lane) {
        v2f* v = st.vbuf.mine(lane);
#pragma unroll
        for (int q = 0; q < G::NV / CPL; ++q) {
            const rvec<CPL> x = vec_ld<CPL>(row + CPL * lane + CPL * q);
#pragma unroll
            for (int e = 0; e < CPL / 2; ++e) v[q * (CPL / 2) + e] = mk2(x[2 * e], x[2 * e + 1]);
        }
    }
}

// row filter of the window in vbuf -> lh
template <int HLEN, int CPL>
PDWT_DEVICE void fwd_ring_rowpass(FwdRingState<HLEN, CPL>& st, const FilterBankI& fb) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int OPL = G::OPL, PADL = G::PADL;
    if (PDWT_RING_DIAG & 1) {
        PDWT_WAVE_LANES(lane) {
            const v2f* v = st.vbuf.mine(lane);
            v2f s = v[0];
            for (int q = 1; q < G::NV / 2; ++q) s = mk2(s.x + v[q].x, s.y + v[q].y);  // keeps the window alive
            for (int o = 0; o < OPL; ++o) st.lh.mine(lane)[o] = s;
        }
        return;
    }
    PDWT_WAVE_LANES(lane) {
        const v2f* v = st.vbuf.mine(lane);
        // consecutive packed multiply-adds are kept INDEPENDENT (hipcc pads a dependent back-to-back pair with an s_nop):
        // two output columns alternate; a single output column is summed as even taps + odd taps
        constexpr int NP = OPL == 1 ? 2 : 1;
        constexpr int kPadl = PADL;
        v2f part[OPL][NP] = {};
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const v2f tap = fb.t[HLEN - 1 - j];
#pragma unroll
            for (int o = 0; o < OPL; ++o) {
                const int e = kPadl + 2 * o + j;
                const v2f p = v[e >> 1];
                v2f& s = part[o][j % NP];
                if (j < NP) s = (e & 1) ? bmul_y(p, tap) : bmul_x(p, tap);
                else s = (e & 1) ? bfma_y(p, tap, s) : bfma_x(p, tap, s);
            }
        }
        v2f* lh = st.lh.mine(lane);
#pragma unroll
        for (int o = 0; o < OPL; ++o) {
            lh[o] = part[o][0];
            if (NP == 2) lh[o] = mk2(part[o][0].x + part[o][NP - 1].x, part[o][0].y + part[o][NP - 1].y);
        }
    }
}

// The filtered row w of the wavefront's walk (w = 0 is the first row it needs) is added to the running column sums:
// it is tap j = (w & 1) + 2 d -- of the filter taken in the walk's direction -- of the output rows i = (w >> 1) - d,
// d = 0 .. NS-1, counted in the walk's direction too.  R1 = w & 1, IH = (w >> 1) mod NS (static).  WARM: one of the first
// HLEN - 2 rows (contributes only to the segment's own output rows, i >= 0: DMAX = w >> 1).
template <int HLEN, int CPL, bool UP, int R1, int IH, bool WARM, int DMAX>
PDWT_DEVICE void fwd_ring_colpass(FwdRingState<HLEN, CPL>& st, const FilterBankI& fb) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int NS = G::NS, OPL = G::OPL;
    if (PDWT_RING_DIAG & 1) {
        PDWT_WAVE_LANES(lane) {
            constexpr int kNS = NS;
            v2f* s = st.acc.mine(lane) + 2 * OPL * (((IH - (R1 ? kNS - 1 : 0)) % kNS + kNS) % kNS);
            for (int o = 0; o < 2 * OPL; ++o) s[o] = st.lh.mine(lane)[o % OPL];
        }
        return;
    }
    PDWT_WAVE_LANES(lane) {
        const v2f* lh = st.lh.mine(lane);
        v2f* a = st.acc.mine(lane);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            if (WARM && d > DMAX) continue;  // output rows outside the segment: another wavefront's
            constexpr int kNS = NS;
            const int slot = ((IH - d) % kNS + kNS) % kNS;
            const int j = R1 + 2 * d;
            const v2f tap = fb.t[UP ? j : HLEN - 1 - j];
            v2f* s = a + 2 * OPL * slot;
#pragma unroll
            for (int o = 0; o < OPL; ++o) {
                s[o] = j == 0 ? bmul_x(lh[o], tap) : bfma_x(lh[o], tap, s[o]);                    // (A,H) += L (lo, hi)
                s[OPL + o] = j == 0 ? bmul_y(lh[o], tap) : bfma_y(lh[o], tap, s[OPL + o]);  // (V,D) += H (lo, hi)
            }
        }
    }
}

// store the finished output row held in accumulator slot SLOT_OUT; row_ok = the row belongs to this segment
template <int HLEN, int CPL, int SLOT_OUT>
PDWT_DEVICE void fwd_ring_store(FwdRingState<HLEN, CPL>& st, const FwdWaveArgs& a, long long rowoff, bool row_ok) {
    constexpr int OPL = CPL / 2;
    const unsigned bytes = row_ok && !(PDWT_RING_DIAG & 2) ? (unsigned)a.Nc2 * kRealBytes : ((PDWT_RING_DIAG & 2) && a.Nc2 == -12345 ? 4u : 0u);
    const RowBuf bA = row_buf(a.A + rowoff, bytes), bV = row_buf(a.V + rowoff, bytes);
    const RowBuf bH = row_buf(a.H + rowoff, bytes), bD = row_buf(a.D + rowoff, bytes);
    PDWT_WAVE_LANES(lane) {
        const v2f* s = st.acc.mine(lane) + 2 * OPL * SLOT_OUT;
        const unsigned o = st.off.mine(lane)[2];
        if constexpr (OPL == 2) {
            row_st8(bA, o, s[0].x, s[1].x);
            row_st8(bH, o, s[0].y, s[1].y);
            row_st8(bV, o, s[2].x, s[3].x);
            row_st8(bD, o, s[2].y, s[3].y);
        } else {
            row_st4(bA, o, s[0].x);
            row_st4(bH, o, s[0].y);
            row_st4(bV, o, s[1].x);
            row_st4(bD, o, s[1].y);
        }
    }
}

// Row w of the walk: its window is in vbuf.  Filter it along x; send row w + 1 through LDS (its ring slot then takes row
// w + 1 + NR); add the filtered row to the column sums while those LDS reads are in flight.
template <int HLEN, int CPL, bool UP, int W_, bool WARM>
PDWT_DEVICE void fwd_ring_row(FwdRingState<HLEN, CPL>& st, const FwdWaveArgs& a, const real_t* in, real_t* lds, RingWalk& walk) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int NS = G::NS, NR = G::NR;
    // the phases keep their order (scheduling fences): the LDS reads issued by the stage are in flight during the column pass
    fwd_ring_rowpass<HLEN, CPL>(st, a.fb);
    PDWT_ROW_FENCE();
    fwd_ring_stage<HLEN, CPL, (W_ + 1) % NR>(st, lds);
    fwd_ring_load<HLEN, CPL>(st, (W_ + 1) % NR, in, a.Nc, walk);
    PDWT_ROW_FENCE();
    fwd_ring_colpass<HLEN, CPL, UP, (W_ & 1), ((W_ >> 1) % NS), WARM, (W_ >> 1)>(st, a.fb);
}

// One group of GR rows of the steady state (after the HLEN - 2 warm-up rows): rows w = HLEN - 2 + GR it + g.
template <int HLEN, int CPL, bool UP, int G0 = 0>
PDWT_DEVICE void fwd_ring_group(FwdRingState<HLEN, CPL>& st, const FwdWaveArgs& a, const real_t* in, real_t* lds,
                                RingWalk& walk, long long& rowoff, int& out_left) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int NS = G::NS, GR = G::GR;
    if constexpr (G0 < GR) {
        constexpr int w = HLEN - 2 + G0;  // + GR it: slot indices do not depend on it
        fwd_ring_row<HLEN, CPL, UP, w, false>(st, a, in, lds, walk);
        if constexpr (G0 & 1) {  // output row (w >> 1) - NS + 1 of the walk is complete
            constexpr int so = ((((w >> 1) - (NS - 1)) % NS) + NS) % NS;
            fwd_ring_store<HLEN, CPL, so>(st, a, rowoff, out_left > 0);
            rowoff += UP ? -(long long)a.Nc2 : (long long)a.Nc2;
            --out_left;
        }
        PDWT_ROW_FENCE();
        fwd_ring_group<HLEN, CPL, UP, G0 + 1>(st, a, in, lds, walk, rowoff, out_left);
    }
}

template <int HLEN, int CPL, bool UP, int R = 0>
PDWT_DEVICE void fwd_ring_warmup(FwdRingState<HLEN, CPL>& st, const FwdWaveArgs& a, const real_t* in, real_t* lds,
                                 RingWalk& walk) {
    if constexpr (R < HLEN - 2) {
        fwd_ring_row<HLEN, CPL, UP, R, true>(st, a, in, lds, walk);
        PDWT_ROW_FENCE();
        fwd_ring_warmup<HLEN, CPL, UP, R + 1>(st, a, in, lds, walk);
    }
}

template <int HLEN, int CPL, bool UP>
PDWT_DEVICE void fwd_ring_walk(FwdRingState<HLEN, CPL>& st, const FwdWaveArgs& a, const real_t* in, real_t* lds, int oy0, int oy_end,
                               long long plane_off) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int C = G::C, GR = G::GR, NR = G::NR;
    int out_left = oy_end - oy0;
    // down: image rows 2 oy0 - C, + 1, ...; outputs oy0, oy0 + 1, ...    up: image rows 2 (oy_end - 1) - C + HLEN - 1, - 1, ...;
    // outputs oy_end - 1, oy_end - 2, ...
    long long rowoff = plane_off + (long long)(UP ? oy_end - 1 : oy0) * a.Nc2;
    RingWalk walk;
    walk.start(UP ? 2 * (oy_end - 1) - C + HLEN - 1 : 2 * oy0 - C, a.Nr, true, UP);
    // rows 0 .. NR-1 in flight (row w lives in ring slot w % NR); row 0 through LDS before the first row is filtered
#pragma unroll
    for (int p = 0; p < NR; ++p) fwd_ring_load<HLEN, CPL, true>(st, p, in, a.Nc, walk);
    fwd_ring_stage<HLEN, CPL, 0>(st, lds);
    fwd_ring_load<HLEN, CPL>(st, 0, in, a.Nc, walk);
    fwd_ring_warmup<HLEN, CPL, UP>(st, a, in, lds, walk);
    const int ngroups = (out_left + GR / 2 - 1) / (GR / 2);
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) fwd_ring_group<HLEN, CPL, UP>(st, a, in, lds, walk, rowoff, out_left);
}

// One wavefront: strip `strip` (image columns [W strip, W strip + W)), output rows [seg * seg_out, (seg + 1) * seg_out)
// of image bz; lds = this wavefront's FwdRingGeom::LDS_REALS reals.  Odd segments walk up.
template <int HLEN, int CPL>
PDWT_DEVICE void dwt2_fwd_ring(const FwdWaveArgs& a, int strip, int seg, int bz, real_t* lds) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int W = G::W, HL = G::HL, OPL = G::OPL;

    const int oy0 = seg * a.seg_out;
    int oy_end = oy0 + a.seg_out;
    if (oy_end > a.Nr2) oy_end = a.Nr2;
    if (oy_end <= oy0) return;
    const int x0 = strip * W;
    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;

    FwdRingState<HLEN, CPL> st;
    PDWT_WAVE_LANES(lane) {
        // lanes past the right image edge load the periodic continuation: their neighbours need it
        st.off.mine(lane)[0] = kRealBytes * (unsigned)wrap_periodic(x0 + CPL * lane, a.Nc);
        const int h = lane < 2 * HL ? lane : 2 * HL - 1;  // the other lanes repeat the last halo lane's load (same cache line)
        st.off.mine(lane)[1] = kRealBytes * (unsigned)wrap_periodic(h < HL ? x0 - HL + h : x0 + W + (h - HL), a.Nc);
        st.off.mine(lane)[2] = kRealBytes * (unsigned)((x0 >> 1) + OPL * lane);  // past the row end: dropped by the range check
        st.lidx.mine(lane)[0] = lane < HL ? lane : (lane < 2 * HL ? W + lane : -1);
    }
    if (seg & 1) fwd_ring_walk<HLEN, CPL, true>(st, a, in, lds, oy0, oy_end, (long long)bz * a.out_bstride);
    else fwd_ring_walk<HLEN, CPL, false>(st, a, in, lds, oy0, oy_end, (long long)bz * a.out_bstride);
}

// ------------------------------------------------------------------------------------------------
// inverse level: A, H, V, D (Nrc, Ncc) -> out (Nr, Nc);  Nc == 2 Ncc, Ncc even, 16-B aligned rows.
//
// Synthesis along one axis as in dwt2_wave_kernels.hpp (pdwt/src/separable.cu:246-328 restated): with H2 = hlen/2,
// C = H2/2, S = 1 - (H2 & 1), coefficient "pair" K yields the two samples
//     out[2K + 2C - S]     = sum_j  lo[hlen-2-2j] a[K+j] + hi[hlen-2-2j] d[K+j]
//     out[2K + 2C - S + 1] = sum_j  lo[hlen-1-2j] a[K+j] + hi[hlen-1-2j] d[K+j]        j = 0 .. H2-1
// (indices periodic).  A wavefront walks a strip of 64 CCL coefficient columns (CCL = CPL / 2 per lane, CPL image
// columns below them).  Per coefficient row: the four band rows go through the wavefront's LDS row as interleaved (A,V) and
// (H,D) pairs -- the strip's halo columns of all four bands by ONE 4-B load per lane --; ROW synthesis on the lane's window:
// (A,V) x (lo, hi) tap pair element-wise, the two halves add up to the low-y sample, (H,D) likewise to the high-y sample,
// u = (low-y, high-y) for the lane's CPL image columns; the COLUMN synthesis is the running sum of dwt2_wave_kernels.hpp
// over the H2 pairs of image rows the row contributes to, accumulators paired as (even row, odd row).  A finished pair of
// image rows leaves as two range-checked buffer stores per lane.  Odd segments walk up (taps of the column synthesis in
// reverse order), as in the forward kernel.
// ------------------------------------------------------------------------------------------------
struct InvRingArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int strips;      // ceil(Ncc / (64 CCL))
    int segs;        // ceil(Nrc / seg_pairs)
    int seg_pairs;   // coefficient rows ("pairs" of image rows) per wavefront
    FilterBankI fb;  // (rec_lo[j], rec_hi[j])
    v2f pl[kRingMaxHlen / 2], ph[kRingMaxHlen / 2];  // (lo[h-2-2d], lo[h-1-2d]), (hi[h-2-2d], hi[h-1-2d])
};

#ifndef PDWT_RING_INV_NR16
#define PDWT_RING_INV_NR16 4  // load-ring slots of the 16-tap inverse (4 or 8; A/B builds)
#endif
constexpr int inv_ring_slots(int hlen) {  // a divisor of the group length
    return hlen == 10 ? 5 : hlen == 12 ? 3 : hlen == 14 ? 7 : hlen == 16 ? PDWT_RING_INV_NR16 : hlen == 18 ? 3 : hlen == 20 ? 5 : 4;
}

template <int HLEN, int CPL>
struct InvRingGeom {
    static_assert(CPL == 2 || CPL == 4, "image columns per lane");
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int S = (H2 & 1) ? 0 : 1;
    static constexpr int CCL = CPL / 2;        // coefficient columns per lane
    static constexpr int WC = 64 * CCL;        // coefficient columns per strip
    static constexpr int HLC = (C + CCL - 1) / CCL * CCL;  // halo columns staged left of the strip
    static constexpr int PADC = HLC - C;                   // columns between a lane's aligned window start and the first one it uses
    static constexpr int NCOLS = CCL + S + H2 - 1;         // coefficient columns a lane's CPL samples depend on
    static constexpr int NVC = (PADC + NCOLS + CCL - 1) / CCL * CCL;  // window of a lane, in columns
    static constexpr int HRC = NVC - CCL - HLC > 0 ? NVC - CCL - HLC : 0;  // halo columns right of the strip
    static constexpr int NH = HLC + HRC;       // halo columns per band
    static constexpr int ROWC = HLC + WC + HRC;  // columns of an LDS row
    static constexpr int PLANE = 2 * ROWC;       // reals of one pair plane ((A,V) or (H,D)) of a row
    static constexpr int LDS_REALS = 2 * PLANE;  // one row of two planes per wavefront
    static constexpr int GR = H2;                // rows per unrolled group
    static constexpr int NR = inv_ring_slots(HLEN);
    static_assert(HLEN >= kRingMinHlen && HLEN <= kRingMaxHlen && (HLEN & 1) == 0, "even filters of 10-20 taps");
    static_assert(4 * NH <= 64 && 63 * CCL + NVC <= ROWC, "halo lanes / window inside the staged row");
    static_assert(GR % NR == 0 && GR % H2 == 0, "slot indices are static inside a group");
};

// N reals to a range-checked row
template <int N>
PDWT_DEVICE void row_stn(const RowBuf& r, unsigned off, const real_t* v) {
#ifdef PDWT_CPU_EMU
    if (off < r.bytes) {
        real_t* p = reinterpret_cast<real_t*>(r.base + off);
        for (int i = 0; i < N; ++i) p[i] = v[i];
    }
#else
    if constexpr (N * sizeof(real_t) == 16) {
        rvec<N> x;
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = v[i];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pdwt_u4, x), r.rsrc, (int)off, 0, 0);
    } else if constexpr (N * sizeof(real_t) == 32) {
        row_stn<N / 2>(r, off, v);
        row_stn<N / 2>(r, off + 16u, v + N / 2);
    } else {
        static_assert(N == 2, "pairs");
        row_st8(r, off, v[0], v[1]);
    }
#endif
}

template <int HLEN, int CPL>
struct InvRingState {
    using G = InvRingGeom<HLEN, CPL>;
    WaveReg<real_t, 4 * G::CCL * G::NR> ring;  // [slot][band A,H,V,D][CCL]: the lane's coefficient columns
    WaveReg<real_t, G::NR> hring;              // [slot]: this lane's halo sample (one band, one column)
    WaveReg<v2f, 2 * G::NVC> vbuf;             // [plane][NVC]: the lane's window, (A,V) then (H,D) pairs
    WaveReg<v2f, CPL> u;                       // the row-synthesised (low-y, high-y) samples of the lane's image columns
    WaveReg<v2f, CPL * G::H2> acc;             // [slot][image column] = (even row, odd row)
    WaveReg<unsigned, 2> off;                  // byte offsets: own coefficient columns, own output samples
    WaveReg<const real_t*, 1> hptr;            // halo sample: its band's plane + its (wrapped) column
    WaveReg<int, 1> lidx;                      // halo sample: LDS real index inside a row (-1: none)
};

// ro = element offset of the coefficient row (image + row) inside every band
template <int HLEN, int CPL>
PDWT_DEVICE void inv_ring_load(InvRingState<HLEN, CPL>& st, int slot, const InvRingArgs& a, long long boff, RingWalk& walk) {
    constexpr int CCL = CPL / 2;
    const long long ro = boff + (long long)walk.next() * a.Ncc;
    PDWT_WAVE_LANES(lane) {
        const unsigned o = st.off.mine(lane)[0];
        real_t* r = st.ring.mine(lane) + 4 * CCL * slot;
        if constexpr (CCL == 2) {
            const rvec<2> xa = wave_ldn<2>(a.A + ro, o), xh = wave_ldn<2>(a.H + ro, o);
            const rvec<2> xv = wave_ldn<2>(a.V + ro, o), xd = wave_ldn<2>(a.D + ro, o);
            r[0] = xa[0]; r[1] = xa[1]; r[2] = xh[0]; r[3] = xh[1];
            r[4] = xv[0]; r[5] = xv[1]; r[6] = xd[0]; r[7] = xd[1];
        } else {
            r[0] = wave_ld1(a.A + ro, o); r[1] = wave_ld1(a.H + ro, o);
            r[2] = wave_ld1(a.V + ro, o); r[3] = wave_ld1(a.D + ro, o);
        }
        st.hring.mine(lane)[slot] = st.hptr.mine(lane)[0][ro];
    }
}

// the coefficient row in ring slot SLOT goes through the wavefront's LDS row and comes back as every lane's window
template <int HLEN, int CPL, int SLOT>
PDWT_DEVICE void inv_ring_stage(InvRingState<HLEN, CPL>& st, real_t* row) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int CCL = G::CCL;
    PDWT_WAVE_SYNC();
    PDWT_WAVE_LANES(lane) {
        const real_t* r = st.ring.mine(lane) + 4 * CCL * SLOT;
        rvec<2 * CCL> av, hd;  // (A,V) and (H,D) interleaved per column
#pragma unroll
        for (int c = 0; c < CCL; ++c) {
            av[2 * c] = r[c]; av[2 * c + 1] = r[2 * CCL + c];
            hd[2 * c] = r[CCL + c]; hd[2 * c + 1] = r[3 * CCL + c];
        }
        vec_st<2 * CCL>(row + 2 * (G::HLC + CCL * lane), av);
        vec_st<2 * CCL>(row + G::PLANE + 2 * (G::HLC + CCL * lane), hd);
        const int hi = st.lidx.mine(lane)[0];
        if (hi >= 0) row[hi] = st.hring.mine(lane)[SLOT];
    }
    PDWT_WAVE_SYNC();
    PDWT_WAVE_LANES(lane) {
        v2f* v = st.vbuf.mine(lane);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int q = 0; q < G::NVC / CCL; ++q) {
                const rvec<2 * CCL> x = vec_ld<2 * CCL>(row + pl * G::PLANE + 2 * CCL * lane + 2 * CCL * q);
#pragma unroll
                for (int c = 0; c < CCL; ++c) v[pl * G::NVC + q * CCL + c] = mk2(x[2 * c], x[2 * c + 1]);
            }
    }
}

// row synthesis of the window in vbuf -> u
template <int HLEN, int CPL>
PDWT_DEVICE void inv_ring_rowpass(InvRingState<HLEN, CPL>& st, const InvRingArgs& a) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int H2 = G::H2, S = G::S, PADC = G::PADC, NVC = G::NVC;
    PDWT_WAVE_LANES(lane) {
        const v2f* pAV = st.vbuf.mine(lane);  // window column m = coefficient column k0 - HLC + m
        const v2f* pHD = pAV + NVC;
        // image columns q = 0 .. CPL-1 of this lane.  (A,V) x (lo, hi) tap pair, element-wise: the two halves add up to the
        // low-y sample; (H,D) likewise to the high-y sample.  The 2 CPL chains alternate, so consecutive packed multiply-adds
        // are independent.
        v2f sl[CPL] = {}, sh[CPL] = {};
#pragma unroll
        for (int j = 0; j < H2; ++j) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const bool odd = q & 1;
                const bool use_te = S ? odd : !odd;  // taps hlen-2-2j ("te") or hlen-1-2j ("to")
                constexpr int kPadc = PADC;
                const int m0 = kPadc + (q >> 1) + ((S && odd) ? 1 : 0);  // first column: k0 + (q >> 1) - C (+ 1)
                const v2f tap = a.fb.t[use_te ? HLEN - 2 - 2 * j : HLEN - 1 - 2 * j];
                sl[q] = j == 0 ? mul2(pAV[m0 + j], tap) : fma2(pAV[m0 + j], tap, sl[q]);
                sh[q] = j == 0 ? mul2(pHD[m0 + j], tap) : fma2(pHD[m0 + j], tap, sh[q]);
            }
        }
        v2f* u = st.u.mine(lane);  // (low-y band, high-y band)
#pragma unroll
        for (int q = 0; q < CPL; ++q) u[q] = mk2(sl[q].x + sl[q].y, sh[q].x + sh[q].y);
    }
}

// Column synthesis: the synthesised row w of the walk is element d -- of the taps taken in the walk's direction -- of the
// pairs p = w - d (counted in the walk's direction); N2 = w mod H2 (static).  WARM: one of the first H2 - 1 rows (DMAX = w).
template <int HLEN, int CPL, bool UP, int N2, bool WARM, int DMAX>
PDWT_DEVICE void inv_ring_colpass(InvRingState<HLEN, CPL>& st, const InvRingArgs& a) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int H2 = G::H2;
    PDWT_WAVE_LANES(lane) {
        const v2f* u = st.u.mine(lane);
        v2f* acc = st.acc.mine(lane);
#pragma unroll
        for (int d = 0; d < H2; ++d) {
            if (WARM && d > DMAX) continue;  // pairs outside the segment: another wavefront's
            constexpr int kH2 = H2;
            const int slot = ((N2 - d) % kH2 + kH2) % kH2;
            const int t = UP ? kH2 - 1 - d : d;
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                acc[CPL * slot + q] = d == 0 ? bmul_x(u[q], a.pl[t]) : bfma_x(u[q], a.pl[t], acc[CPL * slot + q]);
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[CPL * slot + q] = bfma_y(u[q], a.ph[t], acc[CPL * slot + q]);
        }
    }
}

// store the finished pair of image rows held in accumulator slot SLOT_OUT
template <int HLEN, int CPL, int SLOT_OUT>
PDWT_DEVICE void inv_ring_store(InvRingState<HLEN, CPL>& st, const InvRingArgs& a, long long boff, int oy_e, int oy_o,
                                bool pair_ok) {
    const unsigned row_bytes = (unsigned)a.Nc * kRealBytes;
    const RowBuf be = row_buf(a.out + boff + (long long)oy_e * a.Nc, pair_ok && oy_e < a.Nr ? row_bytes : 0u);
    const RowBuf bo = row_buf(a.out + boff + (long long)oy_o * a.Nc, pair_ok && oy_o < a.Nr ? row_bytes : 0u);
    PDWT_WAVE_LANES(lane) {
        const v2f* s = st.acc.mine(lane) + CPL * SLOT_OUT;
        const unsigned o = st.off.mine(lane)[1];
        real_t e[CPL], d[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            e[q] = s[q].x;
            d[q] = s[q].y;
        }
        row_stn<CPL>(be, o, e);
        row_stn<CPL>(bo, o, d);
    }
}

// image rows of the next pair, in either direction: (oy_e, oy_e + 1) advance by two, periodic over 2 Nrc
struct RingPairRows {
    int oy_e, period, step2;
    PDWT_DEVICE void start(int K, int C, int S, int Nrc, bool up) {
        period = 2 * Nrc;
        oy_e = true_mod(2 * K + 2 * C - S, period);
        step2 = up ? -2 : 2;
    }
    PDWT_DEVICE int even() const { return oy_e; }
    PDWT_DEVICE int odd() const { return oy_e + 1 == period ? 0 : oy_e + 1; }
    PDWT_DEVICE void step() {
        oy_e += step2;
        if (oy_e >= period) oy_e -= period;
        if (oy_e < 0) oy_e += period;
    }
};

template <int HLEN, int CPL, bool UP, int W_, bool WARM>
PDWT_DEVICE void inv_ring_row(InvRingState<HLEN, CPL>& st, const InvRingArgs& a, real_t* lds, long long bin, RingWalk& walk) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int H2 = G::H2, NR = G::NR;
    inv_ring_rowpass<HLEN, CPL>(st, a);
    PDWT_ROW_FENCE();
    inv_ring_stage<HLEN, CPL, (W_ + 1) % NR>(st, lds);
    inv_ring_load<HLEN, CPL>(st, (W_ + 1) % NR, a, bin, walk);
    PDWT_ROW_FENCE();
    inv_ring_colpass<HLEN, CPL, UP, W_ % H2, WARM, W_>(st, a);
}

template <int HLEN, int CPL, bool UP, int G0 = 0>
PDWT_DEVICE void inv_ring_group(InvRingState<HLEN, CPL>& st, const InvRingArgs& a, real_t* lds, long long bin, long long bout,
                                RingWalk& walk, RingPairRows& rows, int& pairs_left) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int H2 = G::H2, GR = G::GR;
    if constexpr (G0 < GR) {
        constexpr int w = H2 - 1 + G0;  // + GR it
        inv_ring_row<HLEN, CPL, UP, w, false>(st, a, lds, bin, walk);
        constexpr int so = (((w - (H2 - 1)) % H2) + H2) % H2;  // pair w - H2 + 1 of the walk is complete
        inv_ring_store<HLEN, CPL, so>(st, a, bout, rows.even(), rows.odd(), pairs_left > 0);
        rows.step();
        --pairs_left;
        PDWT_ROW_FENCE();
        inv_ring_group<HLEN, CPL, UP, G0 + 1>(st, a, lds, bin, bout, walk, rows, pairs_left);
    }
}

template <int HLEN, int CPL, bool UP, int N = 0>
PDWT_DEVICE void inv_ring_warmup(InvRingState<HLEN, CPL>& st, const InvRingArgs& a, real_t* lds, long long bin, RingWalk& walk) {
    if constexpr (N < HLEN / 2 - 1) {
        inv_ring_row<HLEN, CPL, UP, N, true>(st, a, lds, bin, walk);
        PDWT_ROW_FENCE();
        inv_ring_warmup<HLEN, CPL, UP, N + 1>(st, a, lds, bin, walk);
    }
}

template <int HLEN, int CPL, bool UP>
PDWT_DEVICE void inv_ring_walk(InvRingState<HLEN, CPL>& st, const InvRingArgs& a, real_t* lds, int K0, int K_end, long long bin,
                               long long bout) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int H2 = G::H2, C = G::C, S = G::S, GR = G::GR, NR = G::NR;
    int pairs_left = K_end - K0;
    // down: coefficient rows K0, K0 + 1, ...; pairs K0, K0 + 1, ...    up: rows K_end - 1 + H2 - 1, - 1, ...; pairs K_end - 1, ...
    RingWalk walk;
    walk.start(UP ? K_end - 1 + H2 - 1 : K0, a.Nrc, false, UP);
    RingPairRows rows;
    rows.start(UP ? K_end - 1 : K0, C, S, a.Nrc, UP);
#pragma unroll
    for (int p = 0; p < NR; ++p) inv_ring_load<HLEN, CPL>(st, p, a, bin, walk);
    inv_ring_stage<HLEN, CPL, 0>(st, lds);
    inv_ring_load<HLEN, CPL>(st, 0, a, bin, walk);
    inv_ring_warmup<HLEN, CPL, UP>(st, a, lds, bin, walk);
    const int ngroups = (pairs_left + GR - 1) / GR;
#pragma unroll 1
    for (int it = 0; it < ngroups; ++it) inv_ring_group<HLEN, CPL, UP>(st, a, lds, bin, bout, walk, rows, pairs_left);
}

// One wavefront: strip `strip` (coefficient columns [WC strip, WC strip + WC)), pairs [seg * seg_pairs, (seg + 1) * seg_pairs)
// of image bz; lds = this wavefront's InvRingGeom::LDS_REALS reals.  Odd segments walk up.
template <int HLEN, int CPL>
PDWT_DEVICE void dwt2_inv_ring(const InvRingArgs& a, int strip, int seg, int bz, real_t* lds) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int CCL = G::CCL, WC = G::WC, HLC = G::HLC, NH = G::NH;

    const int K0 = seg * a.seg_pairs;
    int K_end = K0 + a.seg_pairs;
    if (K_end > a.Nrc) K_end = a.Nrc;
    if (K_end <= K0) return;
    const int kx0 = strip * WC;
    const long long bin = (long long)bz * a.in_bstride, bout = (long long)bz * a.out_bstride;

    InvRingState<HLEN, CPL> st;
    PDWT_WAVE_LANES(lane) {
        st.off.mine(lane)[0] = kRealBytes * (unsigned)wrap_periodic(kx0 + CCL * lane, a.Ncc);
        st.off.mine(lane)[1] = kRealBytes * (unsigned)(2 * kx0 + CPL * lane);  // past the row end: dropped by the range check
        // halo: lane h = band b, halo column c (left ones first); the other lanes repeat the last halo lane's load
        const int h = lane < 4 * NH ? lane : 4 * NH - 1;
        const int b = h / NH, c = h - b * NH;
        const int col = wrap_periodic(c < HLC ? kx0 - HLC + c : kx0 + WC + (c - HLC), a.Ncc);
        const real_t* plane = b == 0 ? a.A : (b == 1 ? a.H : (b == 2 ? a.V : a.D));
        st.hptr.mine(lane)[0] = plane + col;
        const int rowcol = c < HLC ? c : WC + c;
        st.lidx.mine(lane)[0] = lane < 4 * NH ? (b & 1) * G::PLANE + 2 * rowcol + (b >> 1) : -1;  // planes (A,V), (H,D)
    }
    if (seg & 1) inv_ring_walk<HLEN, CPL, true>(st, a, lds, K0, K_end, bin, bout);
    else inv_ring_walk<HLEN, CPL, false>(st, a, lds, K0, K_end, bin, bout);
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int CPL, int NT, int MINB>
__global__ void __launch_bounds__(NT, MINB) dwt2_fwd_ring_kernel(const FwdWaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, seg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!wave_tile(blockIdx.x, wave, NT / 64, a.strips, a.segs, strip, seg)) return;
    dwt2_fwd_ring<HLEN, CPL>(a, strip, seg, blockIdx.y, pdwt_smem + wave * FwdRingGeom<HLEN, CPL>::LDS_REALS);
}

template <int HLEN, int CPL, int NT, int MINB>
__global__ void __launch_bounds__(NT, MINB) dwt2_inv_ring_kernel(const InvRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, seg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!wave_tile(blockIdx.x, wave, NT / 64, a.strips, a.segs, strip, seg)) return;
    dwt2_inv_ring<HLEN, CPL>(a, strip, seg, blockIdx.y, pdwt_smem + wave * InvRingGeom<HLEN, CPL>::LDS_REALS);
}
#endif

}  // namespace pdwt
