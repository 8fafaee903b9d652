// dwt2_long_kernels.hpp -- 2D DWT levels of LONG filters (22-40 taps; built for every even length from 10 on), gfx950.
//
// Why another kernel family.  The LDS tiles of dwt2_fast_kernels.hpp were sized for 2-16 taps.  At 32-40 taps a 32 x 8
// inverse tile stages (8 + 21) x (32 + 24) coefficient quadruples for 8 x 32 of them (6.3x), runs its column synthesis over
// the whole staged width (1.75x the arithmetic), and reads LDS once per tap and output (16 B per 4 packed FMAs): the level-1
// inverse of db20 at 4096^2 took ~88 us where the packed-FMA floor is ~20 us and the HBM floor 17 us.  A long filter is
// ARITHMETIC-bound (40 taps: 40 packed FMAs per sample against 8 B), so this family removes the recomputation, not bytes:
//
//   * a workgroup owns a STRIP of TXC coefficient columns and walks DOWN it in steps of TY coefficient rows; what the second
//     pass needs from earlier steps stays in an LDS RING of row-filtered data, so nothing is filtered twice along y (the tiles
//     re-filter hlen - 2 halo rows per tile) and the x halo is only ever LOADED, never filtered twice;
//   * both passes are register-blocked: a thread of the second pass owns M consecutive outputs of one column and slides down
//     the ring -- every ring element is read once per M outputs instead of once per tap;
//   * every multiply-add is a packed v_pk_fma_f32 on a data pair and a (lo, hi) tap pair from the kernel-argument segment
//     (SGPRs); the inverse adds the two halves once per output;
//   * the next step's global loads are in flight while the current step is computed (registers, then one LDS write).
//
// Order of the inverse: ROW synthesis first (on the TY new coefficient rows only), column synthesis out of the ring.  The
// reference runs columns first (pdwt/src/separable.cu:332-395); the two orders are the same sums in a different association,
// i.e. they differ by fp32 rounding only, like every other kernel of this library against the oracle.
//
// Preconditions (checked by the host, launch_dwt2_long.hip): even filter length, even image sides, rows of whole 16-B
// groups (forward: Nc % 4 == 0; inverse: Ncc % 4 == 0, Nc == 2 Ncc), 16-B aligned planes.  Anything else stays on the tiles.
//
// Written as workgroup functions with barrier-separated phases (kernels_common.hpp) so that tests/cpu_emu runs the same
// code on the CPU.
#pragma once

#include "dwt2_fast_kernels.hpp"

namespace pdwt {

constexpr int kLongMinHlen = 10;  // shortest filter the family is instantiated for

// timing experiments (tools/longbench.hip -DPDWT_LONG_DIAG=mask, wrong results): 1 no row pass, 2 no column pass, 4 no global
// loads / staging, 8 no carry, 16 no barriers
#ifndef PDWT_LONG_DIAG
#define PDWT_LONG_DIAG 0
#endif
#if PDWT_LONG_DIAG & 32  // per-wavefront phase clocks (s_memtime) into pdwt_long_prof[8 * wavefront + phase]
extern __device__ unsigned long long pdwt_long_prof[];
#define PDWT_LONG_CLK(k) do { const unsigned long long now_ = clock64(); prof_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define PDWT_LONG_CLK(k) ((void)0)
#endif
// the items of a pass, tid + k NT: a fixed number of rounds, unrolled in the fp32 library (the scheduler overlaps the tail of one
// item -- stores, LDS writes -- with the first reads of the next); the fp64 library keeps the loop (twice the registers per value:
// two items in flight spill)
#ifdef PDWT_DOUBLE
#define PDWT_LONG_ITEMS(it, tid, ITEMS, NT) \
    _Pragma("unroll 1") for (int it = (tid), pdwt_k_ = 0; pdwt_k_ < ((ITEMS) + (NT) - 1) / (NT); ++pdwt_k_, it += (NT))
#else
#define PDWT_LONG_ITEMS(it, tid, ITEMS, NT) \
    _Pragma("unroll") for (int it = (tid), pdwt_k_ = 0; pdwt_k_ < ((ITEMS) + (NT) - 1) / (NT); ++pdwt_k_, it += (NT))
#endif
// The barriers of these kernels order LDS accesses only (nothing a workgroup's threads exchange goes through global memory):
// a release / acquire pair on the LOCAL address space around s_barrier.  __syncthreads() also waits for every outstanding global
// access of the wavefront (s_waitcnt vmcnt(0)): the step's stores and the next step's prefetched rows.
#if PDWT_LONG_DIAG & 16
#define PDWT_LONG_SYNC() ((void)0)
#elif defined(PDWT_CPU_EMU)
#define PDWT_LONG_SYNC() PDWT_SYNC()
#else
#define PDWT_LONG_SYNC()                                                    \
    do {                                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     \
        __builtin_amdgcn_s_barrier();                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");     \
    } while (0)
#endif

struct FwdLongArgs {
    const real_t* in;
    real_t *A, *H, *V, *D;
    int Nr, Nc, Nr2, Nc2;
    long long in_bstride, out_bstride;
    int strips;  // column strips of TXC output columns
    int segs;    // row segments per strip
    int seg;     // output rows per segment (a multiple of TY)
    FilterBankI fb;  // (dec_lo, dec_hi)
};

// Synthesis taps in "pair-out" order (see SynTaps): pl[i] = (rec_lo[te(i)], rec_lo[to(i)]), ph[i] the same of rec_hi; a
// half without a tap is 0 and never multiplied (the kernels use a one-sided FMA there, so that a non-finite coefficient
// outside a sample's support cannot reach it).
struct InvLongArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int strips;  // column strips of TXC coefficient columns
    int segs;
    int seg;     // coefficient rows per segment (a multiple of TY)
    v2f pl[kMaxTaps / 2 + 1], ph[kMaxTaps / 2 + 1];
};

constexpr int long_cdiv(int a, int b) { return (a + b - 1) / b; }

// a value every lane of the wavefront holds: keep it in a scalar register (addresses and guards built from it cost no
// vector instructions).  The caller guarantees the uniformity.
#ifdef PDWT_CPU_EMU
static inline int wave_uniform(int v) { return v; }
#else
static __device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// plane[row_off + col] = v with `row_off` SCALAR (the same in every lane) and `col` the lane's column: a buffer store takes
// the scalar part as its soffset operand and the lane's part as a 32-bit offset -- no vector instruction per store (written
// as a pointer expression, hipcc keeps a 64-bit per-lane address and adds the row to it: one v_lshl_add_u64 per store).
// Offsets are 32-bit BYTE offsets: the host sends planes of 4 GiB and more to the tiles.
#if !defined(PDWT_CPU_EMU) && !defined(PDWT_DOUBLE)
struct LanePlane {
    __amdgpu_buffer_rsrc_t rsrc;
};
PDWT_DEVICE LanePlane lane_plane(real_t* base) {
    LanePlane p;
    p.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)0xffffffffu, 0x00020000);
    return p;
}
PDWT_DEVICE void st_lane(const LanePlane& p, unsigned row_off, int col, real_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), p.rsrc, col * (int)sizeof(real_t),
                                          (int)(row_off * (unsigned)sizeof(real_t)), 0);
}
#else
struct LanePlane {
    real_t* base;
};
PDWT_DEVICE LanePlane lane_plane(real_t* base) { return LanePlane{base}; }
PDWT_DEVICE void st_lane(const LanePlane& p, unsigned row_off, int col, real_t v) { p.base[(size_t)row_off + col] = v; }
#endif

// ---------------------------------------------------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------------------------------------------------
// Synthesis in "pair" form (oracle/pdwt_oracle.c: syn_params; pdwt/src/separable.cu:250-287): the outputs (2m, 2m + 1) are
// fed by the WN = H2 + S coefficients m - C + i, i = 0 .. WN - 1 (H2 = hlen / 2, C = H2 / 2, S = 1 for even H2), output 2m
// through tap te(i), output 2m + 1 through tap to(i):
//     S = 0:  te(i) = hlen - 2 - 2i,               to(i) = hlen - 1 - 2i
//     S = 1:  te(i) = hlen - 1 - 2i  (i < H2),     to(i) = hlen - 2i  (i >= 1)
// so an accumulator PAIR (out[2m], out[2m + 1]) takes one packed FMA per coefficient and filter: the coefficient broadcast
// against (f[te(i)], f[to(i)]).  Nothing is interleaved, nothing is added across halves.
template <int HLEN>
struct SynTaps {
    static constexpr int H2 = HLEN / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1, WN = H2 + S;
    static constexpr int te(int i) { return S ? (i < H2 ? HLEN - 1 - 2 * i : -1) : HLEN - 2 - 2 * i; }
    static constexpr int to(int i) { return S ? (i >= 1 ? HLEN - 2 * i : -1) : HLEN - 1 - 2 * i; }
};

// host side: the pair-out tap tables of InvLongArgs from the reconstruction filters
template <int HLEN>
static inline void long_syn_tables(InvLongArgs& a, const real_t* rec_lo, const real_t* rec_hi) {
    using ST = SynTaps<HLEN>;
    for (int i = 0; i < kMaxTaps / 2 + 1; i++) {
        const int e = i < ST::WN ? ST::te(i) : -1, o = i < ST::WN ? ST::to(i) : -1;
        a.pl[i].x = e >= 0 ? rec_lo[e] : real_t(0); a.pl[i].y = o >= 0 ? rec_lo[o] : real_t(0);
        a.ph[i].x = e >= 0 ? rec_hi[e] : real_t(0); a.ph[i].y = o >= 0 ? rec_hi[o] : real_t(0);
    }
}

// acc += (d, d) * t for window position i (a constant after unrolling), d = the low / high half of `pr`; a half of
// position i that has no tap is left alone
template <int HLEN>
PDWT_DEVICE v2f syn_fma(int i, bool hi_half, v2f pr, v2f t, v2f acc) {
    using ST = SynTaps<HLEN>;
    const bool e = ST::te(i) >= 0, o = ST::to(i) >= 0;
    if (e && o) return hi_half ? fma2_by_v(pr, t, acc) : fma2_bx_v(pr, t, acc);  // t: in vector registers, see the kernel
    const real_t d = hi_half ? pr.y : pr.x;
    if (e) acc.x = pdwt_fma(d, t.x, acc.x);
    else if (o) acc.y = pdwt_fma(d, t.y, acc.y);
    return acc;
}

// The history buffer is LINEAR, not circular: rows [0, D) are the last D rows of the previous step ("carry"), rows
// [D, D + TY) the step's new rows, so that the window of an output is a compile-time row offset from a per-thread base (a
// circular buffer costs one vector add per row read -- the wrap point depends on the thread's rows --, 15 % more vector
// instructions in the second pass).  The carry rows move through registers between two barriers: source and destination
// overlap when D > TY.
template <int HLEN, int TXC, int TY>
struct InvLongGeom {
    using ST = SynTaps<HLEN>;
    static constexpr int H2 = ST::H2, C = ST::C, S = ST::S, WN = ST::WN;
    static constexpr int D = WN - 1;                     // rows (columns) of history an output pair needs
    static constexpr int W = long_cdiv(D, TY);           // warm-up steps: row synthesis only
    static constexpr int SKIP = W * TY - D;              // rows of step 0 that no output of the segment needs
    static constexpr int BR = D + TY;                    // buffer rows
    static constexpr int PADL = (4 - (C & 3)) & 3;       // columns between the aligned load origin and column k0 - C
    static constexpr int CXA = (PADL + TXC + D + 3) & ~3;  // staged coefficient columns per row
    static constexpr int OW = 2 * TXC;                   // output columns of the strip
    static constexpr int LDS_REALS = 4 * TY * CXA + 2 * BR * OW;
};

// One workgroup: strip `strip`, segment `seg`, image `bz`.
//   KB  coefficient columns per row-synthesis item (a multiple of 4: 2 KB outputs of both intermediate planes)
//   M   output row PAIRS per column-synthesis item, XB output columns per item (1: 8-B buffer reads, 2: 16-B)
template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB>
PDWT_DEVICE void dwt2_inv_long_wg(const InvLongArgs& a, int strip, int seg, int bz, real_t* smem) {
    using G = InvLongGeom<HLEN, TXC, TY>;
    constexpr int C = G::C, WN = G::WN, D = G::D, W = G::W, PADL = G::PADL, CXA = G::CXA, OW = G::OW;
    static_assert(TXC % KB == 0 && !(KB & 3), "row-synthesis items tile the strip in whole 16-B groups");
    static_assert(TY % M == 0 && OW % XB == 0 && (XB == 1 || XB == 2), "column-synthesis items tile the step");
    constexpr int V4 = CXA / 4, TOTAL = TY * V4, TRIPS = (TOTAL + NT - 1) / NT;
    constexpr int CARRY = D * OW / 2, CTRIPS = (CARRY + NT - 1) / NT;  // 16-B groups of the carry rows

    real_t* sA = smem;            // TY x CXA coefficients of the step's rows, one plane per band
    real_t* sV = sA + TY * CXA;
    real_t* sH = sV + TY * CXA;
    real_t* sD = sH + TY * CXA;
    v2f* buf = reinterpret_cast<v2f*>(sD + TY * CXA);  // BR x OW (u1,u2): row-synthesised (A,V) / (H,D), full output width

    const int k0 = strip * TXC, m0 = seg * a.seg;
    const int nm = a.Nrc - m0 < a.seg ? a.Nrc - m0 : a.seg;  // coefficient rows (= output row pairs) of this segment
    if (nm <= 0) return;
    const int T = W + (nm + TY - 1) / TY;
    const int rbase = m0 - C + D - W * TY;  // first coefficient row of step 0
    const int cxa = k0 - C - PADL;          // multiple of 4
    const long long boff = (long long)bz * a.in_bstride;
    const LanePlane oplane = lane_plane(a.out + (long long)bz * a.out_bstride);
    // The 2 WN tap pairs live in VECTOR registers (the same value in every lane): beside the kernel's pointers and sizes
    // 84 tap SGPRs spill (hipcc parks them in VGPR lanes: 216 v_readlane in a build with scalar taps), and a workgroup
    // that LDS holds to two per CU may use 256 VGPRs per lane anyway.
    v2f tl[WN], th[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        tl[i] = in_vgprs(a.pl[i]);
        th[i] = in_vgprs(a.ph[i]);
    }

    // Staging plan of a thread, fixed for the whole walk: LDS offset, source column and -- advanced by TY per step, one
    // conditional subtraction (Nrc >= TY, checked by the host) -- source row of each of its TRIPS coefficient groups.
    PDWT_PER_THREAD(int, plan, 3 * TRIPS, NT);
    PDWT_PER_THREAD(v4f, pre, 4 * TRIPS, NT);  // the next step's coefficient quadruples, in flight across the step
    PDWT_PER_THREAD(v4f, car, CTRIPS > 0 ? CTRIPS : 1, NT);  // the carry rows on their way to the top of the buffer
    auto make_plan = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;  // constant trip count: the clamped threads re-load (and re-write) the last group
            const int r = idx / V4;
            const int g = idx - r * V4;
            pl[3 * q + 0] = r * CXA + 4 * g;
            pl[3 * q + 1] = wrap_periodic(cxa + 4 * g, a.Ncc);  // Ncc % 4 == 0: a group never straddles the row end
            pl[3 * q + 2] = true_mod(rbase + r, a.Nrc);
        }
    };
    auto issue = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int sy = pl[3 * q + 2];
            const long long o = boff + (long long)sy * a.Ncc + pl[3 * q + 1];
            p[4 * q + 0] = *reinterpret_cast<const v4f*>(a.A + o);
            p[4 * q + 1] = *reinterpret_cast<const v4f*>(a.V + o);
            p[4 * q + 2] = *reinterpret_cast<const v4f*>(a.H + o);
            p[4 * q + 3] = *reinterpret_cast<const v4f*>(a.D + o);
            const int ny = sy + TY;
            pl[3 * q + 2] = ny >= a.Nrc ? ny - a.Nrc : ny;
        }
    };
    auto commit = [&](int tid) {
        const int* pl = PDWT_MINE(plan, tid);
        const v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            *reinterpret_cast<v4f*>(sA + pl[3 * q]) = p[4 * q + 0];
            *reinterpret_cast<v4f*>(sV + pl[3 * q]) = p[4 * q + 1];
            *reinterpret_cast<v4f*>(sH + pl[3 * q]) = p[4 * q + 2];
            *reinterpret_cast<v4f*>(sD + pl[3 * q]) = p[4 * q + 3];
        }
    };
    auto carry_read = [&](int tid) {  // buffer rows [TY, TY + D)
        v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            c[q] = lds_load16(buf + TY * OW + 2 * idx);
        }
    };
    auto carry_write = [&](int tid) {  // ... to rows [0, D)
        const v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            *reinterpret_cast<v4f*>(buf + 2 * idx) = c[q];
        }
    };

    // ---- row synthesis of the step's TY coefficient rows: (A,V) -> u1, (H,D) -> u2, both at full output width, into buffer
    // rows [D, D + TY).  Item = (row r, KB coefficient columns): a window of KB + D columns per band, read as 16-B groups,
    // the next group's reads issued before the current group's arithmetic.
    auto row_synth = [&](int tid, int first_row) {
        constexpr int NKB = TXC / KB, ITEMS = TY * NKB;
        constexpr int NQ = (PADL + KB + D + 3) / 4;  // 16-B groups per band
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            const int r = it / NKB;
            const int kb = it - r * NKB;
            if (it >= ITEMS || r < first_row) continue;  // (r: warm-up rows nobody needs)
            const int so = r * CXA + KB * kb;
            v2f u1[KB], u2[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) u1[kk] = u2[kk] = mk2(real_t(0), real_t(0));
            v4f w[2][4];
            w[0][0] = lds_load16(sA + so); w[0][1] = lds_load16(sV + so);
            w[0][2] = lds_load16(sH + so); w[0][3] = lds_load16(sD + so);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
                    w[(q + 1) & 1][0] = lds_load16(sA + so + 4 * (q + 1));
                    w[(q + 1) & 1][1] = lds_load16(sV + so + 4 * (q + 1));
                    w[(q + 1) & 1][2] = lds_load16(sH + so + 4 * (q + 1));
                    w[(q + 1) & 1][3] = lds_load16(sD + so + 4 * (q + 1));
                }
                v4f* cur = w[q & 1];
                lds_pin(cur[0]); lds_pin(cur[1]); lds_pin(cur[2]); lds_pin(cur[3]);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const v2f pa = h < 2 ? mk2(cur[0].x, cur[0].y) : mk2(cur[0].z, cur[0].w);
                    const v2f pv = h < 2 ? mk2(cur[1].x, cur[1].y) : mk2(cur[1].z, cur[1].w);
                    const v2f ph = h < 2 ? mk2(cur[2].x, cur[2].y) : mk2(cur[2].z, cur[2].w);
                    const v2f pd = h < 2 ? mk2(cur[3].x, cur[3].y) : mk2(cur[3].z, cur[3].w);
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk) {
                        const int i = 4 * q + h - PADL - kk;  // window position of this column for output pair kk
                        if (i >= 0 && i < WN) {
                            u1[kk] = syn_fma<HLEN>(i, h & 1, pa, tl[i < 0 || i >= WN ? 0 : i], u1[kk]);
                            u2[kk] = syn_fma<HLEN>(i, h & 1, ph, tl[i < 0 || i >= WN ? 0 : i], u2[kk]);
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk) {  // (a second sweep: two FMAs into one accumulator are never adjacent)
                        const int i = 4 * q + h - PADL - kk;
                        if (i >= 0 && i < WN) {
                            u1[kk] = syn_fma<HLEN>(i, h & 1, pv, th[i < 0 || i >= WN ? 0 : i], u1[kk]);
                            u2[kk] = syn_fma<HLEN>(i, h & 1, pd, th[i < 0 || i >= WN ? 0 : i], u2[kk]);
                        }
                    }
                }
            }
            real_t* dst = reinterpret_cast<real_t*>(buf + (D + r) * OW + 2 * KB * kb);
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {  // (u1,u2) of outputs 2k, 2k + 1: 4-B stores (pairs of them merge into ds_write2_b32)
                dst[4 * kk + 0] = u1[kk].x;
                dst[4 * kk + 2] = u1[kk].y;
                dst[4 * kk + 1] = u2[kk].x;
                dst[4 * kk + 3] = u2[kk].y;
            }
        }
    };

    // ---- column synthesis out of the buffer: item = (M output row pairs, XB output columns); the pair mm of chunk ch reads
    // buffer rows ch M + mm + i, i = 0 .. WN - 1.  The chunk index is the same in every lane (whole wavefronts per chunk):
    // row pointers and row guards are scalar.
    auto col_synth = [&](int tid, int t) {
        constexpr int NXQ = OW / XB, ITEMS = (TY / M) * NXQ;
        constexpr int NWIN = M + D;
        constexpr int GB = 6;                         // buffer rows read together, double-buffered
        constexpr int NG = (NWIN + GB - 1) / GB;
        constexpr bool UNI = NXQ % 64 == 0 && NT % 64 == 0;
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            if (it >= ITEMS) continue;
            int ch = it / NXQ;
            const int x = XB * (it - ch * NXQ);
            if (UNI) ch = wave_uniform(ch);
            const v2f* base = buf + ch * M * OW + x;
            v2f acc[M][XB];  // (out[2m], out[2m + 1]) of column x + xb
#pragma unroll
            for (int mm = 0; mm < M; ++mm)
#pragma unroll
                for (int xb = 0; xb < XB; ++xb) acc[mm][xb] = mk2(real_t(0), real_t(0));
            v4f w[2][GB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int e = 0; e < GB; ++e)
                    if (g * GB + e < NWIN) {
                        const v2f* src = base + (g * GB + e) * OW;
                        if (XB == 2) {
                            w[g & 1][e] = lds_load16(src);
                        } else {
                            const v2f one = *src;
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
                        if (XB == 2) lds_pin(w[g & 1][e]);
#pragma unroll
                        for (int xb = 0; xb < XB; ++xb) {
                            const v2f u = xb ? mk2(w[g & 1][e].z, w[g & 1][e].w) : mk2(w[g & 1][e].x, w[g & 1][e].y);
#pragma unroll
                            for (int mm = 0; mm < M; ++mm) {
                                const int ii = i - mm;
                                if (ii >= 0 && ii < WN) acc[mm][xb] = syn_fma<HLEN>(ii, false, u, tl[ii < 0 || ii >= WN ? 0 : ii], acc[mm][xb]);
                            }
#pragma unroll
                            for (int mm = 0; mm < M; ++mm) {
                                const int ii = i - mm;
                                if (ii >= 0 && ii < WN) acc[mm][xb] = syn_fma<HLEN>(ii, true, u, th[ii < 0 || ii >= WN ? 0 : ii], acc[mm][xb]);
                            }
                        }
                    }
                }
            }
            const int mrow = m0 + (t - W) * TY + ch * M;  // first output row pair of the item
            const int ox = 2 * k0 + x;
            if (ox + XB - 1 < a.Nc) {  // Nc and ox are even: an item is inside the row or outside
                const unsigned o0 = (unsigned)(2 * mrow) * (unsigned)a.Nc;  // scalar (ch is): row offsets ride in the soffset operand
                if (mrow + M <= m0 + nm && 2 * (mrow + M) <= a.Nr) {  // the whole item is inside: no row guards
#pragma unroll
                    for (int mm = 0; mm < M; ++mm)
#pragma unroll
                        for (int xb = 0; xb < XB; ++xb) {
                            st_lane(oplane, o0 + (unsigned)(2 * mm) * (unsigned)a.Nc, ox + xb, acc[mm][xb].x);
                            st_lane(oplane, o0 + (unsigned)(2 * mm + 1) * (unsigned)a.Nc, ox + xb, acc[mm][xb].y);
                        }
                } else {
#pragma unroll
                    for (int mm = 0; mm < M; ++mm) {
                        if (mrow + mm < m0 + nm) {
#pragma unroll
                            for (int xb = 0; xb < XB; ++xb) {
                                st_lane(oplane, o0 + (unsigned)(2 * mm) * (unsigned)a.Nc, ox + xb, acc[mm][xb].x);
                                if (2 * (mrow + mm) + 1 < a.Nr)
                                    st_lane(oplane, o0 + (unsigned)(2 * mm + 1) * (unsigned)a.Nc, ox + xb, acc[mm][xb].y);
                            }
                        }
                    }
                }
            }
        }
    };

    // Two barriers per step.  Phase B: the carry rows land at the top of the buffer, the row pass fills the rows below.
    // Phase C: the column pass; then, by every thread as it gets there, what the NEXT step needs -- the carry rows into
    // registers, the staged coefficients of step t + 1 into LDS (nobody reads the staging planes in this phase), the
    // global loads of step t + 2.  Nothing but arithmetic phases: two workgroups of a CU that run in step (they start
    // together and do the same work) do not both sit in a load / store phase.
    PDWT_FOR_THREADS(tid, NT) {
        make_plan(tid);
        if (!(PDWT_LONG_DIAG & 4)) {
            issue(tid);
            commit(tid);
            if (T > 1) issue(tid);
        }
    }
    PDWT_LONG_SYNC();
#if PDWT_LONG_DIAG & 32
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
    const unsigned long long t0_ = last_;
#endif
    for (int t = 0; t < T; ++t) {
        PDWT_FOR_THREADS(tid, NT) {
            if (t > 0 && D > 0 && !(PDWT_LONG_DIAG & 8)) carry_write(tid);
            if (!(PDWT_LONG_DIAG & 1)) row_synth(tid, t == 0 ? G::SKIP : 0);
        }
        PDWT_LONG_CLK(0);
        PDWT_LONG_SYNC();
        PDWT_LONG_CLK(1);
        PDWT_FOR_THREADS(tid, NT) {
            if (t >= W && !(PDWT_LONG_DIAG & 2)) col_synth(tid, t);
            PDWT_LONG_CLK(2);
            if (t + 1 < T) {
                if (D > 0 && !(PDWT_LONG_DIAG & 8)) carry_read(tid);
                if (!(PDWT_LONG_DIAG & 4)) {
                    commit(tid);
                    if (t + 2 < T) issue(tid);
                }
            }
        }
        PDWT_LONG_CLK(3);
        if (t + 1 < T) PDWT_LONG_SYNC();
        PDWT_LONG_CLK(4);
    }
#if PDWT_LONG_DIAG & 32
    if ((threadIdx.x & 63) == 0) {
        const int wv = (blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + threadIdx.x / 64;
        for (int k = 0; k < 5; ++k) pdwt_long_prof[8 * wv + k] = prof_[k];
        pdwt_long_prof[8 * wv + 5] = clock64() - t0_;
        pdwt_long_prof[8 * wv + 6] = t0_;
        pdwt_long_prof[8 * wv + 7] = T;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// out[k] = sum_j x[2k - C + j] f[hlen - 1 - j], C = hlen / 2 - 1 (oracle_analysis_rows; pdwt/src/separable.cu:91-131).
template <int HLEN, int TXC, int TY>
struct FwdLongGeom {
    static constexpr int C = HLEN / 2 - 1;
    static constexpr int D = HLEN - 2;                   // input rows of history an output row needs
    static constexpr int R2 = 2 * TY;                    // input rows per step
    static constexpr int W = long_cdiv(D, R2);           // warm-up steps: row analysis only
    static constexpr int SKIP = W * R2 - D;              // input rows of step 0 that no output of the segment needs
    static constexpr int BR = D + R2;                    // buffer rows of (L,H) pairs: carry + new
    static constexpr int PADL = (4 - (C & 3)) & 3;       // samples between the aligned load origin and column 2 k0 - C
    static constexpr int RXA = (PADL + 2 * TXC + D + 3) & ~3;  // staged samples per row
    static constexpr int LDS_REALS = R2 * RXA + 2 * BR * TXC;
};

//   KB  output columns per row-analysis item;  M output rows per column-analysis item, XB output columns per item (1 or 2)
template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB>
PDWT_DEVICE void dwt2_fwd_long_wg(const FwdLongArgs& a, int strip, int seg, int bz, real_t* smem) {
    using G = FwdLongGeom<HLEN, TXC, TY>;
    constexpr int C = G::C, D = G::D, R2 = G::R2, W = G::W, PADL = G::PADL, RXA = G::RXA;
    static_assert(TXC % KB == 0 && !(KB & 1), "row-analysis items tile the strip in whole 16-B groups");
    static_assert(TY % M == 0 && TXC % XB == 0 && (XB == 1 || XB == 2), "column-analysis items tile the step");
    constexpr int V4 = RXA / 4, TOTAL = R2 * V4, TRIPS = (TOTAL + NT - 1) / NT;
    constexpr int CARRY = D * TXC / 2, CTRIPS = (CARRY + NT - 1) / NT;

    real_t* sIn = smem;                                     // R2 x RXA samples of the step's input rows
    v2f* buf = reinterpret_cast<v2f*>(smem + R2 * RXA);    // BR x TXC (L,H) pairs

    const int k0 = strip * TXC, m0 = seg * a.seg;
    const int nm = a.Nr2 - m0 < a.seg ? a.Nr2 - m0 : a.seg;  // output rows of this segment
    if (nm <= 0) return;
    const int T = W + (nm + TY - 1) / TY;
    const int rbase = 2 * m0 - C + D - W * R2;  // first input row of step 0
    const int xa = 2 * k0 - C - PADL;           // multiple of 4
    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const long long boff = (long long)bz * a.out_bstride;
    const LanePlane pA = lane_plane(a.A + boff), pH = lane_plane(a.H + boff), pV = lane_plane(a.V + boff), pD = lane_plane(a.D + boff);
    v2f tv[HLEN];  // the (lo, hi) tap pairs in vector registers, see the inverse
#pragma unroll
    for (int j = 0; j < HLEN; ++j) tv[j] = in_vgprs(a.fb.t[j]);

    PDWT_PER_THREAD(int, plan, 3 * TRIPS, NT);  // LDS offset, source column, source row (advanced by R2 per step; Nr >= R2)
    PDWT_PER_THREAD(v4f, pre, TRIPS, NT);
    PDWT_PER_THREAD(v4f, car, CTRIPS > 0 ? CTRIPS : 1, NT);
    auto make_plan = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < TOTAL ? idx : TOTAL - 1;
            const int r = idx / V4;
            const int g = idx - r * V4;
            pl[3 * q + 0] = r * RXA + 4 * g;
            pl[3 * q + 1] = wrap_periodic(xa + 4 * g, a.Nc);  // Nc % 4 == 0
            pl[3 * q + 2] = true_mod(rbase + r, a.Nr);        // Nr even: plain periodization
        }
    };
    auto issue = [&](int tid) {
        int* pl = PDWT_MINE(plan, tid);
        v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int sy = pl[3 * q + 2];
            p[q] = *reinterpret_cast<const v4f*>(in + (long long)sy * a.Nc + pl[3 * q + 1]);
            const int ny = sy + R2;
            pl[3 * q + 2] = ny >= a.Nr ? ny - a.Nr : ny;
        }
    };
    auto commit = [&](int tid) {
        const int* pl = PDWT_MINE(plan, tid);
        const v4f* p = PDWT_MINE(pre, tid);
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) *reinterpret_cast<v4f*>(sIn + pl[3 * q]) = p[q];
    };
    auto carry_read = [&](int tid) {  // buffer rows [R2, R2 + D)
        v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            c[q] = lds_load16(buf + R2 * TXC + 2 * idx);
        }
    };
    auto carry_write = [&](int tid) {  // ... to rows [0, D)
        const v4f* c = PDWT_MINE(car, tid);
#pragma unroll
        for (int q = 0; q < CTRIPS; ++q) {
            int idx = tid + q * NT;
            idx = idx < CARRY ? idx : CARRY - 1;
            *reinterpret_cast<v4f*>(buf + 2 * idx) = c[q];
        }
    };

    // ---- row analysis of the step's R2 input rows into buffer rows [D, D + R2): item = (row, KB output columns); a sample
    // is broadcast against the (lo, hi) tap pair
    auto row_ana = [&](int tid, int first_row) {
        constexpr int NKB = TXC / KB, ITEMS = R2 * NKB;
        constexpr int NQ = (PADL + 2 * (KB - 1) + HLEN + 3) / 4;  // 16-B groups read per item
        constexpr int GB = 3, NG = (NQ + GB - 1) / GB;
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            const int r = it / NKB;
            const int kb = it - r * NKB;
            if (it >= ITEMS || r < first_row) continue;
            const real_t* p4 = sIn + r * RXA + 2 * KB * kb;
            v2f acc[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) acc[kk] = mk2(real_t(0), real_t(0));
            v4f w[2][GB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int e = 0; e < GB; ++e)
                    if (g * GB + e < NQ) w[g & 1][e] = lds_load16(p4 + 4 * (g * GB + e));
            };
            load_group(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1);
#pragma unroll
                for (int e = 0; e < GB; ++e) {
                    if (g * GB + e < NQ) {
                        v4f& cur = w[g & 1][e];
                        lds_pin(cur);
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const int f = 4 * (g * GB + e) + h - PADL;  // window sample
                            const v2f pr = h < 2 ? mk2(cur.x, cur.y) : mk2(cur.z, cur.w);
#pragma unroll
                            for (int kk = 0; kk < KB; ++kk) {
                                const int j = f - 2 * kk;
                                if (j >= 0 && j < HLEN) {
                                    const v2f tap = tv[HLEN - 1 - (j < 0 || j >= HLEN ? 0 : j)];
                                    acc[kk] = (h & 1) ? fma2_by_v(pr, tap, acc[kk]) : fma2_bx_v(pr, tap, acc[kk]);
                                }
                            }
                        }
                    }
                }
            }
            v2f* dst = buf + (D + r) * TXC + KB * kb;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 2) {
                real4_t w4;
                w4.x = acc[kk].x; w4.y = acc[kk].y; w4.z = acc[kk + 1].x; w4.w = acc[kk + 1].y;
                *reinterpret_cast<real4_t*>(dst + kk) = w4;
            }
        }
    };

    // ---- column analysis out of the buffer: item = (M output rows, XB output columns); output row mm of chunk ch reads
    // buffer rows 2 (ch M + mm) + j.  L x (lo, hi) -> (A, H), H x (lo, hi) -> (V, D): the DATA half is broadcast, the tap pair
    // is taken as it lies in the SGPRs (broadcasting a tap half instead makes hipcc build a second SGPR pair per tap: 729
    // v_readlane of spilled scalars in the first build of this kernel).  Whole wavefronts per chunk: scalar row pointers.
    auto col_ana = [&](int tid, int t) {
        constexpr int NXQ = TXC / XB, ITEMS = (TY / M) * NXQ;
        constexpr int NWIN = 2 * (M - 1) + HLEN;
        constexpr int GB = 6, NG = (NWIN + GB - 1) / GB;
        constexpr bool UNI = NXQ % 64 == 0 && NT % 64 == 0;
        PDWT_LONG_ITEMS(it, tid, ITEMS, NT) {
            if (it >= ITEMS) continue;
            int ch = it / NXQ;
            const int x = XB * (it - ch * NXQ);
            if (UNI) ch = wave_uniform(ch);
            const v2f* base = buf + 2 * ch * M * TXC + x;
            v2f accAH[M][XB], accVD[M][XB];
#pragma unroll
            for (int mm = 0; mm < M; ++mm)
#pragma unroll
                for (int xb = 0; xb < XB; ++xb) accAH[mm][xb] = accVD[mm][xb] = mk2(real_t(0), real_t(0));
            v4f w[2][GB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int e = 0; e < GB; ++e)
                    if (g * GB + e < NWIN) {
                        const v2f* src = base + (g * GB + e) * TXC;
                        if (XB == 2) {
                            w[g & 1][e] = lds_load16(src);
                        } else {
                            const v2f one = *src;
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
                        if (XB == 2) lds_pin(w[g & 1][e]);
#pragma unroll
                        for (int xb = 0; xb < XB; ++xb) {
                            const v2f lh = xb ? mk2(w[g & 1][e].z, w[g & 1][e].w) : mk2(w[g & 1][e].x, w[g & 1][e].y);
#pragma unroll
                            for (int mm = 0; mm < M; ++mm) {
                                const int j = i - 2 * mm;
                                if (j >= 0 && j < HLEN) {
                                    const v2f tap = tv[HLEN - 1 - (j < 0 || j >= HLEN ? 0 : j)];
                                    accAH[mm][xb] = fma2_bx_v(lh, tap, accAH[mm][xb]);
                                    accVD[mm][xb] = fma2_by_v(lh, tap, accVD[mm][xb]);
                                }
                            }
                        }
                    }
                }
            }
            const int mrow = m0 + (t - W) * TY + ch * M;
            const int ox = k0 + x;
            if (ox + XB - 1 < a.Nc2) {  // Nc2 and ox are even when XB == 2
                const unsigned o0 = (unsigned)mrow * (unsigned)a.Nc2;  // scalar: ch is
#pragma unroll
                for (int mm = 0; mm < M; ++mm) {
                    if (mrow + mm < m0 + nm) {
#pragma unroll
                        for (int xb = 0; xb < XB; ++xb) {
                            st_lane(pA, o0 + (unsigned)mm * (unsigned)a.Nc2, ox + xb, accAH[mm][xb].x);
                            st_lane(pH, o0 + (unsigned)mm * (unsigned)a.Nc2, ox + xb, accAH[mm][xb].y);
                            st_lane(pV, o0 + (unsigned)mm * (unsigned)a.Nc2, ox + xb, accVD[mm][xb].x);
                            st_lane(pD, o0 + (unsigned)mm * (unsigned)a.Nc2, ox + xb, accVD[mm][xb].y);
                        }
                    }
                }
            }
        }
    };

    // two barriers per step, see the inverse
    PDWT_FOR_THREADS(tid, NT) {
        make_plan(tid);
        if (!(PDWT_LONG_DIAG & 4)) {
            issue(tid);
            commit(tid);
            if (T > 1) issue(tid);
        }
    }
    PDWT_LONG_SYNC();
    for (int t = 0; t < T; ++t) {
        PDWT_FOR_THREADS(tid, NT) {
            if (t > 0 && !(PDWT_LONG_DIAG & 8)) carry_write(tid);
            if (!(PDWT_LONG_DIAG & 1)) row_ana(tid, t == 0 ? G::SKIP : 0);
        }
        PDWT_LONG_SYNC();
        PDWT_FOR_THREADS(tid, NT) {
            if (t >= W && !(PDWT_LONG_DIAG & 2)) col_ana(tid, t);
            if (t + 1 < T) {
                if (!(PDWT_LONG_DIAG & 8)) carry_read(tid);
                if (!(PDWT_LONG_DIAG & 4)) {
                    commit(tid);
                    if (t + 2 < T) issue(tid);
                }
            }
        }
        if (t + 1 < T) PDWT_LONG_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
// Block numbering: xcd_tile gives each XCD a contiguous range of (segment, strip) pairs, strips fastest: horizontally
// adjacent strips (which share their x halo) and the segments below them meet in one L2.
template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB, int MINB>
__global__ void __launch_bounds__(NT, MINB) dwt2_inv_long_kernel(const InvLongArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, seg;
    if (!xcd_tile(blockIdx.x, a.strips, a.segs, strip, seg)) return;
    dwt2_inv_long_wg<HLEN, TXC, TY, NT, KB, M, XB>(a, strip, seg, blockIdx.y, pdwt_smem);
}

template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB, int MINB>
__global__ void __launch_bounds__(NT, MINB) dwt2_fwd_long_kernel(const FwdLongArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int strip, seg;
    if (!xcd_tile(blockIdx.x, a.strips, a.segs, strip, seg)) return;
    dwt2_fwd_long_wg<HLEN, TXC, TY, NT, KB, M, XB>(a, strip, seg, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
