// dwt1_reg_kernels.hpp -- up to THREE levels of a (batched) 1D DWT per launch, entirely in registers (gfx950).
//
// Why a third 1D scheme: the workgroup-wide fused pyramid (dwt1_fused_kernels.hpp) executes 2.6x the packed
// FMAs the arithmetic needs (every tap window is re-read from LDS by every work item, 1.1e7 vector and 1e7
// scalar wavefront instructions per launch on 2^24 samples, rocprofv3 --pmc) and is instruction-bound at
// 0.46-0.51 of the HBM roofline; the one-wavefront cascade with LDS rings (dwt1_wave_kernels.hpp) has the
// same problem.  Here nothing is staged anywhere:
//   * a lane owns 16 CONSECUTIVE samples (four 16-B loads, 64 B per lane, 4 KiB contiguous per wavefront);
//     a "block" is the 1024 samples of the 64 lanes;
//   * a tap window of level k spans the lane's own n_k = 16 >> (k-1) values and hlen-2 (+1) values of the
//     next lane(s): those are fetched with DPP wave_shl:1 register moves (one per value; a value two lanes
//     away is the shift applied to the already shifted copy), 15 moves against 128 / 64 / 32 packed FMAs
//     at levels 1 / 2 / 3 of a 16-tap filter;
//   * every multiply-add is a v_pk_fma_f32 on (approximation, detail) pairs with the (lo, hi) tap pair in
//     SGPRs; the approximation of level k stays in the lane's registers as the input of level k+1;
//   * lanes at the top of the block lack their neighbours: after K levels V = 64 - sum_k ceil(E_k / n_k)
//     lanes hold valid results (57 for hlen 16, K = 3) and blocks advance by 16 V samples -- the overlap
//     (11 %) is re-read from L1/L2 and recomputed, there is no state carried between blocks, no LDS and no
//     barrier.  Stores are range-checked raw-buffer stores: the invalid lanes carry an offset no
//     descriptor accepts, so every vector-memory instruction of the block loop is unconditional and the
//     next block's loads stay in flight under exact s_waitcnt counts.
// More than three levels = the kernel again on the approximation it left (2^-3 of the samples).
//
// Index bookkeeping (c = hlen/2 - 1, everything compile-time): the window of the lane's output q at level k
// starts at its local input e_k + 2q, e_k chosen so that the GLOBAL start 2 o - c has the right parity:
//   e_k = (c + d_{k-1}) & 1,   d_k = (e_k + d_{k-1} + c) / 2,   d_0 = 0,
// and the lane's outputs are the global indices  (base >> k) + (16 >> k) lane + q + d_k  (base = first
// sample of the block, a multiple of 16).  E_k = hlen - 2 + e_k values come from the next lanes.
// Arithmetic as everywhere else: out[o] = sum_j x[2o - c + j] f[hlen-1-j] (pdwt/src/separable.cu:91-131),
// periodic; exact for the reference's per-level periodization when every level length is even.
//
// The inverse runs the same scheme backwards: a lane ends with 16 consecutive output samples, the
// coefficient windows of level k start at the lane's own first coefficient (ownership offsets o_k below)
// and extend hlen/2 - 1 (+1) coefficients into the next lanes; (approximation, detail) pairs are multiplied
// by (rec_lo, rec_hi) tap pairs, the two sums are added at the end like the reference does
// (separable.cu:293-328, polyphase form restated in oracle/pdwt_oracle.c).
#pragma once

#include "dwt2_wave_kernels.hpp"  // WaveReg, DPP shifts, RowBuf, wave_ld16 / wave_ld8
#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

// all outstanding vector-memory operations of the wavefront (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15)
#ifdef PDWT_CPU_EMU
#define PDWT_WAIT_VMEM() ((void)0)
#else
#define PDWT_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

constexpr int kReg1MaxLevels = 3;
constexpr int kReg1MaxHlen = 20;          // E_k <= 19: at most five lanes ahead at level 3.  (Round 4 instantiated 22-40 taps too -- the code is
                                          // generic --: 8-10 % ahead of the LDS pyramids on ONE row of 2^24 samples, 3-15 % behind on
                                          // 4096 rows of 4096 and on rows of <= 2^20: not dispatched, profiles/r04u_reg1d_long.txt)
constexpr int kReg1LdsStride = 20;            // floats per lane of the wavefront-private transposition tile
constexpr int kReg1LdsFloats = 64 * kReg1LdsStride;
constexpr unsigned kReg1Dropped = 0x80000000u;  // byte offset >= any row (rows are < 2 GiB); stays out of range
                                                // while block strides are added to it

struct Reg1Geom {
    int e[kReg1MaxLevels + 1], d[kReg1MaxLevels + 1], E[kReg1MaxLevels + 1], m[kReg1MaxLevels + 1];
    int V;
};

// forward: see the header comment
constexpr Reg1Geom reg1_fwd_geom(int hlen, int K) {
    Reg1Geom g{};
    const int c = hlen / 2 - 1;
    g.d[0] = 0;
    g.V = 64;
    for (int k = 1; k <= K; ++k) {
        const int n = 16 >> (k - 1);
        // Output q of the lane is global output (n/2) L + d[k] + q and reads the lane's window from e[k] = 2 d[k] - c - d[k-1]
        // on.  The smallest admissible d[k] (e[k] = 0 or 1) leaves the lane's n/2 outputs at an ODD offset for half of the
        // filter lengths, i.e. 4-B stores: levels 1-3 of 2^24 samples took 37.9 / 33.4 / 35.8 / 40.2 / 42.8 us for 4 / 6 /
        // 12 / 14 / 20 taps next to 28.1 / 27.9 / 27.0 / 27.4 for 8 / 10 / 16 / 18 (found in round 4, tools/cliffs.py).
        // d[1] is therefore rounded up to a multiple of 4 (16-B stores of the level-1 details, half of all bytes written): the
        // window moves up by at most 7 samples (a few more lane shifts) and no filter length loses a valid lane except 6
        // taps (61 -> 60).  Aligning levels 2 and 3 as well costs one or two valid lanes and measured SLOWER in the step
        // (sym8 2^24 L6 forward+inverse 57.4 -> 60.2 us, db2 54.4 -> 56.6; profiles/r04o_reg1_align_ab.txt).
#ifndef PDWT_REG1_ALIGN
#define PDWT_REG1_ALIGN 1  // 0: smallest offset (rounds 2-3), 1: level 1 only, 2: every level
#endif
        const int al = PDWT_REG1_ALIGN == 0 ? 1 : ((PDWT_REG1_ALIGN == 1 && k > 1) ? 1 : (n / 2 < 4 ? n / 2 : 4));
        g.d[k] = ((c + g.d[k - 1] + 1) / 2 + al - 1) / al * al;
        g.e[k] = 2 * g.d[k] - c - g.d[k - 1];
        g.E[k] = hlen - 2 + g.e[k];
        g.m[k] = (g.E[k] + n - 1) / n;
        g.V -= g.m[k];
    }
    return g;
}

// inverse: level k turns the lane's n_k = 16 >> k (A_k, D_k) pairs (+ E_k of the next lanes) into its
// 2 n_k values of A_{k-1}.  With h2 = hlen/2, c = h2/2, s = (h2 even): output g reads the coefficients
// floor((g + s)/2) - c + j, j < h2.  The lane owns the level-k coefficients (16 >> k) L + o_k + i
// (L = global lane number), o_0 = the output offset, o_k = floor((o_{k-1} + s)/2) - c; here d[k] = o_k,
// e[k] = (o_{k-1} + s) & 1 (parity of the lane's first output), E[k] = h2 - 1 + e[k].
constexpr int reg1_floor_half(int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); }
constexpr Reg1Geom reg1_inv_geom(int hlen, int K, int o0) {
    Reg1Geom g{};
    const int h2 = hlen / 2, c = h2 / 2, s = (h2 & 1) ? 0 : 1;
    g.d[0] = o0;
    g.V = 64;
    for (int k = 1; k <= K; ++k) {
        const int n = 16 >> k;
        g.e[k] = (g.d[k - 1] + s) & 1;
        g.d[k] = reg1_floor_half(g.d[k - 1] + s) - c;
        g.E[k] = h2 - 1 + g.e[k];
        g.m[k] = (g.E[k] + n - 1) / n;
        g.V -= g.m[k];
    }
    return g;
}
// Output offset of the inverse: the smallest multiple of 16 for which no ownership offset is negative (so
// that only the LAST blocks of a row wrap, like in the forward kernel).
constexpr int reg1_inv_o0(int hlen, int K) {
    for (int o0 = 0; o0 <= 1024; o0 += 16) {
        const Reg1Geom g = reg1_inv_geom(hlen, K, o0);
        bool ok = true;
        for (int k = 1; k <= K; ++k) ok = ok && g.d[k] >= 0;
        if (ok) return o0;
    }
    return -1;
}

struct Fwd1DRegArgs {
    const real_t* in;                 // (rows, N0)
    real_t* det[kReg1MaxLevels];      // det[k-1] = D_k: (rows, N0 >> k)
    real_t* app;                      // A_K: (rows, N0 >> K)
    int rows, N0;
    int nblk;                        // blocks per row: ceil(N0 / (16 V))
    int nplain;                      // blocks [0, nplain) of a row touch no index beyond the row (no wrap arithmetic)
    int bpw;                         // blocks per wavefront
    int wpr;                         // wavefronts per row: ceil(nblk / bpw)
    FilterBankI fb;                  // (dec_lo, dec_hi)
};

struct Inv1DRegArgs {
    const real_t* app;                // A_K
    const real_t* det[kReg1MaxLevels];
    real_t* out;                      // (rows, N0)
    int rows, N0;
    int nblk, nplain, bpw, wpr;
    FilterBankI fb;                  // (rec_lo, rec_hi)
};

// Host: block geometry of a row of N0 samples.  A block is "plain" when none of its loads or (valid-lane) stores
// reaches past the end of its row at any level, so that the kernel needs no wrap arithmetic for it; the blocks
// at the end of a row wrap periodically (GUARD variants).
inline void reg1_fwd_blocks(int hlen, int K, int N0, int* nblk, int* nplain) {
    const Reg1Geom g = reg1_fwd_geom(hlen, K);
    const long long step = 16LL * g.V;
    *nblk = (int)((N0 + step - 1) / step);
    auto plain = [&](int b) {
        const long long base = b * step;
        bool ok = base + 1024 <= N0;
        for (int k = 1; k <= K && ok; ++k) ok = (base >> k) + (16 >> k) * (long long)g.V + g.d[k] <= (N0 >> k);
        return ok;
    };
    int np = *nblk;  // the conditions are monotone in the block number: walk back from the end (a few blocks)
    while (np > 0 && !plain(np - 1)) --np;
    *nplain = np;
}

// range-checked store of four values (16 B; fp64: two 16-B halves) -- see RowBuf in dwt2_wave_kernels.hpp
#ifdef PDWT_CPU_EMU
PDWT_DEVICE void row_st16(const RowBuf& r, unsigned off, real_t x, real_t y, real_t z, real_t w) {
    if (off < r.bytes) { real_t* p = reinterpret_cast<real_t*>(r.base + off); p[0] = x; p[1] = y; p[2] = z; p[3] = w; }
}
#else
static __device__ __forceinline__ void row_st16(const RowBuf& r, unsigned off, float x, float y, float z, float w) {
    pdwt_u4 d;
    d.x = __builtin_bit_cast(unsigned, x);
    d.y = __builtin_bit_cast(unsigned, y);
    d.z = __builtin_bit_cast(unsigned, z);
    d.w = __builtin_bit_cast(unsigned, w);
    __builtin_amdgcn_raw_buffer_store_b128(d, r.rsrc, (int)off, 0, 0);
}
static __device__ __forceinline__ void row_st16(const RowBuf& r, unsigned off, double x, double y, double z, double w) {
    row_st8(r, off, x, y);  // an out-of-range offset (kReg1Dropped) stays out of range 16 B further on
    row_st8(r, off + 16u, z, w);
}
#endif

// ext[N .. N+E) = the first E values of the following lanes (lane + m holds ext[m N + i] = its own value i)
template <int N, int E, int NE>
PDWT_DEVICE void reg1_extend(WaveReg<real_t, NE>& ext) {
    static_assert(N + E <= NE, "extension fits");
#pragma unroll
    for (int m = 1; (m - 1) * N < E; ++m) {
        PDWT_WAVE_LANES(lane) {
            real_t* v = ext.mine(lane);
#pragma unroll
            for (int i = 0; i < N; ++i)
                if ((m - 1) * N + i < E) v[m * N + i] = ext.from_next((m - 1) * N + i, lane, 0.f);
        }
    }
}

// n consecutive values of one band row, first global index `idx` (per lane), in units of U floats; GUARD wraps each
// unit at the row length Nk (units never straddle it); `ok` false -> nothing is stored (out-of-range offset)
template <int NV, int U, bool GUARD>
PDWT_DEVICE void reg1_store(const RowBuf& rb, const real_t* v, int idx, int Nk, bool ok) {
#pragma unroll
    for (int j = 0; j < NV / U; ++j) {
        int p = idx + U * j;
        if (GUARD && p >= Nk) p -= Nk;
        const unsigned off = ok ? kRealBytes * (unsigned)p : kReg1Dropped;
        if (U == 4) row_st16(rb, off, v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        else if (U == 2) row_st8(rb, off, v[2 * j], v[2 * j + 1]);
        else row_st4(rb, off, v[j]);
    }
}
constexpr int reg1_unit(int n, int d) { return (n % 4 == 0 && d % 4 == 0) ? 4 : ((n % 2 == 0 && d % 2 == 0) ? 2 : 1); }

// ---------------------------------------------------------------------------------------------- forward
template <int HLEN, int K>
struct Fwd1DRegState {
    WaveReg<real_t, 16> x;     // the block being transformed
    WaveReg<real_t, 16> xn;    // the next block's samples, in flight
};

template <bool GUARD>
PDWT_DEVICE void fwd1d_reg_load(WaveReg<real_t, 16>& x, const real_t* row, long long base, int N0) {
    PDWT_WAVE_LANES(lane) {
        long long s = base + 16 * lane;
        if (GUARD) s %= N0;  // N0 % 16 == 0: the lane's 64 B never straddle the end of the row
        const unsigned off = kRealBytes * (unsigned)s;
        real_t* v = x.mine(lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4f w = wave_ld16(row, off + 4u * kRealBytes * i);
            v[4 * i] = w.x; v[4 * i + 1] = w.y; v[4 * i + 2] = w.z; v[4 * i + 3] = w.w;
        }
    }
}

// one level: `src` (NIN values per lane) -> A (NIN/2 per lane, the next level's input), details to `outD`
// (and A to `outA` at the last level)
template <int HLEN, int K, int LV, bool GUARD, int NIN>
PDWT_DEVICE void fwd1d_reg_level(const Fwd1DRegArgs& a, WaveReg<real_t, NIN>& src, WaveReg<real_t, NIN / 2>& A,
                                 long long base, long long row) {
    constexpr Reg1Geom g = reg1_fwd_geom(HLEN, K);
    constexpr int E = g.E[LV], e = g.e[LV], d = g.d[LV], NO = NIN / 2, NE = NIN + E;
    WaveReg<real_t, NE> ext;
    PDWT_WAVE_LANES(lane) {
        real_t* v = ext.mine(lane);
        const real_t* s = src.mine(lane);
#pragma unroll
        for (int i = 0; i < NIN; ++i) v[i] = s[i];
    }
    reg1_extend<NIN, E, NE>(ext);
    const int Nk = a.N0 >> LV;
    const RowBuf bD = row_buf(a.det[LV - 1] + row * Nk, kRealBytes * (unsigned)Nk);
    const RowBuf bA = row_buf((LV == K ? a.app : a.det[LV - 1]) + row * Nk, kRealBytes * (unsigned)Nk);  // used at the last level only
    PDWT_WAVE_LANES(lane) {
        const real_t* v = ext.mine(lane);
        real_t* av = A.mine(lane);
        real_t dv[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            v2f acc = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < HLEN; ++j) acc = fma2(bc(v[e + 2 * q + j]), a.fb.t[HLEN - 1 - j], acc);
            av[q] = acc.x;
            dv[q] = acc.y;
        }
        int idx = (int)(base >> LV) + NO * lane + d;
        if (GUARD) idx %= Nk;
        constexpr int U = reg1_unit(NO, d);
        const bool ok = lane < g.V;
        reg1_store<NO, U, GUARD>(bD, dv, idx, Nk, ok);
        if (LV == K) reg1_store<NO, U, GUARD>(bA, av, idx, Nk, ok);
    }
}

template <int HLEN, int K, bool GUARD>
PDWT_DEVICE void fwd1d_reg_block(const Fwd1DRegArgs& a, WaveReg<real_t, 16>& x, long long base, long long row) {
    WaveReg<real_t, 8> a1;
    fwd1d_reg_level<HLEN, K, 1, GUARD, 16>(a, x, a1, base, row);
    if constexpr (K >= 2) {
        WaveReg<real_t, 4> a2;
        fwd1d_reg_level<HLEN, K, 2, GUARD, 8>(a, a1, a2, base, row);
        if constexpr (K >= 3) {
            WaveReg<real_t, 2> a3;
            fwd1d_reg_level<HLEN, K, 3, GUARD, 4>(a, a2, a3, base, row);
        }
    }
}

// wavefront `w` of the launch: blocks [first, first + count) of row w / wpr
template <int HLEN, int K>
PDWT_DEVICE void dwt1_fwd_reg(const Fwd1DRegArgs& a, long long w) {
    constexpr Reg1Geom g = reg1_fwd_geom(HLEN, K);
    constexpr int STEP = 16 * g.V;
    const long long row = w / a.wpr;
    if (row >= a.rows) return;
    const int first = (int)(w - row * a.wpr) * a.bpw;
    const int last = first + a.bpw < a.nblk ? first + a.bpw : a.nblk;  // exclusive
    const int plain_end = last < a.nplain ? last : a.nplain;
    const real_t* rin = a.in + row * a.N0;
    Fwd1DRegState<HLEN, K> st;
    if (first < plain_end) {
        fwd1d_reg_load<false>(st.xn, rin, (long long)first * STEP, a.N0);
        PDWT_WAIT_VMEM();  // see dwt1_inv_reg
        for (int b = first; b < plain_end; ++b) {
            PDWT_WAVE_LANES(lane) {
                real_t* v = st.x.mine(lane);
                const real_t* n = st.xn.mine(lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = n[i];
            }
            const int nb = b + 1 < plain_end ? b + 1 : b;  // the last trip re-requests its own block: no branch
            fwd1d_reg_load<false>(st.xn, rin, (long long)nb * STEP, a.N0);
            PDWT_ROW_FENCE();  // the next block's loads stay in front of this block's arithmetic
            fwd1d_reg_block<HLEN, K, false>(a, st.x, (long long)b * STEP, row);
        }
    }
    for (int b = first > plain_end ? first : plain_end; b < last; ++b) {
        fwd1d_reg_load<true>(st.x, rin, (long long)b * STEP, a.N0);
        fwd1d_reg_block<HLEN, K, true>(a, st.x, (long long)b * STEP, row);
    }
}

// ---------------------------------------------------------------------------------------------- inverse
// The 16 input values of a lane and block: [D_1 (8) | D_2 (4) | D_3 (2) | A_3 (2)] for K = 3, [D_1 | D_2 (4) | A_2 (4)]
// for K = 2, [D_1 (8) | A_1 (8)] for K = 1.
constexpr int reg1_inv_slot(int k) { return 16 - (16 >> (k - 1)); }  // D_k; A_K sits at reg1_inv_slot(K + 1)

inline void reg1_inv_blocks(int hlen, int K, int N0, int* nblk, int* nplain) {
    const int o0 = reg1_inv_o0(hlen, K);
    const Reg1Geom g = reg1_inv_geom(hlen, K, o0);
    const long long step = 16LL * g.V;
    *nblk = (int)((N0 + step - 1) / step);
    auto plain = [&](int b) {
        const long long base = b * step;
        bool ok = base + o0 + 1024 <= N0;
        for (int k = 1; k <= K && ok; ++k) ok = (base >> k) + (16 >> k) * 64LL + g.d[k] <= (N0 >> k);
        return ok;
    };
    int np = *nblk;
    while (np > 0 && !plain(np - 1)) --np;
    *nplain = np;
}

PDWT_DEVICE real_t wave_ld4(const real_t* base, unsigned byte_off) {
    return *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

// n consecutive values of one band row from global index idx (per lane), in units of U floats
template <int NV, int U, bool GUARD>
PDWT_DEVICE void reg1_load(const real_t* row, real_t* v, int idx, int Nk) {
#pragma unroll
    for (int j = 0; j < NV / U; ++j) {
        int p = idx + U * j;
        if (GUARD && p >= Nk) p -= Nk;
        const unsigned off = kRealBytes * (unsigned)p;
        if (U == 4) {
            const v4f w = wave_ld16(row, off);
            v[4 * j] = w.x; v[4 * j + 1] = w.y; v[4 * j + 2] = w.z; v[4 * j + 3] = w.w;
        } else if (U == 2) {
            const v2f w = wave_ld8(row, off);
            v[2 * j] = w.x; v[2 * j + 1] = w.y;
        } else {
            v[j] = wave_ld4(row, off);
        }
    }
}

template <int HLEN, int K, bool GUARD>
PDWT_DEVICE void inv1d_reg_load(const Inv1DRegArgs& a, WaveReg<real_t, 16>& in, long long base, long long row) {
    constexpr Reg1Geom g = reg1_inv_geom(HLEN, K, reg1_inv_o0(HLEN, K));
    PDWT_WAVE_LANES(lane) {
        real_t* v = in.mine(lane);
#pragma unroll
        for (int k = 1; k <= K; ++k) {
            const int n = 16 >> k, Nk = a.N0 >> k;
            int idx = (int)(base >> k) + n * lane + g.d[k];
            if (GUARD) idx %= Nk;
            // unit of the access: the ownership offset and the count decide the alignment (compile time per level)
            if (k == 1) reg1_load<8, reg1_unit(8, g.d[1]), GUARD>(a.det[0] + row * Nk, v + reg1_inv_slot(1), idx, Nk);
            if (k == 2) reg1_load<4, reg1_unit(4, g.d[2]), GUARD>(a.det[1] + row * Nk, v + reg1_inv_slot(2), idx, Nk);
            if (k == 3) reg1_load<2, reg1_unit(2, g.d[3]), GUARD>(a.det[2] + row * Nk, v + reg1_inv_slot(3), idx, Nk);
            if (k == K) {
                if (K == 1) reg1_load<8, reg1_unit(8, g.d[1]), GUARD>(a.app + row * Nk, v + reg1_inv_slot(2), idx, Nk);
                if (K == 2) reg1_load<4, reg1_unit(4, g.d[2]), GUARD>(a.app + row * Nk, v + reg1_inv_slot(3), idx, Nk);
                if (K == 3) reg1_load<2, reg1_unit(2, g.d[3]), GUARD>(a.app + row * Nk, v + reg1_inv_slot(4), idx, Nk);
            }
        }
    }
}

// level LV: the lane's N = 16 >> LV approximations `A` and details in.mine()[reg1_inv_slot(LV) ..] -> its 2 N values of
// A_{LV-1} in `out`
template <int HLEN, int K, int LV, int N>
PDWT_DEVICE void inv1d_reg_level(const FilterBankI& fb, WaveReg<real_t, N>& A, WaveReg<real_t, 16>& in, WaveReg<real_t, 2 * N>& out) {
    constexpr Reg1Geom g = reg1_inv_geom(HLEN, K, reg1_inv_o0(HLEN, K));
    constexpr int H2 = HLEN / 2, S = (H2 & 1) ? 0 : 1, E = g.E[LV], OP = g.d[LV - 1], NE = 2 * (N + E);
    WaveReg<real_t, NE> ext;  // (A, D) pairs: own N, then E of the following lanes
    PDWT_WAVE_LANES(lane) {
        real_t* v = ext.mine(lane);
        const real_t* av = A.mine(lane);
        const real_t* dv = in.mine(lane) + reg1_inv_slot(LV);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v[2 * i] = av[i];
            v[2 * i + 1] = dv[i];
        }
    }
    reg1_extend<2 * N, 2 * E, NE>(ext);
    PDWT_WAVE_LANES(lane) {
        const real_t* v = ext.mine(lane);
        real_t* o = out.mine(lane);
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) {
            constexpr int kBase = reg1_floor_half(OP + S);
            const int w = reg1_floor_half(OP + i + S) - kBase;   // first coefficient of output i (static after unrolling)
            const int par = 1 - ((OP + i + S) & 1);
            v2f acc = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < H2; ++j) acc = fma2(mk2(v[2 * (w + j)], v[2 * (w + j) + 1]), fb.t[HLEN - 1 - (2 * j + par)], acc);
            o[i] = acc.x + acc.y;
        }
    }
}

template <int HLEN, int K, bool GUARD>
PDWT_DEVICE void inv1d_reg_block(const Inv1DRegArgs& a, WaveReg<real_t, 16>& in, long long base, long long row, real_t* lds) {
    constexpr int O0 = reg1_inv_o0(HLEN, K);
    constexpr Reg1Geom g = reg1_inv_geom(HLEN, K, O0);
    WaveReg<real_t, 16> x;
    if constexpr (K == 3) {
        WaveReg<real_t, 2> a3;
        PDWT_WAVE_LANES(lane) { a3.mine(lane)[0] = in.mine(lane)[reg1_inv_slot(4)]; a3.mine(lane)[1] = in.mine(lane)[reg1_inv_slot(4) + 1]; }
        WaveReg<real_t, 4> a2;
        inv1d_reg_level<HLEN, K, 3, 2>(a.fb, a3, in, a2);
        WaveReg<real_t, 8> a1;
        inv1d_reg_level<HLEN, K, 2, 4>(a.fb, a2, in, a1);
        inv1d_reg_level<HLEN, K, 1, 8>(a.fb, a1, in, x);
    } else if constexpr (K == 2) {
        WaveReg<real_t, 4> a2;
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a2.mine(lane)[i] = in.mine(lane)[reg1_inv_slot(3) + i];
        }
        WaveReg<real_t, 8> a1;
        inv1d_reg_level<HLEN, K, 2, 4>(a.fb, a2, in, a1);
        inv1d_reg_level<HLEN, K, 1, 8>(a.fb, a1, in, x);
    } else {
        WaveReg<real_t, 8> a1;
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a1.mine(lane)[i] = in.mine(lane)[reg1_inv_slot(2) + i];
        }
        inv1d_reg_level<HLEN, K, 1, 8>(a.fb, a1, in, x);
    }
    // The lane's 16 samples are 64 contiguous bytes: stored directly, one instruction would scatter 64 pieces of
    // 16 B over 4 KiB (four times the write requests of a coalesced store: measured 30 us instead of 24 for levels
    // 1-3 of 2^24 samples).  They go through a wavefront-private LDS tile instead (lane stride 20 floats: the 16-B
    // accesses of both passes are conflict-free) and leave as four 1-KiB-contiguous stores.  No barrier: the tile
    // belongs to this wavefront and LDS executes a wavefront's accesses in order.
    PDWT_WAVE_LANES(lane) {
        const real_t* v = x.mine(lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            real4_t w;
            w.x = v[4 * i]; w.y = v[4 * i + 1]; w.z = v[4 * i + 2]; w.w = v[4 * i + 3];
            *reinterpret_cast<real4_t*>(lds + kReg1LdsStride * lane + 4 * i) = w;
        }
    }
    const RowBuf bo = row_buf(a.out + row * a.N0, kRealBytes * (unsigned)a.N0);
    PDWT_WAVE_LANES(lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 64 * i + lane;  // quad q of the block = samples 4q .. 4q+3, computed by lane q / 4
            const real4_t w = *reinterpret_cast<const real4_t*>(lds + kReg1LdsStride * (q >> 2) + 4 * (q & 3));
            long long sidx = base + O0 + 4 * q;
            if (GUARD) sidx %= a.N0;  // N0 % 16 == 0: a quad never straddles the end of the row
            const bool ok = (q >> 2) < g.V;
            row_st16(bo, ok ? kRealBytes * (unsigned)sidx : kReg1Dropped, w.x, w.y, w.z, w.w);
        }
    }
}

template <int HLEN, int K>
PDWT_DEVICE void dwt1_inv_reg(const Inv1DRegArgs& a, long long w, real_t* lds) {
    constexpr Reg1Geom g = reg1_inv_geom(HLEN, K, reg1_inv_o0(HLEN, K));
    constexpr int STEP = 16 * g.V;
    const long long row = w / a.wpr;
    if (row >= a.rows) return;
    const int first = (int)(w - row * a.wpr) * a.bpw;
    const int last = first + a.bpw < a.nblk ? first + a.bpw : a.nblk;  // exclusive
    const int plain_end = last < a.nplain ? last : a.nplain;
    WaveReg<real_t, 16> in, nx;
    if (first < plain_end) {
        inv1d_reg_load<HLEN, K, false>(a, nx, (long long)first * STEP, row);
        // Without this the loop header inherits "first block's loads pending" from here and "four stores pending" from
        // the back edge, and hipcc's merged s_waitcnt at the top of every trip then waits for the PREVIOUS block's
        // stores to complete (vmcnt(10) of 12 outstanding): one store round trip per block.
        PDWT_WAIT_VMEM();
        for (int b = first; b < plain_end; ++b) {
            PDWT_WAVE_LANES(lane) {
                real_t* v = in.mine(lane);
                const real_t* n = nx.mine(lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = n[i];
            }
            const int nb = b + 1 < plain_end ? b + 1 : b;
            inv1d_reg_load<HLEN, K, false>(a, nx, (long long)nb * STEP, row);
            PDWT_ROW_FENCE();  // keep the next block's loads HERE: hipcc otherwise sinks them below two levels of arithmetic
            inv1d_reg_block<HLEN, K, false>(a, in, (long long)b * STEP, row, lds);
        }
    }
    for (int b = first > plain_end ? first : plain_end; b < last; ++b) {
        inv1d_reg_load<HLEN, K, true>(a, in, (long long)b * STEP, row);
        inv1d_reg_block<HLEN, K, true>(a, in, (long long)b * STEP, row, lds);
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int K, int NT>
__global__ void __launch_bounds__(NT) dwt1_fwd_reg_kernel(const Fwd1DRegArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: rows, blocks and descriptors in SGPRs
    dwt1_fwd_reg<HLEN, K>(a, (long long)blockIdx.x * (NT / 64) + wave);
}
template <int HLEN, int K, int NT>
__global__ void __launch_bounds__(NT) dwt1_inv_reg_kernel(const Inv1DRegArgs a) {
    __shared__ __attribute__((aligned(32))) real_t tile[(NT / 64) * kReg1LdsFloats];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    dwt1_inv_reg<HLEN, K>(a, (long long)blockIdx.x * (NT / 64) + wave, tile + wave * kReg1LdsFloats);
}
#endif

}  // namespace pdwt
