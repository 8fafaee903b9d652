// swt2_fused_kernels.hpp -- up to THREE levels of a 2D stationary (a-trous) transform with a 2-tap filter bank
// (haar) in ONE launch, one wavefront per strip, everything in registers (gfx950).
//
// Why: the SWT writes four full-size planes per level, so a level-per-launch transform of L levels moves
// 5 L planes per direction (reference: pdwt/src/separable.cu:409-537, :553-672; here swt_kernels.hpp) although
// only 3 L + 2 are compulsory: every intermediate approximation is written and read back.  Unlike the decimated
// transform, the halo of fused SWT levels does NOT grow relative to the data -- K levels starting at dilation
// f0 reach (hlen-1) f0 (2^K - 1) samples, 7 for haar levels 1-3 -- so fusing costs a few lanes, not a factor.
// Levels l0 .. l0+K-1 (f0 = 2^(l0-1)) in one launch move 3K + 2 planes instead of 5K: 11 instead of 15 for
// levels 1-3, 8 instead of 10 for levels 4-5.
//
// Scheme (forward).  A wavefront owns a strip of 256 columns (a lane owns 4: one 16-B load per row) and walks
// DOWN the rows of ONE dilation phase of f0 (rows py, py + f0, py + 2 f0, ...: inside a phase the group's
// dilations are 1, 2, 4 rows).  Per input row:
//   * the row filter of level k needs the sample d_k = f0 2^k columns to the right: in the same lane or
//     (d_k + c) / 4 lanes ahead -- DPP wave_shl:1 moves;
//   * the filtered row (L, H: 8 values per lane) goes into a register ring of 2 / 4 / 8 rows (level 1 / 2 / 3 of
//     the group); the column filter combines it with the row 1 / 2 / 4 phase rows earlier and emits the level's
//     A, H, V, D row: H, V, D are stored (16 B per lane and plane), A is the next level's input row, in registers;
//   * an output row of level k lags the input by 2^k - 1 rows: a wavefront reads 2^K - 1 rows past its segment
//     and stores only the rows it owns: the store offset of a row it does not own (and of a lane without output)
//     lies beyond the plane's buffer descriptor and the hardware drops it -- no branch around a store, exact
//     s_waitcnt counts, 2^K - 1 rows of loads in flight.
// Strips overlap by ceil(sum_k d_k / 4) lanes (2 of 64 for levels 1-3): those lanes lack their right neighbours
// and store nothing.  The inverse walks the same way with the dependencies reversed (left neighbours, earlier
// rows: warm-up rows BEFORE the segment), all K levels on the same row per step (1 + 3 K loads per row, three rows
// ahead; 286 VGPRs, one wavefront per SIMD); the row synthesis' terms that come from the left are combined in the
// source lane and shifted afterwards (half the lane moves, bit-identical sums); a pending soft threshold is applied
// to the details as they are loaded (v_med3_f32), like swt2_inv_vec_tile does.
// Measured (one 2048^2 image, haar, levels 1-3 + 4-5, profiles/r02z_*, r02y_*): forward 34.5 + 26.0 us at 5.3 TB/s
// of algorithmic bytes; inverse 44.5 + 34 us -- it fetches 1.4x its algorithmic bytes, the 2^K - 1 warm-up rows of
// every 24-row segment from all planes.
//
// Arithmetic (restated in oracle/pdwt_oracle.c): analysis out[g] = x[g] f[1] + x[g + d] f[0]; synthesis
// out[g] = 0.5 (a[g - d] rlo[1] + b[g - d] rhi[1] + a[g] rlo[0] + b[g] rhi[0]), periodic in both directions.
#pragma once

#include "dwt2_wave_kernels.hpp"  // WaveReg, DPP shifts, RowBuf, wave_ld16
#include "dwt1_reg_kernels.hpp"   // row_st16, kReg1Dropped
#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

constexpr int kSwtFusedMaxLevels = 3;

// Planes of ANY size (round 6; the reference's kernels take any width and height: pdwt/src/separable.cu:409-493).  The
// kernels below were written for rows of whole 16-B groups and row counts the first dilation f0 divides; the GEN
// instantiations lift both conditions with the same walk:
//   * columns: a lane still owns four consecutive columns, loaded and stored as 16 B at 4-B alignment.  No lane may straddle
//     the row end (its last columns would come from the next row instead of the row start), so every strip but the first
//     starts `pad` = (4 - Nc mod 4) mod 4 columns further left: the lanes then hold columns x with x = Nc (mod 4) up to the
//     row end and 0, 4, 8 ... after it.  Strips 0 and 1 overlap by `pad` columns and store the same values there.
//   * rows: the rows a wavefront walks, r, r + f0, r + 2 f0 ... (mod Nr), form gcd(f0, Nr) closed CHAINS of Nr / gcd rows
//     instead of f0 phases of Nr / f0; inside a chain the group's dilations are 1, 2, 4 positions exactly as inside a
//     phase, so only the map from position to row changes: (c + f0 i) mod Nr, by a multiplication with ceil(2^32 / Nr).
struct SwtWalk {
    int phases;       // chains of rows: gcd(f0, Nr) (= f0 when f0 divides Nr)
    int rows_phase;   // rows of a chain: Nr / phases
    unsigned magic;   // ceil(2^32 / Nr)
    int pad;          // columns every strip but the first starts further left
};

static inline SwtWalk swt_walk(int Nr, int Nc, int f0, int cols_per_lane) {
    int g = f0, r = Nr % f0;
    while (r) { const int q = g % r; g = r; r = q; }
    SwtWalk w;
    w.phases = g;
    w.rows_phase = Nr / g;
    w.magic = (unsigned)(((1ULL << 32) + (unsigned)Nr - 1) / (unsigned)Nr);
    w.pad = (cols_per_lane - Nc % cols_per_lane) % cols_per_lane;
    return w;
}
static inline int swt_walk_strips(const SwtWalk& w, int Nc, int strip_cols) { return (Nc + w.pad + strip_cols - 1) / strip_cols; }
static inline bool swt_walk_general(int Nr, int Nc, int f0) { return (Nc % 4) != 0 || (Nr % f0) != 0; }

// row of position `idx` of chain `py`
template <bool GEN, int F0>
PDWT_DEVICE int swt_walk_row(const SwtWalk& wk, int Nr, int py, int idx) {
    const int n = py + F0 * idx;
    if constexpr (!GEN) {
        return n;
    } else {
        // n < f0 (Nr + 1) <= 2^20: the estimate of n / Nr is exact or one too large
        const int q = (int)(((unsigned long long)(unsigned)n * wk.magic) >> 32);
        const int r = n - q * Nr;
        return r < 0 ? r + Nr : r;
    }
}
PDWT_DEVICE int swt_strip_x0(const SwtWalk& wk, int strip, int strip_cols) { return strip * strip_cols - (strip > 0 ? wk.pad : 0); }

// 16-B / 8-B loads like wave_ld16 / wave_ld8 (dwt2_wave_kernels.hpp); GEN: at the alignment of one element
#ifdef PDWT_CPU_EMU
template <bool GEN> PDWT_DEVICE v4f swt_ld16(const real_t* base, unsigned byte_off) {
    const real_t* p = reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(base) + byte_off);
    v4f r; r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    return r;
}
template <bool GEN> PDWT_DEVICE v2f swt_ld8(const real_t* base, unsigned byte_off) {
    const real_t* p = reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(base) + byte_off);
    return mk2(p[0], p[1]);
}
#else
typedef real_t pdwt_v4u __attribute__((ext_vector_type(4), aligned(sizeof(real_t))));
typedef real_t pdwt_v2u __attribute__((ext_vector_type(2), aligned(sizeof(real_t))));
template <bool GEN> static __device__ __forceinline__ v4f swt_ld16(const real_t* base, unsigned byte_off) {
    if constexpr (GEN) {
        const pdwt_v4u v = *reinterpret_cast<const pdwt_v4u*>(reinterpret_cast<const char*>(base) + byte_off);
        v4f r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
        return r;
    } else {
        return wave_ld16(base, byte_off);
    }
}
template <bool GEN> static __device__ __forceinline__ v2f swt_ld8(const real_t* base, unsigned byte_off) {
    if constexpr (GEN) {
        const pdwt_v2u v = *reinterpret_cast<const pdwt_v2u*>(reinterpret_cast<const char*>(base) + byte_off);
        return mk2(v.x, v.y);
    } else {
        return wave_ld8(base, byte_off);
    }
}
#endif


struct SwtFusedArgs {
    const real_t* in;                    // forward: A_{l0-1}; inverse: A_{l0+K-1}
    real_t* out;                         // forward: A_{l0+K-1}; inverse: A_{l0-1}
    real_t* H[kSwtFusedMaxLevels];       // detail planes of the group's levels (forward: written; inverse: read)
    real_t* V[kSwtFusedMaxLevels];
    real_t* D[kSwtFusedMaxLevels];
    int Nr, Nc;
    long long bstride;                  // floats between the images of a batch
    int strips;                         // ceil((Nc + wk.pad) / (4 V))
    int segs;                           // segments per phase: ceil(wk.rows_phase / seg_rows)
    int seg_rows;                       // phase rows a wavefront owns (multiple of 2^K)
    real_t beta[kSwtFusedMaxLevels];     // inverse: soft threshold of each level's details (0 = none)
    real_t lo[2], hi[2];                 // analysis (forward) / synthesis (inverse) taps
    SwtWalk wk;                          // swt_walk(Nr, Nc, f0, columns per lane)
};

template <int K, int F0>
struct SwtFusedGeom {
    static_assert(K >= 2 && K <= kSwtFusedMaxLevels && (F0 == 1 || F0 % 4 == 0), "two or three levels; levels 1.. or whole-lane dilations");
    static constexpr int dist(int k) { return F0 << k; }                  // columns between the two taps of level k
    static constexpr int halo_cols = F0 * ((1 << K) - 1);
    static constexpr int halo_lanes = (halo_cols + 3) / 4;
    static constexpr int V = 64 - halo_lanes;                            // lanes that own output columns
    static constexpr int W = (1 << K) - 1;                               // extra phase rows a segment reads
    static constexpr int P = 1 << K;                                     // rows per unrolled group: ring and load-slot periods divide it
    static constexpr int NR = P;                                         // forward: input rows in flight (+ the current one)
    static_assert(halo_lanes < 32, "strip wide enough");
};

// sh[c] = the value D columns to the RIGHT of the lane's column c (forward) -- (c + D) / 4 lanes ahead
template <int D>
PDWT_DEVICE void swt_shift_right(WaveReg<real_t, 4>& src, WaveReg<real_t, 4>& sh) {
    constexpr int M = (3 + D) / 4;  // most lanes any column looks ahead
    WaveReg<real_t, 4 * (M + 1)> hop;  // hop[m] = the row as lane + m holds it
    PDWT_WAVE_LANES(lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hop.mine(lane)[i] = src.mine(lane)[i];
    }
#pragma unroll
    for (int m = 1; m <= M; ++m) {
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) hop.mine(lane)[4 * m + i] = hop.from_next(4 * (m - 1) + i, lane, 0.f);
        }
    }
    PDWT_WAVE_LANES(lane) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sh.mine(lane)[c] = hop.mine(lane)[4 * ((c + D) / 4) + (c + D) % 4];
    }
}
// Store targets are ONE range-checked buffer descriptor per plane and image, built once per wavefront (a descriptor
// per row and plane cost 4 SGPRs x 10 planes x 8 unrolled rows: hipcc spilled 250-450 SGPRs to VGPR lanes and the
// loop was 60 % v_readlane / v_writelane).  The byte offset of a store = (uniform) row offset + (per-lane) column
// offset; a row the wavefront does not own contributes kSwtRowDropped, a lane without output kSwtLaneDropped: either
// way the sum is >= 2^30 > the plane's bytes (the host checks Nr Nc <= 2^28) and the hardware drops the store.
constexpr unsigned kSwtRowDropped = 0x80000000u, kSwtLaneDropped = 0x40000000u;
PDWT_DEVICE RowBuf swt_plane(real_t* plane, long long image_off, int Nr, int Nc) {
    return row_buf(plane + image_off, kRealBytes * (unsigned)Nr * (unsigned)Nc);
}

// ---------------------------------------------------------------------------------------------- forward
template <int K, int F0>
struct SwtFwdState {
    using G = SwtFusedGeom<K, F0>;
    WaveReg<real_t, 4 * G::NR> ld;        // input rows in flight: slot (row mod NR)
    WaveReg<real_t, 8 * 2> ring1;         // (L, H) rows of level 1 of the group: [slot][L0..3 H0..3]
    WaveReg<real_t, 8 * 4> ring2;
    WaveReg<real_t, 8 * 8> ring3;
    WaveReg<unsigned, 2> off;            // byte offsets in a row: load (wrapped), store (or kSwtLaneDropped)
    RowBuf bH[3], bV[3], bD[3], bA;      // the output planes of this wavefront's image
};

// one level of one step: `a` = the level's input row (row index q of the walk), ring depth RD, lag LAG = RD / 2:
// emits the level's output row q - LAG: details through the descriptors, approximation into `anext`
template <int D, int RD, int SLOT, int NRING>
PDWT_DEVICE void swt_fwd_level(const SwtFusedArgs& a, WaveReg<real_t, 4>& ain, WaveReg<real_t, NRING>& ring, WaveReg<real_t, 4>& anext,
                               WaveReg<unsigned, 2>& off, const RowBuf& bH, const RowBuf& bV, const RowBuf& bD, unsigned rowoff) {
    constexpr int LAG = RD / 2, OLD = (SLOT - LAG + RD) % RD;
    WaveReg<real_t, 4> sh;
    swt_shift_right<D>(ain, sh);
    PDWT_WAVE_LANES(lane) {
        const real_t* x = ain.mine(lane);
        const real_t* s = sh.mine(lane);
        real_t* cur = ring.mine(lane) + 8 * SLOT;
        const real_t* old = ring.mine(lane) + 8 * OLD;
        // row filter: out = x f[1] + x(+d) f[0]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cur[c] = pdwt_fma(s[c], a.lo[0], x[c] * a.lo[1]);
            cur[4 + c] = pdwt_fma(s[c], a.hi[0], x[c] * a.hi[1]);
        }
        // column filter with the row LAG phase rows earlier (the earlier row is the output's own row)
        real_t* an = anext.mine(lane);
        real_t h[4], v[4], d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            an[c] = pdwt_fma(cur[c], a.lo[0], old[c] * a.lo[1]);
            h[c] = pdwt_fma(cur[c], a.hi[0], old[c] * a.hi[1]);
            v[c] = pdwt_fma(cur[4 + c], a.lo[0], old[4 + c] * a.lo[1]);
            d[c] = pdwt_fma(cur[4 + c], a.hi[0], old[4 + c] * a.hi[1]);
        }
        const unsigned o = off.mine(lane)[1] + rowoff;
        row_st16(bH, o, h[0], h[1], h[2], h[3]);
        row_st16(bV, o, v[0], v[1], v[2], v[3]);
        row_st16(bD, o, d[0], d[1], d[2], d[3]);
    }
}

// step R of a group of P rows (R static): input row r = g0 + R of the walk
template <int K, int F0, bool GEN, int R>
PDWT_DEVICE void swt_fwd_step(const SwtFusedArgs& a, SwtFwdState<K, F0>& st, const real_t* in, int g0, int i0, int rows_phase,
                              int py, long long boff) {
    using G = SwtFusedGeom<K, F0>;
    const int r = g0 + R;
    // request row r + NR - 1 (wrapped): its slot was consumed at the previous step
    {
        // never past the last row this wavefront filters (seg_rows + W - 1): the tail of the walk re-requests that
        // row (a cache hit) instead of rows nobody will use
        int rr = r + G::NR - 1;
        rr = rr < a.seg_rows + G::W ? rr : a.seg_rows + G::W - 1;
        rr = (i0 + rr) % rows_phase;
        const real_t* row = in + (long long)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, rr) * a.Nc;
        PDWT_WAVE_LANES(lane) {
            const v4f w = swt_ld16<GEN>(row, st.off.mine(lane)[0]);
            real_t* v = st.ld.mine(lane) + 4 * ((R + G::NR - 1) % G::NR);
            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
        }
    }
    PDWT_ROW_FENCE();
    WaveReg<real_t, 4> a0, a1, a2, a3;
    PDWT_WAVE_LANES(lane) {
        const real_t* v = st.ld.mine(lane) + 4 * (R % G::NR);
#pragma unroll
        for (int c = 0; c < 4; ++c) a0.mine(lane)[c] = v[c];
    }
    // output rows of this step (relative to the segment): level k emits row r - (2^(k+1) - 1); byte offset of that
    // row in a plane, or kSwtRowDropped for a row another wavefront owns
    auto rowoff = [&](int rel) -> unsigned {
        const bool ow = rel >= 0 && rel < a.seg_rows && i0 + rel < rows_phase;
        return ow ? kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, i0 + rel) * (unsigned)a.Nc : kSwtRowDropped;
    };
    {
        const unsigned ro = rowoff(r - 1);
        swt_fwd_level<G::dist(0), 2, R % 2, 16>(a, a0, st.ring1, a1, st.off, st.bH[0], st.bV[0], st.bD[0], ro);
    }
    {
        const unsigned ro = rowoff(r - 3);
        // a1 is row q = r - 1 of level 2's input: slot q mod 4 = (R + 3) mod 4
        swt_fwd_level<G::dist(1), 4, (R + 3) % 4, 32>(a, a1, st.ring2, a2, st.off, st.bH[1], st.bV[1], st.bD[1], ro);
        if constexpr (K == 2) {
            PDWT_WAVE_LANES(lane) { const real_t* v = a2.mine(lane); row_st16(st.bA, st.off.mine(lane)[1] + ro, v[0], v[1], v[2], v[3]); }
        }
    }
    if constexpr (K >= 3) {
        const unsigned ro = rowoff(r - 7);
        // a2 is row p = r - 3 of level 3's input: slot p mod 8 = (R + 5) mod 8
        swt_fwd_level<G::dist(2), 8, (R + 5) % 8, 64>(a, a2, st.ring3, a3, st.off, st.bH[2], st.bV[2], st.bD[2], ro);
        PDWT_WAVE_LANES(lane) { const real_t* v = a3.mine(lane); row_st16(st.bA, st.off.mine(lane)[1] + ro, v[0], v[1], v[2], v[3]); }
    }
}

template <int K, int F0, bool GEN, int R>
PDWT_DEVICE void swt_fwd_group(const SwtFusedArgs& a, SwtFwdState<K, F0>& st, const real_t* in, int g0, int i0, int rows_phase,
                               int py, long long boff) {
    if constexpr (R < SwtFusedGeom<K, F0>::P) {
        swt_fwd_step<K, F0, GEN, R>(a, st, in, g0, i0, rows_phase, py, boff);
        swt_fwd_group<K, F0, GEN, R + 1>(a, st, in, g0, i0, rows_phase, py, boff);
    }
}

// wavefront `w` of the launch: (image, phase, segment, strip)
template <int K, int F0, bool GEN = false>
PDWT_DEVICE void swt2_fwd_fused(const SwtFusedArgs& a, long long w) {
    using G = SwtFusedGeom<K, F0>;
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
    SwtFwdState<K, F0> st;
    PDWT_WAVE_LANES(lane) {
        const int x = (GEN ? swt_strip_x0(a.wk, strip, 4 * G::V) : strip * 4 * G::V) + 4 * lane;
        // columns past the row end wrap (no lane straddles it: Nc % 4 == 0 or SwtWalk::pad; strips * 4 V < 2 Nc) as far as a valid lane's window reaches;
        // further right nothing is used: those lanes re-read the row's last group (a cache hit, no extra traffic)
        int xl = x >= a.Nc ? x - a.Nc : x;
        if (x >= a.Nc + G::halo_cols + 3) xl = a.Nc - 4;
        st.off.mine(lane)[0] = kRealBytes * (unsigned)xl;
        st.off.mine(lane)[1] = (lane < G::V && x < a.Nc) ? kRealBytes * (unsigned)x : kSwtLaneDropped;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        st.bH[k] = swt_plane(a.H[k], boff, a.Nr, a.Nc);
        st.bV[k] = swt_plane(a.V[k], boff, a.Nr, a.Nc);
        st.bD[k] = swt_plane(a.D[k], boff, a.Nr, a.Nc);
    }
    st.bA = swt_plane(a.out, boff, a.Nr, a.Nc);
    // rows 0 .. NR-2 of the walk in flight before the first step
#pragma unroll
    for (int p = 0; p < G::NR - 1; ++p) {
        const real_t* row = in + (long long)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, (i0 + p) % rows_phase) * a.Nc;
        PDWT_WAVE_LANES(lane) {
            const v4f v4 = swt_ld16<GEN>(row, st.off.mine(lane)[0]);
            real_t* v = st.ld.mine(lane) + 4 * p;
            v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        }
    }
    PDWT_WAIT_VMEM();  // the loop header must not inherit pending loads (see dwt1_inv_reg)
    // seg_rows + W input rows, in groups of P (seg_rows is a multiple of P: one extra group covers the W <= P - 1 rows)
    const int ngroups = a.seg_rows / G::P + 1;
    for (int g = 0; g < ngroups; ++g) swt_fwd_group<K, F0, GEN, 0>(a, st, in, g * G::P, i0, rows_phase, py, boff);
}

// ---------------------------------------------------------------------------------------------- inverse
// C = columns per lane: 4 (16-B accesses, strips of 256 columns; what the host launches) or 2 (8-B accesses, strips
// of 128 columns: twice the wavefronts for the same segment length and half the registers -- built to afford longer
// segments for one image, measured 3-8 % slower at every segment length, kept selectable for re-measurement).
template <int K, int F0, int C>
struct SwtInvGeom {
    static_assert(C == 2 || C == 4, "8-B or 16-B lanes");
    using G = SwtFusedGeom<K, F0>;
    static constexpr int halo_lanes = (G::halo_cols + C - 1) / C;
    static constexpr int V = 64 - halo_lanes;
    static_assert(halo_lanes < 32, "strip wide enough");
};

// sh[c] = the value D columns to the LEFT of the lane's column c
template <int D, int C>
PDWT_DEVICE void swt_shift_left(WaveReg<real_t, C>& src, WaveReg<real_t, C>& sh) {
    constexpr int M = (D + C - 1) / C;
    WaveReg<real_t, C * (M + 1)> hop;  // hop[m] = the row as lane - m holds it
    PDWT_WAVE_LANES(lane) {
#pragma unroll
        for (int i = 0; i < C; ++i) hop.mine(lane)[i] = src.mine(lane)[i];
    }
#pragma unroll
    for (int m = 1; m <= M; ++m) {
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int i = 0; i < C; ++i) hop.mine(lane)[C * m + i] = hop.from_prev(C * (m - 1) + i, lane, 0.f);
        }
    }
    PDWT_WAVE_LANES(lane) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            constexpr int kBig = 1 << 20;                      // keeps the C++ division / modulo non-negative
            const int q = c - D + C * kBig;                    // column c - D, as lanes back (kBig - q / C) and index q % C
            sh.mine(lane)[c] = hop.mine(lane)[C * (kBig - q / C) + q % C];
        }
    }
}

template <int K, int F0, int NRI_, int C>
struct SwtInvState {
    using G = SwtFusedGeom<K, F0>;
    // load slots: the current row + NRI - 1 in flight, 1 + 3 K planes each.  8 slots (C = 4: 476 VGPRs, 256 SGPRs
    // spilled) measured 15 % slower than 4, 2 slots the same as 4 (profiles/r02w_bench_cfg4_fused_sweep.txt).
    static constexpr int NRI = NRI_;
    static_assert(G::P % NRI_ == 0, "static slot numbers");
    WaveReg<real_t, C * NRI*(1 + 3 * K)> ld;             // [slot][plane][C]: plane 0 = A, then H, V, D of the group's levels, deepest first
    WaveReg<real_t, 2 * C * 2> ring1;                    // (u1, u2) rows of level 1 of the group
    WaveReg<real_t, 2 * C * 4> ring2;
    WaveReg<real_t, 2 * C * 8> ring3;
    WaveReg<unsigned, 2> off;
    RowBuf bo;                                          // the output plane of this wavefront's image
};

// all planes of input row `ro` (byte offset of the row in a plane of this image) into load slot SLOT.
// `wrow` = the row's position in the walk.  The W = 2^K - 1 warm-up rows in front of a segment only feed the recursion:
// level m of the group (m = 1 finest, dilation 2^(m-1) rows) needs its details from walk row W - (2^m - 1) on, so of the
// 7 x 10 plane-rows in front of a three-level segment only 40 are used.  A warm-up row's unused detail loads are
// redirected (a uniform pointer select: no branch around a load, the vmcnt bookkeeping stays exact) to the approximation
// plane's row, which the same step loads anyway: they hit in cache instead of fetching 30 rows per segment nobody reads.
// The values they deliver are finite and only reach ring slots that are overwritten before the first owned row.
template <int K, int F0, int NRI, int C, bool GEN, int SLOT>
PDWT_DEVICE void swt_inv_load(const SwtFusedArgs& a, SwtInvState<K, F0, NRI, C>& st, long long boff, unsigned ro, int wrow) {
    constexpr int NP = 1 + 3 * K;
    using G = SwtFusedGeom<K, F0>;
    PDWT_WAVE_LANES(lane) {
        const unsigned o = st.off.mine(lane)[0] + ro;
        real_t* base = st.ld.mine(lane) + C * NP * SLOT;
        auto put = [&](int p, const real_t* plane) {
            real_t* v = base + C * p;
            if constexpr (C == 4) {
                const v4f w = swt_ld16<GEN>(plane + boff, o);
                v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
            } else {
                const v2f w = swt_ld8<GEN>(plane + boff, o);
                v[0] = w.x; v[1] = w.y;
            }
        };
        put(0, a.in);
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {  // deepest level first
            const int p = 1 + 3 * (K - 1 - k);
            const bool used = wrow >= G::W - ((2 << k) - 1);  // uniform; always true once the warm-up rows are behind
            put(p, used ? a.H[k] : a.in);
            put(p + 1, used ? a.V[k] : a.in);
            put(p + 2, used ? a.D[k] : a.in);
        }
    }
}

// one synthesis level on the current row: ain (approximation row) + the level's details (thresholded) -> aout;
// the (u1, u2) row goes into ring slot SLOT, the row LAG = RD / 2 phase rows earlier is its column partner
template <int D, int RD, int SLOT, int C, int NRING>
PDWT_DEVICE void swt_inv_level(const SwtFusedArgs& a, WaveReg<real_t, C>& ain, WaveReg<real_t, 3 * C>& det, real_t beta,
                               WaveReg<real_t, NRING>& ring, WaveReg<real_t, C>& aout) {
    constexpr int LAG = RD / 2, OLD = (SLOT - LAG + RD) % RD;
    // the two terms of the row synthesis that come from D columns to the left, combined where they live
    WaveReg<real_t, C> t1, t2, s1, s2;
    PDWT_WAVE_LANES(lane) {
        const real_t* x = ain.mine(lane);
        real_t* dd = det.mine(lane);   // [H | V | D]
#pragma unroll
        for (int i = 0; i < 3 * C; ++i) dd[i] = soft_shrink(dd[i], beta);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            t1.mine(lane)[c] = pdwt_fma(dd[C + c], a.hi[1], x[c] * a.lo[1]);          // A rlo[1] + V rhi[1]
            t2.mine(lane)[c] = pdwt_fma(dd[2 * C + c], a.hi[1], dd[c] * a.lo[1]);     // H rlo[1] + D rhi[1]
        }
    }
    swt_shift_left<D, C>(t1, s1);
    swt_shift_left<D, C>(t2, s2);
    PDWT_WAVE_LANES(lane) {
        const real_t* x = ain.mine(lane);
        const real_t* dd = det.mine(lane);
        real_t* cur = ring.mine(lane) + 2 * C * SLOT;
        const real_t* old = ring.mine(lane) + 2 * C * OLD;
        real_t* o = aout.mine(lane);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            real_t u1 = pdwt_fma(x[c], a.lo[0], s1.mine(lane)[c]);
            u1 = pdwt_fma(dd[C + c], a.hi[0], u1);
            real_t u2 = pdwt_fma(dd[c], a.lo[0], s2.mine(lane)[c]);
            u2 = pdwt_fma(dd[2 * C + c], a.hi[0], u2);
            cur[c] = 0.5f * u1;
            cur[C + c] = 0.5f * u2;
            real_t r = old[c] * a.lo[1];
            r = pdwt_fma(old[C + c], a.hi[1], r);
            r = pdwt_fma(cur[c], a.lo[0], r);
            r = pdwt_fma(cur[C + c], a.hi[0], r);
            o[c] = 0.5f * r;
        }
    }
}

template <int K, int F0, int NRI, int C, bool GEN, int R>
PDWT_DEVICE void swt_inv_step(const SwtFusedArgs& a, SwtInvState<K, F0, NRI, C>& st, int g0, int i0, int rows_phase, int py, long long boff) {
    using G = SwtFusedGeom<K, F0>;
    using S = SwtInvState<K, F0, NRI, C>;
    constexpr int NP = 1 + 3 * K;
    const int r = g0 + R;  // row r of the walk = phase row i0 - W + r
    {
        int rr = r + S::NRI - 1;
        rr = rr < a.seg_rows + G::W ? rr : a.seg_rows + G::W - 1;  // see swt_fwd_step
        const int wrow = rr;
        rr = i0 - G::W + rr;
        rr = ((rr % rows_phase) + rows_phase) % rows_phase;
        swt_inv_load<K, F0, NRI, C, GEN, (R + S::NRI - 1) % S::NRI>(a, st, boff, kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, rr) * (unsigned)a.Nc, wrow);
    }
    PDWT_ROW_FENCE();
    WaveReg<real_t, C> cur, nxt;
    WaveReg<real_t, 3 * C> det;
    auto take = [&](int plane0) {
        PDWT_WAVE_LANES(lane) {
            const real_t* v = st.ld.mine(lane) + C * NP * (R % S::NRI) + C * plane0;
#pragma unroll
            for (int i = 0; i < 3 * C; ++i) det.mine(lane)[i] = v[i];
        }
    };
    PDWT_WAVE_LANES(lane) {
        const real_t* v = st.ld.mine(lane) + C * NP * (R % S::NRI);
#pragma unroll
        for (int c = 0; c < C; ++c) cur.mine(lane)[c] = v[c];
    }
    // deepest level of the group first; ring slots: all levels work on row r
    if constexpr (K >= 3) {
        take(1);
        swt_inv_level<G::dist(2), 8, R % 8, C>(a, cur, det, a.beta[2], st.ring3, nxt);
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int c = 0; c < C; ++c) cur.mine(lane)[c] = nxt.mine(lane)[c];
        }
    }
    if constexpr (K >= 2) {
        take(1 + 3 * (K - 2));
        swt_inv_level<G::dist(1), 4, R % 4, C>(a, cur, det, a.beta[1], st.ring2, nxt);
        PDWT_WAVE_LANES(lane) {
#pragma unroll
            for (int c = 0; c < C; ++c) cur.mine(lane)[c] = nxt.mine(lane)[c];
        }
    }
    take(1 + 3 * (K - 1));
    swt_inv_level<G::dist(0), 2, R % 2, C>(a, cur, det, a.beta[0], st.ring1, nxt);
    const int rel = r - G::W;  // output row relative to the segment
    const bool ow = rel >= 0 && rel < a.seg_rows && i0 + rel < rows_phase;
    const unsigned ro = ow ? kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, i0 + rel) * (unsigned)a.Nc : kSwtRowDropped;
    PDWT_WAVE_LANES(lane) {
        const real_t* v = nxt.mine(lane);
        if constexpr (C == 4) row_st16(st.bo, st.off.mine(lane)[1] + ro, v[0], v[1], v[2], v[3]);
        else row_st8(st.bo, st.off.mine(lane)[1] + ro, v[0], v[1]);
    }
}

template <int K, int F0, int NRI, int C, bool GEN, int R>
PDWT_DEVICE void swt_inv_group(const SwtFusedArgs& a, SwtInvState<K, F0, NRI, C>& st, int g0, int i0, int rows_phase, int py, long long boff) {
    if constexpr (R < SwtFusedGeom<K, F0>::P) {
        swt_inv_step<K, F0, NRI, C, GEN, R>(a, st, g0, i0, rows_phase, py, boff);
        swt_inv_group<K, F0, NRI, C, GEN, R + 1>(a, st, g0, i0, rows_phase, py, boff);
    }
}

// rows 0 .. NRI-2 of the walk into slots 0 .. NRI-2
template <int K, int F0, int NRI, int C, bool GEN, int I, class RowBytes>
PDWT_DEVICE void swt_inv_preload(const SwtFusedArgs& a, SwtInvState<K, F0, NRI, C>& st, long long boff, const RowBytes& rowbytes) {
    if constexpr (I < NRI - 1) {
        swt_inv_load<K, F0, NRI, C, GEN, I>(a, st, boff, rowbytes(I), I);
        swt_inv_preload<K, F0, NRI, C, GEN, I + 1>(a, st, boff, rowbytes);
    }
}

// a.strips = ceil((Nc + wk.pad) / (C V)) with V of SwtInvGeom<K, F0, C>
template <int K, int F0, int NRI, int C, bool GEN = false>
PDWT_DEVICE void swt2_inv_fused(const SwtFusedArgs& a, long long w) {
    using G = SwtFusedGeom<K, F0>;
    using GI = SwtInvGeom<K, F0, C>;
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
    SwtInvState<K, F0, NRI, C> st;
    PDWT_WAVE_LANES(lane) {
        // the first halo_lanes lanes lack their left neighbours: lane halo_lanes owns column strip * C V
        const int x = (GEN ? swt_strip_x0(a.wk, strip, C * GI::V) : strip * C * GI::V) + C * (lane - GI::halo_lanes);
        // left of the row start: wrapped (the halo of the first columns); at or past the row end: nothing is used
        const int xl = x < 0 ? x + a.Nc : (x >= a.Nc ? a.Nc - C : x);
        st.off.mine(lane)[0] = kRealBytes * (unsigned)xl;
        st.off.mine(lane)[1] = (lane >= GI::halo_lanes && x < a.Nc) ? kRealBytes * (unsigned)x : kSwtLaneDropped;
    }
    st.bo = swt_plane(a.out, boff, a.Nr, a.Nc);
    // rows 0 .. NRI-2 of the walk (phase rows i0 - W ...) in flight before the first step
    {
        auto rowbytes = [&](int i) {
            return kRealBytes * (unsigned)swt_walk_row<GEN, F0>(a.wk, a.Nr, py, ((i0 - G::W + i) % rows_phase + rows_phase) % rows_phase) * (unsigned)a.Nc;
        };
        swt_inv_preload<K, F0, NRI, C, GEN, 0>(a, st, boff, rowbytes);
    }
    PDWT_WAIT_VMEM();
    // W warm-up rows + seg_rows rows, in groups of P
    const int ngroups = a.seg_rows / G::P + 1;
    for (int g = 0; g < ngroups; ++g) swt_inv_group<K, F0, NRI, C, GEN, 0>(a, st, g * G::P, i0, rows_phase, py, boff);
}

#ifndef PDWT_CPU_EMU
// Workgroup -> wavefront number.  Consecutive workgroup ids sit on DIFFERENT XCDs (ids b and b + 8 share one), and
// consecutive wavefront numbers are neighbouring column strips of one row segment, which read the same halo lines at the
// same time: XCD x takes the contiguous eighth [x chunk, (x + 1) chunk) of the wavefronts so that the neighbours meet in
// one L2 (the launcher rounds the grid up to 8 chunk workgroups; surplus ones return).
PDWT_DEVICE long long swt_fused_wave(unsigned block, int waves_per_block, int wave, long long waves) {
    const long long blocks = (waves + waves_per_block - 1) / waves_per_block;
    const long long chunk = (blocks + 7) >> 3;
    const long long b = (long long)(block & 7) * chunk + (block >> 3);
    return (block >> 3) < chunk && b < blocks ? b * waves_per_block + wave : waves;
}

template <int K, int F0, int NT, bool GEN>
__global__ void __launch_bounds__(NT, 1) swt2_fwd_fused_kernel(const SwtFusedArgs a, long long waves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = swt_fused_wave(blockIdx.x, NT / 64, wave, waves);
    if (w < waves) swt2_fwd_fused<K, F0, GEN>(a, w);
}
template <int K, int F0, int NRI, int C, int NT, bool GEN>
__global__ void __launch_bounds__(NT, 1) swt2_inv_fused_kernel(const SwtFusedArgs a, long long waves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = swt_fused_wave(blockIdx.x, NT / 64, wave, waves);
    if (w < waves) swt2_inv_fused<K, F0, NRI, C, GEN>(a, w);
}
#endif

}  // namespace pdwt
