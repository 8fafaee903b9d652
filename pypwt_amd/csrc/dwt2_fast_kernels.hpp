// dwt2_fast_kernels.hpp -- tuned fused 2D DWT level kernels for gfx950 (the hot path of
// BASELINE config 2/5: 4096x4096 db4).
//
// Same tile decomposition and arithmetic as dwt2_kernels.hpp (which stays as the generic path for
// odd filter lengths, row lengths that are not a multiple of 4, and unaligned buffers), rebuilt
// around what rocprofv3 showed on the first version (profiles/r01a_*): the level-1 kernel was
// VALU-issue bound (~75 vector instructions per input sample, most of them per-element index
// arithmetic in the staging loop), not HBM bound.  Changes:
//   * staging uses aligned 16-B global loads and ds_write_b128; periodization is resolved per
//     float4 (a row of 4k samples never straddles the wrap) and per staged row, not per sample;
//   * the low/high intermediates are kept INTERLEAVED in LDS as (L,H) pairs and every
//     multiply-add is a packed v_pk_fma_f32 on such a pair: taps come from the kernel-argument
//     segment as (lo,hi) SGPR pairs, broadcast through op_sel, two FMAs per instruction;
//   * each thread produces two adjacent outputs per LDS read group (ds_read_b128 / _b64);
//   * workgroups are renumbered so that the 8 XCDs each work on a contiguous band of tiles and
//     halo rows are re-read from that XCD's own L2.
#pragma once

#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

struct Fwd2DFastArgs {
    const real_t* in;
    real_t *A, *H, *V, *D;
    int Nr, Nc, Nr2, Nc2;
    long long in_bstride, out_bstride;
    int tiles_x, tiles_y;  // tile grid of one image
    FilterBankI fb;        // (dec_lo, dec_hi)
};

struct Inv2DFastArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int tiles_x, tiles_y;
    FilterBankI fb;  // (rec_lo, rec_hi)
};

// 16-B load at 4-B alignment (rows of images whose width is not a multiple of 4): one global_load_dwordx4 on gfx950
#ifdef PDWT_CPU_EMU
static inline real4_t ld_f4_unaligned(const real_t* p) { return real4_t{p[0], p[1], p[2], p[3]}; }
#else
typedef real_t pdwt_f4u __attribute__((ext_vector_type(4), aligned(sizeof(real_t))));
static __device__ __forceinline__ real4_t ld_f4_unaligned(const real_t* p) {
    const pdwt_f4u v = *reinterpret_cast<const pdwt_f4u*>(p);
    real4_t r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
#endif

// 16-B / 8-B stores at 4-B alignment
#ifdef PDWT_CPU_EMU
static inline void st_f4_unaligned(real_t* p, real_t a, real_t b, real_t c, real_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
static inline void st_f2_unaligned(real_t* p, real_t a, real_t b) { p[0] = a; p[1] = b; }
#else
typedef real_t pdwt_f2u __attribute__((ext_vector_type(2), aligned(sizeof(real_t))));
static __device__ __forceinline__ void st_f4_unaligned(real_t* p, real_t a, real_t b, real_t c, real_t d) {
    pdwt_f4u v;
    v.x = a; v.y = b; v.z = c; v.w = d;
    *reinterpret_cast<pdwt_f4u*>(p) = v;
}
static __device__ __forceinline__ void st_f2_unaligned(real_t* p, real_t a, real_t b) {
    pdwt_f2u v;
    v.x = a; v.y = b;
    *reinterpret_cast<pdwt_f2u*>(p) = v;
}
#endif

// Block renumbering: hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
// ids b and b+8 share an L2.  Give XCD x the contiguous tile range [x*chunk, (x+1)*chunk) (row-major),
// i.e. a horizontal band of the image, so vertically adjacent tiles (which share halo rows) meet
// in the same L2.  Placement only affects speed, never results.  Returns false for padding ids.
PDWT_DEVICE bool xcd_tile(int block, int tiles_x, int tiles_y, int& bx, int& by) {
    const int total = tiles_x * tiles_y;
    const int chunk = (total + 7) >> 3;
    const int tile = (block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || tile >= total) return false;
    by = tile / tiles_x;
    bx = tile - by * tiles_x;
    return true;
}

template <int HLEN, int TX>
struct FwdFastGeom {
    static constexpr int C = HLEN / 2 - 1;           // analysis centre
    static constexpr int PADL = (4 - (C & 3)) & 3;   // samples between the aligned load origin and x0
    static constexpr int RXA = (PADL + 2 * TX + HLEN - 2 + 3) & ~3;  // staged row length (multiple of 4)
    static constexpr int NV = (PADL + HLEN + 2 + 3) & ~3;            // LDS floats read per thread per row
};

template <int HLEN, int TX, int TY>
constexpr int fwd2d_fast_lds_floats() {
    return (2 * TY + HLEN - 2) * (FwdFastGeom<HLEN, TX>::RXA + 2 * TX);
}

// row analysis of the staged tile: two adjacent outputs per thread, packed (L,H) accumulators
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void fwd_fast_row_pass(int tid, const real_t* sIn, v2f* tLH, const FilterBankI& fb) {
    using G = FwdFastGeom<HLEN, TX>;
    constexpr int PADL = G::PADL, RXA = G::RXA, NV = G::NV;
    constexpr int RY = 2 * TY + HLEN - 2;
    constexpr int HT = TX / 2;
    const int t = tid % HT;
    for (int r = tid / HT; r < RY; r += NT / HT) {
        real_t v[NV];
        const real_t* p4 = sIn + r * RXA + 4 * t;
        v4f w[NV / 4];
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) w[q] = lds_load16(p4 + 4 * q);
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            lds_pin(w[q]);
            v[4 * q + 0] = w[q].x;
            v[4 * q + 1] = w[q].y;
            v[4 * q + 2] = w[q].z;
            v[4 * q + 3] = w[q].w;
        }
        v2f acc0 = mk2(real_t(0), real_t(0)), acc1 = mk2(real_t(0), real_t(0));
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const v2f tap = fb.t[HLEN - 1 - j];
            acc0 = fma2(bc(v[PADL + j]), tap, acc0);
            acc1 = fma2(bc(v[PADL + 2 + j]), tap, acc1);
        }
        real4_t o;
        o.x = acc0.x;
        o.y = acc0.y;
        o.z = acc1.x;
        o.w = acc1.y;
        *reinterpret_cast<real4_t*>(&tLH[r * TX + 2 * t]) = o;
    }
}

// column analysis; (L,H) pair x tap -> (A,V) with lo, (H,D) with hi.  Each thread owns two adjacent
// columns (one ds_read_b128 per staged row, 8-B stores).
// coh_out (uniform): the A band has a consumer in this launch (dwt2_chain_kernels.hpp) -- written with 16-B sc1 stores,
// an even lane taking its odd neighbour's column pair (8-B sc1 stores cost 2.7x per byte, MI355X_MICROARCH.md)
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void fwd_fast_col_pass(int tid, const v2f* tLH, const Fwd2DFastArgs& a, int bx, int by, int bz,
                                   bool coh_out = false) {
    constexpr int HT = TX / 2;
    constexpr int R = TY / (NT / HT);
    const int t = tid % HT;
    const int ty0 = (tid / HT) * R;
    const int ox = bx * TX + 2 * t;
    v2f accAV[R][2], accHD[R][2];
#pragma unroll
    for (int i = 0; i < R; ++i) accAV[i][0] = accAV[i][1] = accHD[i][0] = accHD[i][1] = mk2(real_t(0), real_t(0));
    constexpr int NRW = 2 * R + HLEN - 2, GB = 12;  // LDS loads are issued GB at a time, then consumed
#pragma unroll
    for (int r0 = 0; r0 < NRW; r0 += GB) {
        v4f w[GB];
#pragma unroll
        for (int g = 0; g < GB; ++g)
            if (r0 + g < NRW) w[g] = lds_load16(&tLH[(2 * ty0 + r0 + g) * TX + 2 * t]);
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int r = r0 + g;
            if (r < NRW) {
                lds_pin(w[g]);
                const v2f lh0 = mk2(w[g].x, w[g].y), lh1 = mk2(w[g].z, w[g].w);
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int j = r - 2 * i;
                    if (j >= 0 && j < HLEN) {
                        const v2f tap = a.fb.t[HLEN - 1 - j];
                        accAV[i][0] = fma2(lh0, bc(tap.x), accAV[i][0]);
                        accHD[i][0] = fma2(lh0, bc(tap.y), accHD[i][0]);
                        accAV[i][1] = fma2(lh1, bc(tap.x), accAV[i][1]);
                        accHD[i][1] = fma2(lh1, bc(tap.y), accHD[i][1]);
                    }
                }
            }
        }
    }
    const long long boff = (long long)bz * a.out_bstride;
#if !defined(PDWT_CPU_EMU)
    if (coh_out) {  // the host guarantees whole tiles and Nc2 % 4 == 0 on this path
        const CohPlane pa = coh_plane(a.A + boff);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int oy = by * TY + ty0 + i;
            const long long o = (long long)oy * a.Nc2 + ox;
            real4_t q;
            q.x = accAV[i][0].x; q.y = accAV[i][1].x;
            q.z = __shfl_xor(accAV[i][0].x, 1); q.w = __shfl_xor(accAV[i][1].x, 1);
            if (!(tid & 1)) coh_store16(pa, o, q);
            real2_t v;
            v.x = accAV[i][0].y; v.y = accAV[i][1].y; *reinterpret_cast<real2_t*>(a.V + boff + o) = v;
            v.x = accHD[i][0].x; v.y = accHD[i][1].x; *reinterpret_cast<real2_t*>(a.H + boff + o) = v;
            v.x = accHD[i][0].y; v.y = accHD[i][1].y; *reinterpret_cast<real2_t*>(a.D + boff + o) = v;
        }
        return;
    }
#endif
    if ((a.Nc2 & 1) || (a.out_bstride & 1)) {  // odd coefficient rows: 8-B stores would straddle / run past the row end
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int oy = by * TY + ty0 + i;
            if (oy < a.Nr2 && ox + 1 < a.Nc2) {  // both columns inside: 8-B stores at 4-B alignment
                const long long o = boff + (long long)oy * a.Nc2 + ox;
                st_f2_unaligned(a.A + o, accAV[i][0].x, accAV[i][1].x);
                st_f2_unaligned(a.V + o, accAV[i][0].y, accAV[i][1].y);
                st_f2_unaligned(a.H + o, accHD[i][0].x, accHD[i][1].x);
                st_f2_unaligned(a.D + o, accHD[i][0].y, accHD[i][1].y);
            } else if (oy < a.Nr2 && ox < a.Nc2) {  // the last column of an odd row
                const long long o = boff + (long long)oy * a.Nc2 + ox;
                a.A[o] = accAV[i][0].x; a.V[o] = accAV[i][0].y;
                a.H[o] = accHD[i][0].x; a.D[o] = accHD[i][0].y;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int oy = by * TY + ty0 + i;
        if (oy < a.Nr2 && ox < a.Nc2) {  // Nc2 is even: ox + 1 is inside too
            const long long o = boff + (long long)oy * a.Nc2 + ox;
            real2_t v;
            v.x = accAV[i][0].x; v.y = accAV[i][1].x; *reinterpret_cast<real2_t*>(a.A + o) = v;
            v.x = accAV[i][0].y; v.y = accAV[i][1].y; *reinterpret_cast<real2_t*>(a.V + o) = v;
            v.x = accHD[i][0].x; v.y = accHD[i][1].x; *reinterpret_cast<real2_t*>(a.H + o) = v;
            v.x = accHD[i][0].y; v.y = accHD[i][1].y; *reinterpret_cast<real2_t*>(a.D + o) = v;
        }
    }
}

// Requirements (checked by the host): HLEN even, Nc % 4 == 0, 16-B aligned image rows.
// coh_in / coh_out (uniform, dwt2_chain_kernels.hpp): the input plane was written by other workgroups of THIS launch
// (sc1 loads; needs the branch-free staging conditions) / the A band is read by other workgroups of this launch.
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_fwd_fast_tile(const Fwd2DFastArgs& a, int bx, int by, int bz, real_t* smem, bool coh_in = false,
                                    bool coh_out = false) {
    using G = FwdFastGeom<HLEN, TX>;
    constexpr int C = G::C, PADL = G::PADL, RXA = G::RXA, NV = G::NV;
    constexpr int RY = 2 * TY + HLEN - 2;
    constexpr int V4 = RXA / 4;        // float4 per staged row
    constexpr int HT = TX / 2;         // threads per row in the row pass (2 outputs each)
    static_assert(NT % HT == 0, "thread layout");
    constexpr int NG = NT / HT;        // thread groups along y in the column pass
    static_assert(TY % NG == 0, "tile height must split over the thread groups");
    [[maybe_unused]] constexpr int R = TY / NG;         // output rows per thread (x 2 columns)
    static_assert(2 * TX - 4 + NV <= RXA, "row-pass reads stay inside the staged row");

    real_t* sIn = smem;                                   // RY x RXA floats
    v2f* tLH = reinterpret_cast<v2f*>(smem + RY * RXA);  // RY x TX (L,H) pairs

    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const int xa = 2 * bx * TX - C - PADL;  // multiple of 4
    const int y0 = 2 * by * TY - C;

    // ---- phase 1: stage rows [y0, y0+RY) x cols [xa, xa+RXA) with 16-B loads.  Branch-free where the image is even
    // and at least as large as the region: a constant number of trips, indices past the end clamped (those threads
    // re-write the last quad with the same value), the periodic wrap as one conditional add/sub -- so a thread's
    // loads are all in flight together (a loop with an exit test per trip waits for every load before the next).
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int TOTAL = RY * V4, TRIPS = (TOTAL + NT - 1) / NT;
        if (coh_in) {
            const CohPlane pin = coh_plane(in);
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                int sy = y0 + r, sx = xa + 4 * g;
                sy = sy < 0 ? sy + a.Nr : (sy >= a.Nr ? sy - a.Nr : sy);
                sx = sx < 0 ? sx + a.Nc : (sx >= a.Nc ? sx - a.Nc : sx);
                const real4_t v = coh_load16(pin, (long long)sy * a.Nc + sx);
                *reinterpret_cast<real4_t*>(sIn + r * RXA + 4 * g) = v;
            }
        } else if (!(a.Nr & 1) && !(a.Nc & 3) && a.Nr >= RY && a.Nc >= RXA) {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                int sy = y0 + r, sx = xa + 4 * g;  // Nc % 4 == 0: the group never straddles
                sy = sy < 0 ? sy + a.Nr : (sy >= a.Nr ? sy - a.Nr : sy);
                sx = sx < 0 ? sx + a.Nc : (sx >= a.Nc ? sx - a.Nc : sx);
                const real4_t v = *reinterpret_cast<const real4_t*>(in + (long long)sy * a.Nc + sx);
                *reinterpret_cast<real4_t*>(sIn + r * RXA + 4 * g) = v;
            }
        } else if (!(a.Nc & 3)) {
            for (int idx = tid; idx < RY * V4; idx += NT) {
                const int r = idx / V4;
                const int g = idx - r * V4;
                const int sy = wrap_analysis(y0 + r, a.Nr);
                const int sx = wrap_periodic(xa + 4 * g, a.Nc);
                const real4_t v = *reinterpret_cast<const real4_t*>(in + (long long)sy * a.Nc + sx);
                *reinterpret_cast<real4_t*>(sIn + r * RXA + 4 * g) = v;
            }
        } else {
            // Row length not a multiple of 4 (odd images, the reference takes them at no extra cost,
            // pdwt/src/separable.cu:116-121): rows start at any 4-B offset.  Groups inside the row are ONE 16-B load at
            // 4-B alignment; only the groups that touch the row ends gather element by element through the analysis
            // extension (last sample repeated for an odd length, then periodic).  Constant trip count, clamped index: a
            // thread's loads are in flight together.
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                const real_t* PDWT_RESTRICT rowp = in + (long long)wrap_analysis(y0 + r, a.Nr) * a.Nc;
                const int sx = xa + 4 * g;
                real4_t v;
                if (sx >= 0 && sx + 3 < a.Nc) {
                    v = ld_f4_unaligned(rowp + sx);
                } else {
                    v.x = rowp[wrap_analysis(sx, a.Nc)];
                    v.y = rowp[wrap_analysis(sx + 1, a.Nc)];
                    v.z = rowp[wrap_analysis(sx + 2, a.Nc)];
                    v.w = rowp[wrap_analysis(sx + 3, a.Nc)];
                }
                *reinterpret_cast<real4_t*>(sIn + r * RXA + 4 * g) = v;
            }
        }
    }
    PDWT_SYNC();

    // ---- phase 2: row analysis -> interleaved (L,H) pairs in LDS
    PDWT_FOR_THREADS(tid, NT) { fwd_fast_row_pass<HLEN, TX, TY, NT>(tid, sIn, tLH, a.fb); }
    PDWT_SYNC();

    // ---- phase 3: column analysis -> A, H, V, D
    PDWT_FOR_THREADS(tid, NT) { fwd_fast_col_pass<HLEN, TX, TY, NT>(tid, tLH, a, bx, by, bz, coh_out); }
}

// Streaming variant: a workgroup walks over several tiles of its XCD's band and issues the 16-B
// global loads of tile i+1 into registers BEFORE it computes tile i (row + column pass), so the
// HBM/L2 latency of the next tile hides behind the arithmetic of the current one.  The one-tile
// kernel above is latency-bound (occupancy x bytes in flight / latency ~ 4 TB/s of reads+writes,
// profiles/r01b_*); this one keeps every resident workgroup's loads in flight all the time.
// Issue the 16-B loads of one forward tile into the staging registers.  Interior tiles (no
// periodic wrap in either direction: all but the border ring) skip the wrap arithmetic, which was
// ~30 % of the kernel's vector instructions.
template <int HLEN, int TX, int TY, int NT, int NLD>
PDWT_DEVICE void fwd_fast_issue_loads(int tid, const Fwd2DFastArgs& a, int bx, int by, int bz, v4f* st) {
    using G = FwdFastGeom<HLEN, TX>;
    constexpr int C = G::C, PADL = G::PADL, RXA = G::RXA;
    constexpr int RY = 2 * TY + HLEN - 2;
    constexpr int V4 = RXA / 4;
    const real_t* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const int xa = 2 * bx * TX - C - PADL, y0 = 2 * by * TY - C;
    const bool interior = xa >= 0 && xa + RXA <= a.Nc && y0 >= 0 && y0 + RY <= a.Nr;
    if (interior) {
        const real_t* base = in + (long long)y0 * a.Nc + xa;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = (tid + i * NT < RY * V4) ? tid + i * NT : RY * V4 - 1;
            const int r = idx / V4, g = idx - r * V4;
            st[i] = *reinterpret_cast<const v4f*>(base + (long long)r * a.Nc + 4 * g);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            // lanes past the end of the tile reload its last group (branch-free: the staging
            // registers must stay in VGPRs; a divergent conditional load sends them to scratch)
            const int idx = (tid + i * NT < RY * V4) ? tid + i * NT : RY * V4 - 1;
            const int r = idx / V4, g = idx - r * V4;
            const int sy = wrap_analysis(y0 + r, a.Nr);
            const int sx = wrap_periodic(xa + 4 * g, a.Nc);  // Nc % 4 == 0: a group never straddles
            st[i] = *reinterpret_cast<const v4f*>(in + (long long)sy * a.Nc + sx);
        }
    }
}

// Position q of a workgroup's walk -> (image, tile).  One XCD's share of the work is the sequence
// q = 0 .. batch*chunk-1 : image q / chunk, tile xcd*chunk + q % chunk of that image.  All workgroups
// advance through the images TOGETHER, so at any time the chip streams through one image (a few
// bands per XCD) rather than through `batch` of them at once (which measured 20 % slower on the
// inverse: too many concurrent DRAM streams).
PDWT_DEVICE bool stream_pos(int q, int xcd, int chunk, int total, int batch, int tiles_x, int& bx, int& by,
                            int& bz) {
    if (q >= batch * chunk) return false;
    bz = q / chunk;
    const int tile = xcd * chunk + (q - bz * chunk);
    if (tile >= total) return false;
    by = tile / tiles_x;
    bx = tile - by * tiles_x;
    return true;
}

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_fwd_fast_stream(const Fwd2DFastArgs& a, int wg, int nwg, int batch, real_t* smem) {
    using G = FwdFastGeom<HLEN, TX>;
    constexpr int RXA = G::RXA;
    constexpr int RY = 2 * TY + HLEN - 2;
    constexpr int V4 = RXA / 4;
    constexpr int NLD = (RY * V4 + NT - 1) / NT;  // float4 loads per thread per tile

    real_t* sIn = smem;
    v2f* tLH = reinterpret_cast<v2f*>(smem + RY * RXA);

    const int total = a.tiles_x * a.tiles_y;
    const int chunk = (total + 7) >> 3;
    const int xcd = wg & 7;       // ids b and b+8 share an XCD (speed only)
    const int stride = nwg >> 3;  // workgroups per XCD
    const int limit = batch * chunk;

    PDWT_PER_THREAD(v4f, stage, NLD, NT);

    int q = wg >> 3;
    int bx = 0, by = 0, bz = 0;
    while (q < limit && !stream_pos(q, xcd, chunk, total, batch, a.tiles_x, bx, by, bz)) q += stride;
    bool have = q < limit;
    if (have) {
        PDWT_FOR_THREADS(tid, NT) {
            fwd_fast_issue_loads<HLEN, TX, TY, NT, NLD>(tid, a, bx, by, bz, PDWT_MINE(stage, tid));
        }
    }
    while (have) {
        PDWT_FOR_THREADS(tid, NT) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + i * NT;
                if (idx < RY * V4) *reinterpret_cast<v4f*>(sIn + 4 * idx) = PDWT_MINE(stage, tid)[i];
            }
        }
        PDWT_SYNC();
        // prefetch the next tile of this workgroup
        int qn = q + stride;
        int nbx = 0, nby = 0, nbz = 0;
        while (qn < limit && !stream_pos(qn, xcd, chunk, total, batch, a.tiles_x, nbx, nby, nbz)) qn += stride;
        const bool have_next = qn < limit;
        if (have_next) {
            PDWT_FOR_THREADS(tid, NT) {
                fwd_fast_issue_loads<HLEN, TX, TY, NT, NLD>(tid, a, nbx, nby, nbz, PDWT_MINE(stage, tid));
            }
        }
        PDWT_FOR_THREADS(tid, NT) { fwd_fast_row_pass<HLEN, TX, TY, NT>(tid, sIn, tLH, a.fb); }
        PDWT_SYNC();
        PDWT_FOR_THREADS(tid, NT) { fwd_fast_col_pass<HLEN, TX, TY, NT>(tid, tLH, a, bx, by, bz); }
        bx = nbx;
        by = nby;
        bz = nbz;
        q = qn;
        have = have_next;
    }
}

// ---------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------
template <int HLEN, int TX>
struct InvFastGeom {
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int S = (H2 & 1) ? 0 : 1;
    static constexpr int PADL = (4 - (C & 3)) & 3;  // cx0 = bx*TX - C ; aligned origin = cx0 - PADL (multiple of 4)
    static constexpr int CXA = (PADL + TX + H2 + 1 + 3) & ~3;
};

template <int HLEN, int TX, int TY>
constexpr int inv2d_fast_lds_floats() {
    using G = InvFastGeom<HLEN, TX>;
    return 4 * (TY + G::H2 + 1) * G::CXA + 2 * (2 * TY) * G::CXA;
}

// Column synthesis core: H2 consecutive coefficient rows (row stride `stride` pairs) of two adjacent
// columns, given as (A,V) and (H,D) pairs -> the (t1,t2) pairs of output rows p even (e) / p odd (o).
template <int HLEN>
PDWT_DEVICE void inv_col_synth2(const v2f* pAV, const v2f* pHD, int stride, const FilterBankI& fb, v2f& e0, v2f& o0,
                                v2f& e1, v2f& o1) {
    constexpr int H2 = HLEN / 2;
    e0 = mk2(real_t(0), real_t(0));
    o0 = e0; e1 = e0; o1 = e0;
    constexpr int GB = 6;  // 2 GB LDS loads are issued, then consumed
#pragma unroll
    for (int j0 = 0; j0 < H2; j0 += GB) {
        v4f wav[GB], whd[GB];
#pragma unroll
        for (int g = 0; g < GB; ++g)
            if (j0 + g < H2) {
                wav[g] = lds_load16(pAV + (j0 + g) * stride);
                whd[g] = lds_load16(pHD + (j0 + g) * stride);
            }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int j = j0 + g;
            if (j < H2) {
                lds_pin(wav[g]);
                lds_pin(whd[g]);
                const v2f te = fb.t[HLEN - 2 - 2 * j];  // p even: par = 1
                const v2f to = fb.t[HLEN - 1 - 2 * j];  // p odd : par = 0
                const v2f av0 = mk2(wav[g].x, wav[g].y), av1 = mk2(wav[g].z, wav[g].w);
                const v2f hd0 = mk2(whd[g].x, whd[g].y), hd1 = mk2(whd[g].z, whd[g].w);
                e0 = fma2(av0, bc(te.x), e0); e0 = fma2(hd0, bc(te.y), e0);
                o0 = fma2(av0, bc(to.x), o0); o0 = fma2(hd0, bc(to.y), o0);
                e1 = fma2(av1, bc(te.x), e1); e1 = fma2(hd1, bc(te.y), e1);
                o1 = fma2(av1, bc(to.x), o1); o1 = fma2(hd1, bc(to.y), o1);
            }
        }
    }
}

// Row synthesis core: the (t1,t2) pairs from `base` (16-B aligned, PE = PADL & 1 pairs before the first
// one used) -> the four samples 2k .. 2k+3 of coefficient columns k, k+1.
template <int HLEN, int PADL>
PDWT_DEVICE void inv_row_synth4(const v2f* base, const FilterBankI& fb, real_t res[4]) {
    constexpr int H2 = HLEN / 2, S = (H2 & 1) ? 0 : 1;
    constexpr int PE = PADL & 1;
    constexpr int NP = (PE + H2 + 2 + 1) & ~1;
    v2f u[NP];
    v4f w[NP / 2];
#pragma unroll
    for (int q = 0; q < NP / 2; ++q) w[q] = lds_load16(base + 2 * q);
#pragma unroll
    for (int q = 0; q < NP / 2; ++q) {
        lds_pin(w[q]);
        u[2 * q] = mk2(w[q].x, w[q].y);
        u[2 * q + 1] = mk2(w[q].z, w[q].w);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {  // coefficient column k + kk -> samples 2(k+kk), 2(k+kk)+1
        v2f r0 = mk2(real_t(0), real_t(0)), r1 = mk2(real_t(0), real_t(0));
        if (S == 0) {
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const v2f x = u[PE + kk + j];
                r0 = fma2(x, fb.t[HLEN - 2 - 2 * j], r0);  // p even
                r1 = fma2(x, fb.t[HLEN - 1 - 2 * j], r1);  // p odd
            }
        } else {
#pragma unroll
            for (int j = 0; j < H2 + 1; ++j) {
                const v2f x = u[PE + kk + j];
                if (j < H2) r0 = fma2(x, fb.t[HLEN - 1 - 2 * j], r0);        // p = 2k+1 (odd), base k
                if (j >= 1) r1 = fma2(x, fb.t[HLEN - 2 - 2 * (j - 1)], r1);  // p = 2k+2 (even), base k+1
            }
        }
        res[2 * kk] = r0.x + r0.y;
        res[2 * kk + 1] = r1.x + r1.y;
    }
}

// column synthesis.  Work item = (m, column pair): the two output rows p = 2m, 2m+1 (p = gy + S)
// share the coefficient rows m .. m+H2-1 (local); even taps feed p odd, odd taps feed p even.
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void inv_fast_col_pass(int tid, const v2f* sAV, const v2f* sHD, v2f* tt, const FilterBankI& fb) {
    using G = InvFastGeom<HLEN, TX>;
    [[maybe_unused]] constexpr int H2 = G::H2, S = G::S, CXA = G::CXA;
    constexpr int OY = 2 * TY;
    constexpr int NM = TY + S;  // m = 0 .. TY-1+S
    constexpr int Q2 = CXA / 2;
    for (int idx = tid; idx < NM * Q2; idx += NT) {
        const int m = idx / Q2;
        const int q = 2 * (idx - m * Q2);
        v2f e0, o0, e1, o1;  // p even / p odd, columns q / q+1
        inv_col_synth2<HLEN>(&sAV[m * CXA + q], &sHD[m * CXA + q], CXA, fb, e0, o0, e1, o1);
        const int ge = 2 * m - S, go = 2 * m + 1 - S;  // local output rows
        real4_t w;
        if (ge >= 0 && ge < OY) {
            w.x = e0.x; w.y = e0.y; w.z = e1.x; w.w = e1.y;
            *reinterpret_cast<real4_t*>(&tt[ge * CXA + q]) = w;
        }
        if (go >= 0 && go < OY) {
            w.x = o0.x; w.y = o0.y; w.z = o1.x; w.w = o1.y;
            *reinterpret_cast<real4_t*>(&tt[go * CXA + q]) = w;
        }
    }
}

// row synthesis, four adjacent samples per thread, 16-B stores
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void inv_fast_row_pass(int tid, const v2f* tt, const Inv2DFastArgs& a, int bx, int by, int bz,
                                   bool coh_out = false) {
    using G = InvFastGeom<HLEN, TX>;
    [[maybe_unused]] constexpr int H2 = G::H2, S = G::S, PADL = G::PADL, CXA = G::CXA;
    constexpr int OY = 2 * TY;
    constexpr int HT = TX / 2;
    constexpr int PE = PADL & 1;                // read origin rounded down to an even pair index
    real_t* PDWT_RESTRICT out = a.out + (long long)bz * a.out_bstride;
    for (int idx = tid; idx < OY * HT; idx += NT) {
        const int gy = idx / HT;
        const int k = 2 * (idx - gy * HT);
        real_t res[4];
        inv_row_synth4<HLEN, PADL>(tt + gy * CXA + (PADL - PE) + k, a.fb, res);
        const int oy = 2 * by * TY + gy;
        const int ox = 2 * (bx * TX + k);
        if ((a.Nc & 3) || (a.out_bstride & 3)) {  // output rows of any length: element stores, each inside the row
            if (oy < a.Nr) {
                real_t* PDWT_RESTRICT orow = out + (long long)oy * a.Nc;
                if (ox + 3 < a.Nc) {
                    st_f4_unaligned(orow + ox, res[0], res[1], res[2], res[3]);  // 16 B at 4-B alignment
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ox + e < a.Nc) orow[ox + e] = res[e];
                }
            }
        } else if (oy < a.Nr && ox < a.Nc) {  // Nc % 4 == 0: the float4 is inside and aligned
            real4_t v;
            v.x = res[0]; v.y = res[1]; v.z = res[2]; v.w = res[3];
            if (coh_out) coh_store16(coh_plane(out), (long long)oy * a.Nc + ox, v);  // read by other workgroups of this launch
            else *reinterpret_cast<real4_t*>(out + (long long)oy * a.Nc + ox) = v;
        }
    }
}

// interleave one float4 of A,V and of H,D into the (A,V) / (H,D) pair planes
PDWT_DEVICE void inv_fast_interleave(v2f* sAV, v2f* sHD, int pair_index, const v4f& vA, const v4f& vV,
                                     const v4f& vH, const v4f& vD) {
    real4_t w;
    real4_t* dAV = reinterpret_cast<real4_t*>(sAV + pair_index);
    real4_t* dHD = reinterpret_cast<real4_t*>(sHD + pair_index);
    w.x = vA.x; w.y = vV.x; w.z = vA.y; w.w = vV.y; dAV[0] = w;
    w.x = vA.z; w.y = vV.z; w.z = vA.w; w.w = vV.w; dAV[1] = w;
    w.x = vH.x; w.y = vD.x; w.z = vH.y; w.w = vD.y; dHD[0] = w;
    w.x = vH.z; w.y = vD.z; w.z = vH.w; w.w = vD.w; dHD[1] = w;
}

// Requirements: HLEN even, Ncc % 4 == 0, Nc == 2*Ncc, 16-B aligned coefficient rows.
// coh_in / coh_out (uniform, dwt2_chain_kernels.hpp): the A plane was written by other workgroups of THIS launch (sc1
// loads) / the output plane is read by other workgroups of this launch (sc1 stores).
template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_inv_fast_tile(const Inv2DFastArgs& a, int bx, int by, int bz, real_t* smem, bool coh_in = false,
                                    bool coh_out = false) {
    using G = InvFastGeom<HLEN, TX>;
    constexpr int H2 = G::H2, C = G::C, PADL = G::PADL, CXA = G::CXA;
    constexpr int CR = TY + H2 + 1;
    constexpr int V4 = CXA / 4;

    v2f* sAV = reinterpret_cast<v2f*>(smem);  // CR x CXA (A,V) pairs
    v2f* sHD = sAV + CR * CXA;                // CR x CXA (H,D) pairs
    v2f* tt = sHD + CR * CXA;                 // 2TY x CXA (t1,t2) pairs

    const long long boff = (long long)bz * a.in_bstride;
    const int cy0 = by * TY - C;
    const int cxa = bx * TX - C - PADL;  // multiple of 4

    // ---- phase 1: stage the four coefficient tiles, interleaved as (A,V) and (H,D)
    // interior tiles (all but the border ring) skip the periodic-wrap arithmetic (a uniform branch)
    const bool quads = !(a.Ncc & 3) && !(a.in_bstride & 3);  // coefficient rows start 16-B aligned, groups never straddle
    const bool interior = quads && cxa >= 0 && cxa + CXA <= a.Ncc && cy0 >= 0 && cy0 + CR <= a.Nrc;
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int TOTAL = CR * V4, TRIPS = (TOTAL + NT - 1) / NT;
        if (coh_in) {  // needs Nrc >= CR and Ncc >= CXA (checked by the host)
            const CohPlane pa = coh_plane(a.A + boff);
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                int sy = cy0 + r, sx = cxa + 4 * g;
                sy = sy < 0 ? sy + a.Nrc : (sy >= a.Nrc ? sy - a.Nrc : sy);
                sx = sx < 0 ? sx + a.Ncc : (sx >= a.Ncc ? sx - a.Ncc : sx);
                const long long oi = (long long)sy * a.Ncc + sx;
                const long long o = boff + oi;
                const real4_t ca = coh_load16(pa, oi);
                v4f va;
                va.x = ca.x; va.y = ca.y; va.z = ca.z; va.w = ca.w;
                inv_fast_interleave(sAV, sHD, r * CXA + 4 * g, va, *reinterpret_cast<const v4f*>(a.V + o),
                                    *reinterpret_cast<const v4f*>(a.H + o), *reinterpret_cast<const v4f*>(a.D + o));
            }
        } else if (interior) {  // branch-free (see dwt2_fwd_fast_tile): all of a thread's loads in flight together
            const long long o0 = boff + (long long)cy0 * a.Ncc + cxa;
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                const long long o = o0 + (long long)r * a.Ncc + 4 * g;
                inv_fast_interleave(sAV, sHD, r * CXA + 4 * g, *reinterpret_cast<const v4f*>(a.A + o),
                                    *reinterpret_cast<const v4f*>(a.V + o), *reinterpret_cast<const v4f*>(a.H + o),
                                    *reinterpret_cast<const v4f*>(a.D + o));
            }
        } else if (quads && a.Nrc >= CR && a.Ncc >= CXA) {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                int sy = cy0 + r, sx = cxa + 4 * g;
                sy = sy < 0 ? sy + a.Nrc : (sy >= a.Nrc ? sy - a.Nrc : sy);
                sx = sx < 0 ? sx + a.Ncc : (sx >= a.Ncc ? sx - a.Ncc : sx);
                const long long o = boff + (long long)sy * a.Ncc + sx;
                inv_fast_interleave(sAV, sHD, r * CXA + 4 * g, *reinterpret_cast<const v4f*>(a.A + o),
                                    *reinterpret_cast<const v4f*>(a.V + o), *reinterpret_cast<const v4f*>(a.H + o),
                                    *reinterpret_cast<const v4f*>(a.D + o));
            }
        } else if (!quads) {
            // coefficient rows of any length / alignment (odd image sizes): groups inside the row are one 16-B load at 4-B
            // alignment per band, groups at the row ends gather element by element (periodic); constant trip count
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4;
                const int g = idx - r * V4;
                const long long ro = boff + (long long)wrap_periodic(cy0 + r, a.Nrc) * a.Ncc;
                const int sx = cxa + 4 * g;
                real4_t qa, qv, qh, qd;
                if (sx >= 0 && sx + 3 < a.Ncc) {
                    qa = ld_f4_unaligned(a.A + ro + sx); qv = ld_f4_unaligned(a.V + ro + sx);
                    qh = ld_f4_unaligned(a.H + ro + sx); qd = ld_f4_unaligned(a.D + ro + sx);
                } else {
                    const int s0 = wrap_periodic(sx, a.Ncc), s1 = wrap_periodic(sx + 1, a.Ncc), s2 = wrap_periodic(sx + 2, a.Ncc),
                              s3 = wrap_periodic(sx + 3, a.Ncc);
                    qa.x = a.A[ro + s0]; qa.y = a.A[ro + s1]; qa.z = a.A[ro + s2]; qa.w = a.A[ro + s3];
                    qv.x = a.V[ro + s0]; qv.y = a.V[ro + s1]; qv.z = a.V[ro + s2]; qv.w = a.V[ro + s3];
                    qh.x = a.H[ro + s0]; qh.y = a.H[ro + s1]; qh.z = a.H[ro + s2]; qh.w = a.H[ro + s3];
                    qd.x = a.D[ro + s0]; qd.y = a.D[ro + s1]; qd.z = a.D[ro + s2]; qd.w = a.D[ro + s3];
                }
                v4f va, vv, vh, vd;
                va.x = qa.x; va.y = qa.y; va.z = qa.z; va.w = qa.w;
                vv.x = qv.x; vv.y = qv.y; vv.z = qv.z; vv.w = qv.w;
                vh.x = qh.x; vh.y = qh.y; vh.z = qh.z; vh.w = qh.w;
                vd.x = qd.x; vd.y = qd.y; vd.z = qd.z; vd.w = qd.w;
                inv_fast_interleave(sAV, sHD, r * CXA + 4 * g, va, vv, vh, vd);
            }
        } else {
            for (int idx = tid; idx < CR * V4; idx += NT) {
                const int r = idx / V4;
                const int g = idx - r * V4;
                const long long o = boff + (long long)wrap_periodic(cy0 + r, a.Nrc) * a.Ncc +
                                    wrap_periodic(cxa + 4 * g, a.Ncc);
                inv_fast_interleave(sAV, sHD, r * CXA + 4 * g, *reinterpret_cast<const v4f*>(a.A + o),
                                    *reinterpret_cast<const v4f*>(a.V + o), *reinterpret_cast<const v4f*>(a.H + o),
                                    *reinterpret_cast<const v4f*>(a.D + o));
            }
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) { inv_fast_col_pass<HLEN, TX, TY, NT>(tid, sAV, sHD, tt, a.fb); }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) { inv_fast_row_pass<HLEN, TX, TY, NT>(tid, tt, a, bx, by, bz, coh_out); }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_fast_kernel(const Fwd2DFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int bx, by;
    if (!xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_fwd_fast_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_fast_stream_kernel(const Fwd2DFastArgs a, int batch) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    dwt2_fwd_fast_stream<HLEN, TX, TY, NT>(a, blockIdx.x, gridDim.x, batch, pdwt_smem);
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_fast_kernel(const Inv2DFastArgs a) {
    extern __shared__ __attribute__((aligned(16))) real_t pdwt_smem[];
    int bx, by;
    if (!xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_inv_fast_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
