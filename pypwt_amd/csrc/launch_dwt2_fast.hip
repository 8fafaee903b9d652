// launch_dwt2_fast.hip -- instantiations + launchers of the tuned 2D DWT level kernels (gfx950).
//
// try_launch_* return hipErrorNotSupported when the level does not meet the fast kernels'
// preconditions (even compile-time filter length, row length multiple of 4, 16-B aligned rows);
// the caller then uses the generic kernels of launch_dwt2.hip.
//
// Tile shape per filter length (kbench on MI355X, profiles/r01b_kbench_*.txt): 64 output columns;
// short filters want SMALL tiles (8 output rows, ~23 KB LDS, 6 workgroups per CU): occupancy beats
// the larger halo; longer filters amortise their halo over 16 / 32 rows.
#include <cstdlib>

#include "dwt2_fast_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"
#include "tuning.hpp"

namespace pdwt {

static void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}

// Workgroups per image for the streaming kernels: as many as stay RESIDENT on the 256 CUs (LDS-limited,
// at most 6 per CU: measured optimum, profiles/r01c_kbench_stream.txt), never more than
// there are (image, tile) pairs; a multiple of 8 (one share per XCD).  Each workgroup then
// walks over tiles/workgroups tiles and prefetches the next one while it computes.
static int stream_workgroups(int tiles, int batch, size_t lds_bytes) {
    int per_cu = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
    if (per_cu > 6) per_cu = 6;
    if (per_cu < 1) per_cu = 1;
    int n = 256 * per_cu;
    const long long work = (long long)((tiles + 7) / 8) * 8 * (batch > 0 ? batch : 1);
    if (n > work) n = (int)work;
    n = (n + 7) & ~7;
    return n < 8 ? 8 : n;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifdef PDWT_TILE_EXPERIMENT  // the streaming form (persistent workgroups prefetching their next tile) is no longer dispatched
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd_fast(const Fwd2DArgs& g, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)fwd2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(real_t);
    hipError_t e = allow_big_lds(dwt2_fwd_fast_stream_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    Fwd2DFastArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.Nr2 = g.Nr2; a.Nc2 = g.Nc2;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.tiles_x = cdiv(g.Nc2, TX); a.tiles_y = cdiv(g.Nr2, TY);
    interleave(a.fb, g.fb);
    hipLaunchKernelGGL((dwt2_fwd_fast_stream_kernel<HLEN, TX, TY, NT>),
                       dim3(stream_workgroups(a.tiles_x * a.tiles_y, batch, lds)), dim3(NT), lds, s, a, batch);
    return hipGetLastError();
}

#endif

// One tile per workgroup (no streaming), for MID-SIZE levels: at 2048^2 (one image) 512 workgroups of 512
// threads with 64x32 tiles are all resident at once and the level takes 7.9 us instead of 8.5-9.0 us with the
// small streaming tiles, whose 2048 tiles need a second round of workgroups (profiles/r01c_kbench_2048.txt).
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd_fast_tile(const Fwd2DArgs& g, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)fwd2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(real_t);
    hipError_t e = allow_big_lds(dwt2_fwd_fast_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    Fwd2DFastArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.Nr2 = g.Nr2; a.Nc2 = g.Nc2;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.tiles_x = cdiv(g.Nc2, TX); a.tiles_y = cdiv(g.Nr2, TY);
    interleave(a.fb, g.fb);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    hipLaunchKernelGGL((dwt2_fwd_fast_kernel<HLEN, TX, TY, NT>), dim3(8 * chunk, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

// levels whose input is between 2^20 and 2^22 samples (one 2048^2 image): the single-round tile shapes
static bool mid_size(long long samples) { return samples > (1LL << tune::tile_small_level_log2) && samples <= (1LL << tune::tile_mid_hi_log2); }

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv_fast(const Inv2DArgs& g, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)inv2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(real_t);
    hipError_t e = allow_big_lds(dwt2_inv_fast_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    Inv2DFastArgs a;
    a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D; a.out = g.out;
    a.Nrc = g.Nrc; a.Ncc = g.Ncc; a.Nr = g.Nr; a.Nc = g.Nc;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.tiles_x = cdiv(g.Nc, 2 * TX); a.tiles_y = cdiv(g.Nr, 2 * TY);
    interleave(a.fb, g.fb);
    // One tile per workgroup.  The streaming (persistent + prefetch) form of this kernel,
    // dwt2_inv_fast_stream, measured SLOWER (25.4 vs 23.5 us at 4096^2, 280 vs 220 us for a batch of 8;
    // profiles/r01c_kbench_stream.txt): workgroups that start together stay in phase and alternate
    // read bursts with write bursts, while hardware dispatch naturally staggers them.
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    hipLaunchKernelGGL((dwt2_inv_fast_kernel<HLEN, TX, TY, NT>), dim3(8 * chunk, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

static bool lds_tiles_off() {  // A/B measurements: every level through the wave / generic kernels
    static const bool off = lab_env("PDWT_NO_LDS_TILES") != nullptr;
    return off;
}

hipError_t try_launch_dwt2_fwd_fast(const Fwd2DArgs& a, int batch, hipStream_t s) {
    if (lds_tiles_off()) return hipErrorNotSupported;
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > kMaxTaps) return hipErrorNotSupported;
    // any row length: rows that are not whole, aligned quads take the unaligned staging / element-store branches of the
    // tile
    if (!aligned16(a.in) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    // One tile per workgroup everywhere.  Since the staging loops are branch-free (all of a thread's loads in flight
    // together) the plain tile kernel is ahead of the streaming form (persistent workgroups that prefetch the next tile
    // into registers) at every size measured -- db4, one 4096^2 image: 21.0 us (64x8 tiles, 256 threads) against 23.4;
    // 2048^2: 7.0 us (64x16, 512 threads) against 8.6; 1024^2: 4.1 against 4.6; 4 x 4096^2: 107 against 116 us; 16 taps,
    // 4096^2: 29.2 us (64x16, 512 threads) against 38.5 (profiles/r02y_kbench_tiles.txt).
#ifdef PDWT_TILE_EXPERIMENT
    {   // A/B (16 and 40 taps only): PDWT_FWD_TILE = shape index
        static const int forced = lab_env("PDWT_FWD_TILE") ? atoi(lab_env("PDWT_FWD_TILE")) : 0;
#define PDWT_FT(h)                                                             \
        if (forced && a.hlen == h) {                                            \
            if (forced == 1) return run_fwd_fast<h, 64, 32, 256>(a, batch, s);       \
            if (forced == 2) return run_fwd_fast_tile<h, 64, 16, 512>(a, batch, s);  \
            if (forced == 3) return run_fwd_fast_tile<h, 32, 16, 256>(a, batch, s);  \
            if (forced == 4) return run_fwd_fast_tile<h, 32, 32, 512>(a, batch, s);  \
            if (forced == 5) return run_fwd_fast_tile<h, 64, 8, 256>(a, batch, s);   \
            if (forced == 6) return run_fwd_fast_tile<h, 32, 8, 128>(a, batch, s);   \
            if (forced == 7) return run_fwd_fast_tile<h, 64, 16, 256>(a, batch, s);  \
            if (forced == 8) return run_fwd_fast_tile<h, 64, 32, 512>(a, batch, s);  \
            if (forced == 9) return run_fwd_fast_tile<h, 64, 32, 1024>(a, batch, s); \
            if (forced == 10) return run_fwd_fast_tile<h, 64, 8, 128>(a, batch, s);  \
            if (forced == 11) return run_fwd_fast_tile<h, 32, 8, 64>(a, batch, s);   \
            if (forced == 12) return run_fwd_fast_tile<h, 128, 8, 512>(a, batch, s); \
            if (forced == 13) return run_fwd_fast_tile<h, 128, 16, 1024>(a, batch, s); \
        }
        PDWT_FT(4) PDWT_FT(8) PDWT_FT(12) PDWT_FT(16) PDWT_FT(18) PDWT_FT(20) PDWT_FT(22) PDWT_FT(24) PDWT_FT(26) PDWT_FT(28) PDWT_FT(30) PDWT_FT(32) PDWT_FT(36) PDWT_FT(40)
#undef PDWT_FT
    }
#endif
    // Batches of NARROW images (round 4): a tile wider than the level it works on computes padding -- 1024 images of 128^2,
    // level 2 (32 output columns): 33.7 us on 64-column tiles.  Levels of at most 32 output columns take 32 x 16 tiles.
    if (a.hlen <= 8 && a.Nc2 <= 32 && sizeof(real_t) == 4) {
        switch (a.hlen) {
            case 2: return run_fwd_fast_tile<2, 32, 16, 256>(a, batch, s);
            case 4: return run_fwd_fast_tile<4, 32, 16, 256>(a, batch, s);
            case 6: return run_fwd_fast_tile<6, 32, 16, 256>(a, batch, s);
            case 8: return run_fwd_fast_tile<8, 32, 16, 256>(a, batch, s);
        }
    }
    if (a.hlen <= 8 && mid_size((long long)batch * a.Nr * a.Nc)) {
        switch (a.hlen) {
            case 2: return run_fwd_fast_tile<2, 64, 16, 512>(a, batch, s);
            case 4: return run_fwd_fast_tile<4, 64, 16, 512>(a, batch, s);
            case 6: return run_fwd_fast_tile<6, 64, 16, 512>(a, batch, s);
            case 8: return run_fwd_fast_tile<8, 64, 16, 512>(a, batch, s);
        }
    }
    // Filters of 10 taps and more (round 4, profiles/r04i_tilesweep.txt, r04j_tilesweep2.txt; event-timed launches):
    //   * a level of fewer than 2^20 samples is ONE round of a few workgroups -- its time is one tile's serial chain, so
    //     small tiles of 32 x 16 outputs (256 threads) win whatever the halo they recompute: 40 taps 15.6 -> 7.9 us per
    //     level from 128^2 to 512^2 (the 64 x 32 streaming tiles of rounds 1-3 were sized for 4096^2), 26 taps 11.6 -> 6.9,
    //     20 taps 7.0 -> 6.6;
    //   * large levels of 18+ taps: 32 x 32 outputs with 512 threads (each thread one column pair of one output row):
    //     40 taps 4096^2 87.0 -> 56.9 us, 2048^2 28.9 -> 20.4; 26 taps 58.8 -> 38.6; 20 taps 39.9 -> 34.9;
    //   * 10-16 taps keep 64 x 16 / 512 (16 taps 4096^2: 31.4 against 32.8).
    const bool small_level = (long long)batch * a.Nr * a.Nc < (1LL << tune::tile_small_level_log2);
    switch (a.hlen) {
#define X(h)                                                                \
    case h:                                                                 \
        if constexpr (h <= 8) return run_fwd_fast_tile<h, 64, 8, 256>(a, batch, s);   \
        else if constexpr (sizeof(real_t) == 8) {                                     \
            if constexpr (h <= 20) return run_fwd_fast_tile<h, 64, 16, 512>(a, batch, s); \
            else if (small_level) return run_fwd_fast_tile<h, 32, 16, 256>(a, batch, s); /* 70 x 168 doubles = 94 KB */ \
            else return run_fwd_fast_tile<h, 32, 32, 512>(a, batch, s); /* 137 KB: one workgroup per CU, but eight wavefronts instead of four */ \
        } else if (small_level) return run_fwd_fast_tile<h, 32, 16, 256>(a, batch, s);  \
        else if constexpr (h <= 16) return run_fwd_fast_tile<h, 64, 16, 512>(a, batch, s); \
        else return run_fwd_fast_tile<h, 32, 32, 512>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t try_launch_dwt2_inv_fast(const Inv2DArgs& a, int batch, hipStream_t s) {
    if (lds_tiles_off()) return hipErrorNotSupported;
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > kMaxTaps) return hipErrorNotSupported;
    if (a.Nc > 2 * a.Ncc || a.Nc < 2 * a.Ncc - 1 || a.Nr > 2 * a.Nrc) return hipErrorNotSupported;  // any row length / alignment
    if (!aligned16(a.out) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    {   // A/B measurements (db4 only): PDWT_INV_TILE = 1: 128x16 / 512 threads, 2: 128x8 / 256, 3: 64x16 / 512, 4: 64x32 / 512
        static const int forced = lab_env("PDWT_INV_TILE") ? atoi(lab_env("PDWT_INV_TILE")) : 0;
        if (forced && a.hlen == 8) {
            if (forced == 1) return run_inv_fast<8, 128, 16, 512>(a, batch, s);
            if (forced == 2) return run_inv_fast<8, 128, 8, 256>(a, batch, s);
            if (forced == 3) return run_inv_fast<8, 64, 16, 512>(a, batch, s);
            if (forced == 4) return run_inv_fast<8, 64, 32, 512>(a, batch, s);
        }
    }
#ifdef PDWT_TILE_EXPERIMENT
    {
        static const int forced = lab_env("PDWT_INV_TILE2") ? atoi(lab_env("PDWT_INV_TILE2")) : 0;
#define PDWT_IT(h)                                                             \
        if (forced && a.hlen == h) {                                            \
            if (forced == 1) return run_inv_fast<h, 64, 32, 256>(a, batch, s);   \
            if (forced == 2) return run_inv_fast<h, 64, 16, 512>(a, batch, s);   \
            if (forced == 3) return run_inv_fast<h, 32, 16, 256>(a, batch, s);   \
            if (forced == 4) return run_inv_fast<h, 32, 16, 512>(a, batch, s);   \
            if (forced == 5) return run_inv_fast<h, 64, 8, 256>(a, batch, s);    \
            if (forced == 6) return run_inv_fast<h, 32, 8, 256>(a, batch, s);    \
            if (forced == 7) return run_inv_fast<h, 64, 16, 1024>(a, batch, s);  \
            if (forced == 8) return run_inv_fast<h, 64, 32, 1024>(a, batch, s);  \
            if (forced == 9) return run_inv_fast<h, 32, 32, 1024>(a, batch, s);  \
            if (forced == 10) return run_inv_fast<h, 64, 8, 128>(a, batch, s);   \
            if (forced == 11) return run_inv_fast<h, 32, 8, 128>(a, batch, s);   \
            if (forced == 12) return run_inv_fast<h, 128, 8, 256>(a, batch, s);  \
            if (forced == 13) return run_inv_fast<h, 128, 16, 512>(a, batch, s); \
        }
        PDWT_IT(4) PDWT_IT(8) PDWT_IT(12) PDWT_IT(16) PDWT_IT(18) PDWT_IT(20) PDWT_IT(22) PDWT_IT(24) PDWT_IT(26) PDWT_IT(28) PDWT_IT(30) PDWT_IT(32) PDWT_IT(36) PDWT_IT(40)
#undef PDWT_IT
    }
#endif
    // narrow images (see the forward): 1024 images of 128^2, level 2 (32 coefficient columns) 63 us on the 128-column tiles
    if (a.hlen <= 8 && a.Ncc <= 32 && sizeof(real_t) == 4) {
        switch (a.hlen) {
            case 2: return run_inv_fast<2, 32, 8, 256>(a, batch, s);
            case 4: return run_inv_fast<4, 32, 8, 256>(a, batch, s);
            case 6: return run_inv_fast<6, 32, 8, 256>(a, batch, s);
            case 8: return run_inv_fast<8, 32, 8, 256>(a, batch, s);
        }
    }
    if (a.hlen <= 8 && a.Ncc > 64 && mid_size((long long)batch * a.Nr * a.Nc)) {  // 128x16 tiles, 512 threads: 8.0 vs 8.8 us at 2048^2
        switch (a.hlen) {
            case 2: return run_inv_fast<2, 128, 16, 512>(a, batch, s);
            case 4: return run_inv_fast<4, 128, 16, 512>(a, batch, s);
            case 6: return run_inv_fast<6, 128, 16, 512>(a, batch, s);
            case 8: return run_inv_fast<8, 128, 16, 512>(a, batch, s);
        }
    }
    // 10 taps and more (same sweeps as the forward): below 2^20 output samples 32 x 8 coefficient tiles of 256 threads
    // (40 taps 20.5 -> 6.9 us per level, 26 taps 13.0 -> 6.6, 16-20 taps 8.2 -> 6.6); large levels: 10-18 taps 64 x 16 /
    // 512 as before (18 taps 4096^2: 38.4 against 39.5-40.0); 26 and 30 taps 32 x 16 / 512 (76.0 -> 46.0, 30 taps 52.9
    // against 55.2); every other length 32 x 8 / 256 (20 taps 47.3 -> 44.0, 22: 47.8 -> 43.0, 24: 57.6 -> 52.1, 28: 65.3 ->
    // 60.6, 32: 97.3 -> 73.2, 36: 105.7 -> 73.8, 40: 131.7 -> 89.8, 2048^2 38.4 -> 29.2); profiles/r04p_tilesweep3.txt
    const bool small_level = (long long)batch * a.Nr * a.Nc < (1LL << tune::tile_small_level_log2);
    switch (a.hlen) {
#define X(h)                                                                \
    case h:                                                                 \
        if constexpr (h <= 8) return run_inv_fast<h, 64, 8, 256>(a, batch, s);   \
        else if constexpr (sizeof(real_t) == 8) {                                \
            if constexpr (h <= 20) return run_inv_fast<h, 64, 16, 512>(a, batch, s); \
            else return run_inv_fast<h, 32, 8, 256>(a, batch, s);                \
        } else if (small_level) return run_inv_fast<h, 32, 8, 256>(a, batch, s);  \
        else if constexpr (h <= 18) return run_inv_fast<h, 64, 16, 512>(a, batch, s); \
        else if constexpr (h == 26 || h == 30) return run_inv_fast<h, 32, 16, 512>(a, batch, s); \
        else return run_inv_fast<h, 32, 8, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
