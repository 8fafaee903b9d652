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
static bool mid_size(long long samples) { return samples > (1LL << 20) && samples <= (1LL << 22); }

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
    static const bool off = getenv("PDWT_NO_LDS_TILES") != nullptr;
    return off;
}

hipError_t try_launch_dwt2_fwd_fast(const Fwd2DArgs& a, int batch, hipStream_t s) {
    if (lds_tiles_off()) return hipErrorNotSupported;
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > kMaxTaps) return hipErrorNotSupported;
    // any row length: rows that are not whole, aligned quads take the unaligned staging / element-store branches of the
    // tile; the streaming form (more than 20 taps) stages whole aligned quads only
    const bool quads = !(a.Nc & 3) && !(a.in_bstride & 3) && !(a.out_bstride & 1);
    if (a.hlen > 20 && !quads) return hipErrorNotSupported;
    if (!aligned16(a.in) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    // One tile per workgroup everywhere.  Since the staging loops are branch-free (all of a thread's loads in flight
    // together) the plain tile kernel is ahead of the streaming form (persistent workgroups that prefetch the next tile
    // into registers) at every size measured -- db4, one 4096^2 image: 21.0 us (64x8 tiles, 256 threads) against 23.4;
    // 2048^2: 7.0 us (64x16, 512 threads) against 8.6; 1024^2: 4.1 against 4.6; 4 x 4096^2: 107 against 116 us; 16 taps,
    // 4096^2: 29.2 us (64x16, 512 threads) against 38.5 (profiles/r02y_kbench_tiles.txt).
    if (a.hlen <= 8 && mid_size((long long)batch * a.Nr * a.Nc)) {
        switch (a.hlen) {
            case 2: return run_fwd_fast_tile<2, 64, 16, 512>(a, batch, s);
            case 4: return run_fwd_fast_tile<4, 64, 16, 512>(a, batch, s);
            case 6: return run_fwd_fast_tile<6, 64, 16, 512>(a, batch, s);
            case 8: return run_fwd_fast_tile<8, 64, 16, 512>(a, batch, s);
        }
    }
    switch (a.hlen) {
#define X(h)                                                                \
    case h:                                                                 \
        if constexpr (h <= 8) return run_fwd_fast_tile<h, 64, 8, 256>(a, batch, s);   \
        else if constexpr (h <= 20) return run_fwd_fast_tile<h, 64, 16, 512>(a, batch, s); \
        else if constexpr (sizeof(real_t) == 8) return hipErrorNotSupported; /* 32-row tiles of doubles exceed 160 KB */ \
        else return run_fwd_fast<h, 64, 32, 256>(a, batch, s);
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
        static const int forced = getenv("PDWT_INV_TILE") ? atoi(getenv("PDWT_INV_TILE")) : 0;
        if (forced && a.hlen == 8) {
            if (forced == 1) return run_inv_fast<8, 128, 16, 512>(a, batch, s);
            if (forced == 2) return run_inv_fast<8, 128, 8, 256>(a, batch, s);
            if (forced == 3) return run_inv_fast<8, 64, 16, 512>(a, batch, s);
            if (forced == 4) return run_inv_fast<8, 64, 32, 512>(a, batch, s);
        }
    }
    if (a.hlen <= 8 && mid_size((long long)batch * a.Nr * a.Nc)) {  // 128x16 tiles, 512 threads: 8.0 vs 8.8 us at 2048^2
        switch (a.hlen) {
            case 2: return run_inv_fast<2, 128, 16, 512>(a, batch, s);
            case 4: return run_inv_fast<4, 128, 16, 512>(a, batch, s);
            case 6: return run_inv_fast<6, 128, 16, 512>(a, batch, s);
            case 8: return run_inv_fast<8, 128, 16, 512>(a, batch, s);
        }
    }
    switch (a.hlen) {
#define X(h)                                                                \
    case h:                                                                 \
        if constexpr (h <= 8) return run_inv_fast<h, 64, 8, 256>(a, batch, s);   \
        else if constexpr (h <= 20) return run_inv_fast<h, 64, 16, 512>(a, batch, s); \
        else if constexpr (sizeof(real_t) == 8) return hipErrorNotSupported;     \
        else return run_inv_fast<h, 64, 32, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
