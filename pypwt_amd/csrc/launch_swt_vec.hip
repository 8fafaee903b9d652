// launch_swt_vec.hip -- launchers of the vectorised a-trous level kernels (gfx950); a translation unit of
// its own so that the 20 unrolled filter lengths compile in parallel with the scalar kernels.
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#include "swt_kernels.hpp"

namespace pdwt {

// vectorised twins: 128 columns x 16 rows of one phase, four columns per thread (16-B accesses)
template <int HLEN, bool INV, int TX = 128, int TY = 16, int NT = 256>
static hipError_t run_vec(const Swt2DArgs& a, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    // the inverse stages its rows in LDS where the dilation allows it (swt_inv_staged): a larger request for those launches
    const bool staged = INV && swt_inv_staged<TX, TY, NT>(HLEN, a.f);
    const size_t lds = (size_t)swt2d_inv_vec_lds_floats<TX, TY, NT>(HLEN, staged) * sizeof(real_t);
    const int M = cdiv(a.Nr, a.f);  // rows of the longest dilation phase
    const int total = cdiv(a.Nc, TX) * cdiv(M, TY) * a.f;
    dim3 grid(8 * ((total + 7) / 8), batch);  // XCD-aware tile order, see swt_vec_tile
    if (INV) {
        hipError_t e = allow_big_lds(swt2_inv_vec_kernel<HLEN, TX, TY, NT>, lds, big);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_inv_vec_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, s, a);
    } else {
        hipError_t e = allow_big_lds(swt2_fwd_vec_kernel<HLEN, TX, TY, NT>, lds, big);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_fwd_vec_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, s, a);
    }
    return hipGetLastError();
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(real_t) - 1)) == 0; }

// preconditions of the vectorised kernels: even compile-time filter length, planes aligned for 4-element accesses.  Rows of any
// length since round 5 (16-B accesses at 4-B alignment, the partial quad at a row's end element by element) -- where that
// pays: measured on 1002^2 / 2047^2 (profiles/r05d_cliffs_swt_odd.txt) the unaligned forward is within 1.1x of the aligned one
// from 6 taps on and the inverse where it stages its band rows in LDS; 2- and 4-tap levels and unstaged inverses are faster on
// the one-column-per-thread tiles (haar 1002^2 forward 14.3 against 18.8 us).
static bool vec_ok(const Swt2DArgs& a, bool inverse) {
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > kMaxTaps || a.Nc < 4) return false;
    if ((a.Nc & 3) || (a.bstride & 3)) {
        if (sizeof(real_t) != 4) return false;
        if (inverse ? !swt_inv_staged_filter(a.hlen, a.f) : a.hlen < 6) return false;
    }
    if (!al16(a.A) || !al16(a.H) || !al16(a.V) || !al16(a.D)) return false;
    return al16(inverse ? (const void*)a.out : (const void*)a.in);
}

hipError_t try_launch_swt2_vec(const Swt2DArgs& a, bool inverse, int batch, hipStream_t s) {
    if (!vec_ok(a, inverse)) return hipErrorNotSupported;
    // Deep levels (round 4): a tile is TY rows of ONE dilation phase, and a phase has only Nr / f rows -- at the last levels of a
    // maximum-level plan (2048^2 haar L11: f = 1024, two rows per phase) a 16-row tile is 7/8 padding: forward level 11 34 us,
    // inverse 51 us against 16-20 us for the levels before.  Phases shorter than 16 rows take tiles of 8, 4 or 2 rows (2-8 taps).
    {
        static const bool deep_tiles = !(lab_env("PDWT_SWT_DEEP_TILE") && atoi(lab_env("PDWT_SWT_DEEP_TILE")) == 0);  // A/B measurements
        const int M = cdiv(a.Nr, a.f);
        if (deep_tiles && M < 16 && a.hlen <= 8 && sizeof(real_t) == 4) {
            switch (a.hlen) {
#define X(h)                                                                                                                   \
    case h:                                                                                                                    \
        if (M >= 8) return inverse ? run_vec<h, true, 128, 8, 256>(a, batch, s) : run_vec<h, false, 128, 8, 256>(a, batch, s); \
        if (M >= 4) return inverse ? run_vec<h, true, 128, 4, 128>(a, batch, s) : run_vec<h, false, 128, 4, 128>(a, batch, s); \
        return inverse ? run_vec<h, true, 128, 2, 64>(a, batch, s) : run_vec<h, false, 128, 2, 64>(a, batch, s);
                X(2) X(4) X(6) X(8)
#undef X
            }
        }
    }
    // short filters: 256-column tiles (1 KiB contiguous per row and band; cfg4 in-step 182 -> 176 us, although
    // a level repeated on cache-resident data is faster with 128: 16.5 vs 18.4 us)
    if (a.Nc >= 512 && a.hlen == 2) return inverse ? run_vec<2, true, 256>(a, batch, s) : run_vec<2, false, 256>(a, batch, s);
    if (a.Nc >= 512 && a.hlen == 4) return inverse ? run_vec<4, true, 256>(a, batch, s) : run_vec<4, false, 256>(a, batch, s);
    // Tile height (profiles/r03l_swt_tile_height.txt, 2048^2, three levels forward): 16 rows are best up to 24 taps (db5 95 us
    // against 100 / 103 with 32 / 64 rows, sym8 123 / 142 / 137, db12 212 / 238 / 357); 40 taps re-filter 39 halo rows per
    // tile and gain with 32 rows (910 -> 773 us).
    // Small levels (round 4): 128 x 16 tiles give a 256^2 image 32 workgroups on 256 CUs, each a long serial chain -- the time
    // of a level is one tile's latency (db4 256^2: 7.6 us forward, 13.4 us inverse per level).  Below 256 workgroups the
    // tiles shrink to 64 x 8 (128 threads): four times the workgroups, each a quarter of the chain.
    static const int small_tiles = lab_env("PDWT_SWT_SMALL_TILE") ? atoi(lab_env("PDWT_SWT_SMALL_TILE")) : 256;  // workgroups; A/B measurements
    const long long wgs = (long long)cdiv(a.Nc, 128) * cdiv(cdiv(a.Nr, a.f), 16) * a.f * batch;
    // (and, whatever the batch, on images of at most 64 columns: a 128-column tile there is half padding)
    if ((wgs < small_tiles || (a.Nc <= 64 && small_tiles > 0)) && a.hlen <= 24 && sizeof(real_t) == 4) {
        switch (a.hlen) {
#define X(h)                                                                                                   \
    case h:                                                                                                    \
        if constexpr (h <= 24)                                                                                 \
            return inverse ? run_vec<h, true, 64, 8, 128>(a, batch, s) : run_vec<h, false, 64, 8, 128>(a, batch, s); \
        break;
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    // Round quantisation (round 6; profiles/r06_sizes_cliff.txt, r06_swt_round_scan.txt, r06_dispatchprobe.txt): a level of 512 of
    // these tiles is two per CU -- one wavefront of each on every SIMD -- and 17.9 us (1024^2, 14 taps); 520 tiles (1032 rows) put a
    // third wavefront on the SIMDs of eight CUs and the level takes 22.6 us, the same as 768 tiles (24.6): a level lasts as long as its
    // fullest SIMD.  Narrower tiles (64 or 32 columns: 2080 one-wavefront workgroups, 8-9 per CU) were built and measured: 22.2 us --
    // the unit that fills up is the SIMD, not the CU, and 2080 wavefronts on 1024 SIMDs are 2-3 per SIMD like 520 x 4.  What would help
    // is bands of unequal height (64 bands of 16-17 rows: exactly two wavefronts per SIMD again); not built.
    // fp64, more than 24 taps: a 128-column tile of doubles is 112-145 KB of LDS -- one workgroup of four wavefronts per CU.  64 x 16
    // tiles of 256 threads are 56 KB: two workgroups, eight wavefronts (round 5; forward + inverse, three levels: 26 taps 1024^2
    // 677 -> 333 us, 32 taps 547 -> 230, 40 taps 512^2 1003 -> 483 -- but 40 taps 2048^2 3288 -> 3934: images below 2^22 samples only;
    // profiles/r05e_f64_ab.txt)
    if (sizeof(real_t) == 8 && a.hlen > 24 && (long long)a.Nr * a.Nc * batch < (1LL << 22)) {
        switch (a.hlen) {
#define X(h)                                                                                                   \
    case h:                                                                                                    \
        if constexpr (h > 24)                                                                                  \
            return inverse ? run_vec<h, true, 64, 16, 256>(a, batch, s) : run_vec<h, false, 64, 16, 256>(a, batch, s); \
        break;
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    switch (a.hlen) {
#define X(h)                                                                                                   \
    case h:                                                                                                    \
        if constexpr (h > 24) {                                                                                \
            if (cdiv(a.Nr, a.f) >= 32)                                                                             \
                return inverse ? run_vec<h, true, 128, 32>(a, batch, s) : run_vec<h, false, 128, 32>(a, batch, s); \
        }                                                                                                      \
        return inverse ? run_vec<h, true>(a, batch, s) : run_vec<h, false>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
