// launch_dwt1_fused.hip -- launchers of the multi-level fused 1D DWT kernels (gfx950).
#include "dwt1_fused_kernels.hpp"
#include "dwt1_rows_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <cstdlib>

namespace pdwt {

static void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}

// Largest number of levels one forward launch may fuse for this filter length so that the two
// ping-pong LDS buffers of a TF = 64 workgroup stay within ~48 KB (>= 3 workgroups per CU).
int dwt1_fused_max_levels(int hlen) {
    int best = 1;
    for (int k = 2; k <= kMaxFusedLevels; ++k)
        if ((size_t)fwd1d_fused_lds_floats(64, hlen, k) * sizeof(float) <= 48 * 1024) best = k;
    return best;
}

template <int HLEN, int TF>
static hipError_t run_fwd(const Fwd1DFusedArgs& a, hipStream_t s) {
    const size_t lds = (size_t)fwd1d_fused_lds_floats(TF, HLEN, a.K) * sizeof(float);
    const int tiles = cdiv(a.N0 >> a.K, TF);
    const long long total = (long long)tiles * a.rows;
    hipLaunchKernelGGL((dwt1_fwd_fused_kernel<HLEN, TF, 256>), dim3((unsigned)(8 * ((total + 7) / 8))), dim3(256),
                       lds, s, a, tiles);
    return hipGetLastError();
}

template <int HLEN, int T0>
static hipError_t run_inv(const Inv1DFusedArgs& a, hipStream_t s) {
    const size_t lds = (size_t)inv1d_fused_lds_floats(T0, HLEN, a.K) * sizeof(float);
    const int tiles = cdiv(a.N0, T0);
    const long long total = (long long)tiles * a.rows;
    hipLaunchKernelGGL((dwt1_inv_fused_kernel<HLEN, T0, 256>), dim3((unsigned)(8 * ((total + 7) / 8))), dim3(256),
                       lds, s, a, tiles);
    return hipGetLastError();
}

// short rows: four row tiles per workgroup of 256 threads (dwt1_*_fused_rows_kernel)
static bool short_rows(int N0) {
    static const bool on = !(lab_env("PDWT_FUSED1D_ROWS") && atoi(lab_env("PDWT_FUSED1D_ROWS")) == 0);  // A/B measurements
    return on && N0 <= 512;
}
template <int HLEN>
static hipError_t run_fwd_rows(const Fwd1DFusedArgs& a, hipStream_t s) {
    constexpr int TF = 64, NT = 64, SUBS = 4;
    const int lds_floats = fwd1d_fused_lds_floats(TF, HLEN, a.K);
    const int tiles = cdiv(a.N0 >> a.K, TF);
    const long long total = (long long)tiles * a.rows;
    hipLaunchKernelGGL((dwt1_fwd_fused_rows_kernel<HLEN, TF, NT, SUBS>), dim3((unsigned)cdivll(total, SUBS)), dim3(NT * SUBS),
                       (size_t)SUBS * lds_floats * sizeof(float), s, a, tiles, lds_floats);
    return hipGetLastError();
}
template <int HLEN>
static hipError_t run_inv_rows(const Inv1DFusedArgs& a, hipStream_t s) {
    constexpr int T0 = 512, NT = 64, SUBS = 4;
    const int lds_floats = (inv1d_fused_lds_floats(T0, HLEN, a.K) + 3) & ~3;
    const int tiles = cdiv(a.N0, T0);
    const long long total = (long long)tiles * a.rows;
    hipLaunchKernelGGL((dwt1_inv_fused_rows_kernel<HLEN, T0, NT, SUBS>), dim3((unsigned)cdivll(total, SUBS)), dim3(NT * SUBS),
                       (size_t)SUBS * lds_floats * sizeof(float), s, a, tiles, lds_floats);
    return hipGetLastError();
}

// Short rows in batches of >= 2^16 samples: several rows per one-wavefront workgroup (1024 samples; fewer rows while the batch is too
// small for 2048 wavefronts), all levels out of LDS
// (dwt1_rows_kernels.hpp).  Forward+inverse us, one row per wavefront -> this (tools/ab_rows_tail.sh, profiles/r04zv_rows_tail.txt):
// 65536 x 64 haar L3 81 -> 33, db4 110 -> 51, 262144 x 64 haar 307 -> 90, 131072 x 32 db2 L2 142 -> 34; rows of 128 samples up to 8
// taps (32768 x 128 db2 L4 72 -> 44; sym8 level, db10 31 -> 38); rows of 256 samples with 2 taps only (16384 x 256 haar L5 54 -> 39;
// db4 L3 35 -> 51); rows of 512: behind everywhere.
// Smaller batches (2^16 .. 2^20 samples): 8192 x 64 haar L3 15.2 -> 9.3, 4096 x 64 db4 13.9 -> 10.5, 16384 x 32 haar L2 18.9 -> 8.3,
// 1024 x 64 level.  PDWT_ROWS_TAIL_ROW = longest such row (0 = never), PDWT_ROWS_TAIL_SAMPLES = samples per workgroup,
// PDWT_ROWS_TAIL_MIN_LOG2 = smallest batch (A/B measurements).
bool dwt1_rows_tail_applies(int rows, int N0, int K, int hlen) {
    static const int max_row = lab_env("PDWT_ROWS_TAIL_ROW") ? atoi(lab_env("PDWT_ROWS_TAIL_ROW")) : 256;
    static const bool forced = lab_env("PDWT_ROWS_TAIL_ROW") != nullptr;  // the knob set: every filter up to that row length
    static const int min_log2 = lab_env("PDWT_ROWS_TAIL_MIN_LOG2") ? atoi(lab_env("PDWT_ROWS_TAIL_MIN_LOG2")) : 16;  // smallest batch (log2 samples)
    if (N0 > max_row || N0 > kRowsTailSamples || K > kRowsTailMaxLevels || hlen > 20 || (long long)rows * N0 < (1LL << min_log2)) return false;
    return forced || N0 <= 64 || (N0 <= 128 && hlen <= 8) || (N0 <= 256 && hlen <= 2);
}
template <int HLEN>
static hipError_t run_rows_tail(RowsTailArgs& a, bool inverse, hipStream_t s) {
    static const int samples = lab_env("PDWT_ROWS_TAIL_SAMPLES") ? atoi(lab_env("PDWT_ROWS_TAIL_SAMPLES")) : kRowsTailSamples;
    constexpr int NT = 64;
    int G = (samples < kRowsTailSamples ? samples : kRowsTailSamples) / a.N0;
    while (G > 1 && a.rows / G < 2048) G >>= 1;  // smaller batches: fewer rows per wavefront rather than an idle chip
    a.G = G < 1 ? 1 : G;
    const size_t lds = rows_tail_lds_elems(a.G * a.N0) * sizeof(real_t);
    const unsigned grid = (unsigned)cdiv(a.rows, a.G);
    if (inverse) hipLaunchKernelGGL((dwt1_rows_tail_inv_kernel<HLEN, NT>), dim3(grid), dim3(NT), lds, s, a);
    else hipLaunchKernelGGL((dwt1_rows_tail_fwd_kernel<HLEN, NT>), dim3(grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}
static hipError_t launch_rows_tail(const float* in, float* const* det, float* out, int rows, int N0, int K, int hlen, bool inverse,
                                   const FilterBank& fb, hipStream_t s) {
    RowsTailArgs a;
    a.in = in; a.out = out; a.rows = rows; a.N0 = N0; a.K = K; a.G = 1; a.hlen = hlen; a.fb = fb;
    for (int k = 0; k < kRowsTailMaxLevels; k++) a.det[k] = k < K ? det[k] : nullptr;
    switch (hlen) {
#define X(h) case h: if constexpr (h <= 20) return run_rows_tail<h>(a, inverse, s); break;
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

// ONE predicate for planner (plan.cpp) and launchers: K consecutive levels starting from a row of N0
// samples can run fused iff hlen is even, 2^(K+2) divides N0 (every level length even, every band row
// 16-B aligned) and N0 < 2^30 (32-bit tile arithmetic)
// Rows of 2^(K+1) samples (and whole quads of input) since round 5: the forward stores its deepest level in pairs, the inverse stages it
// in pairs where its rows are not whole quads.  Until then 2^(K+2): a signal of 10^6 samples, five levels, ran FUSED1D[1-4] + a level
// launch (17.4 us forward+inverse against 12.4 for 2^20).  (`inverse` = the several-rows-per-wavefront kernels' rule: 2^(K+2).)
bool dwt1_fused_supported(int hlen, int N0, int K, bool strict) {
    return !(hlen & 1) && hlen >= 2 && hlen <= kMaxTaps && K >= 1 && K <= kMaxFusedLevels && (N0 & 3) == 0 &&
           (N0 % (1 << (K + (strict ? 2 : 1)))) == 0 && N0 < (1 << 30);
}

// levels: K >= 2 consecutive levels starting from `in` of length N0 per row
hipError_t launch_dwt1_fwd_fused(const float* in, float* const* det, float* app, int rows, int N0, int K, int hlen,
                                 const FilterBank& fb, hipStream_t s) {
    // (a single level only through the several-rows-per-wavefront kernels: the pyramids need two levels to pay)
    const bool rows_tail = dwt1_rows_tail_applies(rows, N0, K, hlen) && dwt1_fused_supported(hlen, N0, K, true);  // (those kernels: 2^(K+2))
    if ((K < 2 && !rows_tail) || !dwt1_fused_supported(hlen, N0, K, false)) return hipErrorNotSupported;
    Fwd1DFusedArgs a;
    a.in = in; a.app = app; a.rows = rows; a.N0 = N0; a.K = K;
    for (int k = 0; k < kMaxFusedLevels; k++) a.det[k] = k < K ? det[k] : nullptr;
    interleave(a.fb, fb);
    if (rows_tail) {
        const hipError_t e = launch_rows_tail(in, det, app, rows, N0, K, hlen, false, fb, s);
        if (e != hipErrorNotSupported) return e;
    }
    // TF = 64 final-level outputs per workgroup at K = 6 (TF = 128 measured 25 % slower on 2^24 sym8 L6: 55 KB of LDS leaves 2
    // workgroups per CU), i.e. a segment of 64 * 2^K = 4096 input samples.  With FEWER levels the segment of TF = 64 shrinks
    // to 512 samples at K = 3 -- two per thread, 32768 workgroups for 4096 rows of 4096: the forward took 55-65 us where the
    // inverse (4096 outputs per workgroup at any K) takes 22-27 (round 4, tools/cliffs.py on batched 1D).  TF grows as K
    // shrinks so that the segment stays at 4096 samples (filters of up to 20 taps; not beyond the row).
    if (short_rows(N0) && hlen <= 20) {
        switch (hlen) {
#define X(h) case h: if constexpr (h <= 20) return run_fwd_rows<h>(a, s); break;
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    static const bool wide = !(lab_env("PDWT_FUSED1D_WIDE") && atoi(lab_env("PDWT_FUSED1D_WIDE")) == 0);  // A/B measurements
    int TF = 64;
    if (wide && hlen <= 20 && K < 6) {
        TF = 64 << (6 - K);
        if (TF > 1024) TF = 1024;
        while (TF > 64 && TF / 2 >= (N0 >> K)) TF /= 2;
    }
    switch (hlen) {
#define X(h)                                                   \
    case h:                                                    \
        if constexpr (h <= 20) {                               \
            if (TF == 1024) return run_fwd<h, 1024>(a, s);     \
            if (TF == 512) return run_fwd<h, 512>(a, s);       \
            if (TF == 256) return run_fwd<h, 256>(a, s);       \
            if (TF == 128) return run_fwd<h, 128>(a, s);       \
        }                                                      \
        return run_fwd<h, 64>(a, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t launch_dwt1_inv_fused(const float* app, const float* const* det, float* out, int rows, int N0, int K,
                                 int hlen, const FilterBank& fb, hipStream_t s) {
    const bool rows_tail = dwt1_rows_tail_applies(rows, N0, K, hlen) && dwt1_fused_supported(hlen, N0, K, true);  // (those kernels: 2^(K+2))
    if ((K < 2 && !rows_tail) || !dwt1_fused_supported(hlen, N0, K, false)) return hipErrorNotSupported;
    Inv1DFusedArgs a;
    a.app = app; a.out = out; a.rows = rows; a.N0 = N0; a.K = K;
    for (int k = 0; k < kMaxFusedLevels; k++) a.det[k] = k < K ? det[k] : nullptr;
    interleave(a.fb, fb);
    if (rows_tail) {
        const hipError_t e = launch_rows_tail(app, const_cast<float* const*>(det), out, rows, N0, K, hlen, true, fb, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (short_rows(N0) && hlen <= 20) {
        switch (hlen) {
#define X(h) case h: if constexpr (h <= 20) return run_inv_rows<h>(a, s); break;
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    // 4096 output samples per workgroup (2048: 73 us, 4096: 58 us, 8192: 61 us on 2^24 sym8 L6)
    switch (hlen) {
#define X(h) case h: return run_inv<h, 4096>(a, s);
        PDWT_EVEN_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
