// launch_swt_split.hip -- launchers of the two-launch a-trous level (swt_split_kernels.hpp).
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#include "tuning.hpp"
#include "swt_kernels_args.hpp"
#include "swt_stream_kernels.hpp"
#ifndef PDWT_DOUBLE
#include "swt_split_kernels.hpp"
#include "swt_colstream_kernels.hpp"
#endif

namespace pdwt {

static int env_int(const char* name, int dflt) {
    const char* e = lab_env(name);
    return e ? atoi(e) : dflt;
}

// shortest filter whose column pass runs on the strip kernels of swt_colstream_kernels.hpp (fp32 library; tuning key "swt_colstream")
static std::atomic<int>& colstream_min() {
    static std::atomic<int> v{(int)tune::swt_colstream_taps};
    return v;
}
int set_swt_colstream_min(int taps) { return colstream_min().exchange(taps < 0 ? 0 : taps); }
int get_swt_colstream_min() { return colstream_min().load(std::memory_order_relaxed); }

// ---- the any-length stream kernels (swt_stream_kernels.hpp, round 5): both libraries ----------------------------------------------
static void stream_taps(SwtStreamArgs& k, const FilterBank& fb, int hlen) {
    for (int j = 0; j < kStreamTaps; ++j) k.tl[j] = k.th[j] = 0;
    for (int j = 0; j < hlen; ++j) {
        k.tl[kStreamPadL + j] = fb.lo[hlen - 1 - j];
        k.th[kStreamPadL + j] = fb.hi[hlen - 1 - j];
    }
}

template <bool SYN, bool ALONG_Y, int NC, int R, int NT>
static hipError_t go_stream(const SwtStreamArgs& k, hipStream_t s) {
    const long long waves = ALONG_Y ? stream_waves_y(k.problems, k.batch, k.Nr, k.Nc, k.f, R, NC)
                                    : stream_waves_x(k.problems, k.batch, k.Nr, k.Nc, k.f, R, NC);
    hipLaunchKernelGGL((swt_stream_kernel<SYN, ALONG_Y, NC, R, NT>), dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), 0, s, k);
    return hipGetLastError();
}
// outputs per work item (A/B builds: -DPDWT_STREAM_R_...): analysis | synthesis, one column | a pair of columns per work item
#ifndef PDWT_STREAM_R_A1
#define PDWT_STREAM_R_A1 4
#endif
#ifndef PDWT_STREAM_R_A2
#define PDWT_STREAM_R_A2 4
#endif
#ifndef PDWT_STREAM_R_S1
#define PDWT_STREAM_R_S1 4
#endif
#ifndef PDWT_STREAM_R_S2
#define PDWT_STREAM_R_S2 4
#endif
#ifndef PDWT_STREAM_NT_Y
#define PDWT_STREAM_NT_Y 512
#endif
// pairs of columns (16-B accesses) where rows are even and the planes 16-B aligned; along x the dilation must be even too
template <bool SYN, bool ALONG_Y>
static hipError_t run_stream(const SwtStreamArgs& k, bool pairs, hipStream_t s) {
    constexpr int NT = ALONG_Y ? PDWT_STREAM_NT_Y : 256;
    constexpr int R1 = SYN ? PDWT_STREAM_R_S1 : PDWT_STREAM_R_A1, R2 = SYN ? PDWT_STREAM_R_S2 : PDWT_STREAM_R_A2;
    if (pairs && (ALONG_Y || !(k.f & 1))) return go_stream<SYN, ALONG_Y, 2, R2, NT>(k, s);
    return go_stream<SYN, ALONG_Y, 1, R1, NT>(k, s);
}
// a pair of adjacent columns per work item: planes aligned to two elements
static bool al_pair(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (2 * sizeof(real_t) - 1)) == 0; }


// one 2D level: a row launch + a column launch through scratch (2 * Nr * Nc * batch elements)
static hipError_t stream_level2d(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    const long long plane = (long long)a.Nr * a.Nc;
    const bool pairs = !(a.Nc & 1) && !(a.bstride & 1) && al_pair(tmp) && al_pair(a.A) && al_pair(a.H) && al_pair(a.V) && al_pair(a.D) &&
                       al_pair(inverse ? (const void*)a.out : (const void*)a.in);
    SwtStreamArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = batch; k.hlen = a.hlen;
    k.scale = (real_t)0.5;
    stream_taps(k, a.fb, a.hlen);
    if (!inverse) {
        SwtStreamArgs r = k;  // in -> lo, hi (scratch: two planes per image)
        r.problems = 1;
        r.in[0][0] = a.in; r.in_bstride = a.bstride;
        r.out[0][0] = tmp; r.out[0][1] = tmp + plane; r.out_bstride = 2 * plane;
        hipError_t e = run_stream<false, false>(r, pairs, s);
        if (e != hipSuccess) return e;
        SwtStreamArgs c = k;  // lo -> A, H ; hi -> V, D
        c.problems = 2;
        c.in[0][0] = tmp; c.in[1][0] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0][0] = a.A; c.out[0][1] = a.H; c.out[1][0] = a.V; c.out[1][1] = a.D; c.out_bstride = a.bstride;
        return run_stream<false, true>(c, pairs, s);
    }
    SwtStreamArgs c = k;  // (A, H) -> L' ; (V, D) -> H' (scratch), the pending soft threshold applied to H, V, D on the way in
    c.problems = 2;
    c.in[0][0] = a.A; c.in[0][1] = a.H; c.in[1][0] = a.V; c.in[1][1] = a.D; c.in_bstride = a.bstride;
    c.soft[0][1] = c.soft[1][0] = c.soft[1][1] = a.soft_beta;
    c.out[0][0] = tmp; c.out[1][0] = tmp + plane; c.out_bstride = 2 * plane;
    hipError_t e = run_stream<true, true>(c, pairs, s);
    if (e != hipSuccess) return e;
    SwtStreamArgs r = k;  // (L', H') -> out
    r.problems = 1;
    r.in[0][0] = tmp; r.in[0][1] = tmp + plane; r.in_bstride = 2 * plane;
    r.out[0][0] = a.out; r.out_bstride = a.bstride;
    return run_stream<true, false>(r, pairs, s);
}


// one pass of the (batched) 1D transform: rows of Nc samples, or columns (along_y) of a plane
static hipError_t stream_pass1d(const SwtPassArgs& a, bool inverse, hipStream_t s) {
    if (a.f < 1 || a.f >= (a.along_y ? a.Nr : a.Nc) || a.hlen < 2 || a.hlen > kMaxTaps) return hipErrorNotSupported;
    const bool pairs = !(a.Nc & 1) && al_pair(a.in0) && al_pair(a.out0) && (inverse ? al_pair(a.in1) : al_pair(a.out1));
    SwtStreamArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = 1; k.hlen = a.hlen; k.problems = 1;
    k.scale = (real_t)0.5;
    stream_taps(k, a.fb, a.hlen);
    k.in[0][0] = a.in0; k.in[0][1] = a.in1; k.out[0][0] = a.out0; k.out[0][1] = a.out1;
    if (a.along_y) return inverse ? run_stream<true, true>(k, pairs, s) : run_stream<false, true>(k, pairs, s);
    return inverse ? run_stream<true, false>(k, pairs, s) : run_stream<false, false>(k, pairs, s);
}

#ifdef PDWT_DOUBLE
// ---- fp64 library: the two launches are ALWAYS the stream kernels ------------------------------------------------------------------
// shortest filter on this path (tuning keys "swt_split_fwd" / "swt_split_inv"; 0 = never, 100 + n = n taps at every size)
static std::atomic<int>& split_min(bool inverse) {
    static std::atomic<int> fwd{env_int("PDWT_SWT_SPLIT_FWD", 12)}, inv{env_int("PDWT_SWT_SPLIT_INV", 6)};
    return inverse ? inv : fwd;
}
int set_swt_split_min(int inverse, int taps) { return split_min(inverse != 0).exchange(taps < 0 ? 0 : taps); }
int get_swt_split_min(int inverse) { return split_min(inverse != 0).load(std::memory_order_relaxed); }

bool swt2_split_supported(int hlen, int Nr, int Nc, int f, bool inverse, long long samples) {
    const Tuning* at = active_tuning();
    int min_taps = at ? (inverse ? at->swt_split_inv : at->swt_split_fwd) : split_min(inverse).load(std::memory_order_relaxed);
    if (min_taps <= 0) return false;
    if (min_taps >= 100) min_taps -= 100;
    (void)samples;
    if (hlen < 2 || hlen > kMaxTaps || hlen < min_taps) return false;
    return f >= 1 && f < Nr && f < Nc;
}

// scratch: 2 * Nr * Nc * batch elements
hipError_t launch_swt2_split(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    if (!swt2_split_supported(a.hlen, a.Nr, a.Nc, a.f, inverse, (long long)batch * a.Nr * a.Nc) || !tmp) return hipErrorNotSupported;
    note_family("stream");
    return stream_level2d(a, tmp, inverse, batch, s);
}

hipError_t try_launch_swt1_split(const SwtPassArgs& a, bool inverse, hipStream_t s) {
    const Tuning* at = active_tuning();  // the same thresholds as the 2D level
    int min_taps = at ? (inverse ? at->swt_split_inv : at->swt_split_fwd) : split_min(inverse).load(std::memory_order_relaxed);
    if (min_taps >= 100) min_taps -= 100;
    if (min_taps <= 0 || a.hlen < min_taps) return hipErrorNotSupported;
    return stream_pass1d(a, inverse, s);
}
#else

// filter lengths the split kernels are built for
#define PDWT_SPLIT_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)

static inline v2f mk2h(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}

// Where the two launches beat the LDS-tiled level kernel (2048^2 levels, tools/swtsweep.py, profiles/r03_swt_split_sweep.txt):
// the inverse from 10 taps on (10 taps 46 -> 41 us per level, 12 taps 80 -> 43, 16 taps 93 -> 40, 40 taps 150-290 -> 57-60),
// the forward from 18 taps on (18 taps 47-61 -> 40, 26 taps 91-111 -> 42, 40 taps 127-158 -> 45-47; at 16 taps the
// tiled kernel's 35-45 is level with the 36-39 of two launches).
// Round 5 re-measured the forward inside whole plans (tools/swt_fwd_split_ab.py, profiles/r05k_swt_fwd_split_ab.txt): from 2 M samples
// on the two launches win from 14 taps -- sym8 2048^2 L3 forward 160 -> 102 us, L5 262 -> 173, 4096^2 L2 397 -> 320, 1080 x 1920 L3
// 94 -> 80; db7 2048^2 L3 145 -> 100 -- at 1024^2 they lose (54 -> 61) and 12 taps are level (2048^2 107 -> 99, 4096^2 291 -> 300):
// the default is 14 taps from 2^21 samples, 18 below.  Tuning keys "swt_split_fwd" / "swt_split_inv" (environment
// PDWT_SWT_SPLIT_FWD / _INV): the shortest filter that takes this path at full size, 0 = never, 100 + n = n taps at EVERY size.
static std::atomic<int>& split_min(bool inverse) {
    static std::atomic<int> fwd{env_int("PDWT_SWT_SPLIT_FWD", (int)tune::swt_split_fwd_big_taps)}, inv{env_int("PDWT_SWT_SPLIT_INV", (int)tune::swt_split_inv_taps)};
    return inverse ? inv : fwd;
}
int set_swt_split_min(int inverse, int taps) { return split_min(inverse != 0).exchange(taps < 0 ? 0 : taps); }
int get_swt_split_min(int inverse) { return split_min(inverse != 0).load(std::memory_order_relaxed); }

// Where the any-length stream kernels (swt_stream_kernels.hpp: one or two columns per work item, plain FMAs) serve the fp32 library
// (round 5; tools/swt_stream32_ab.py, profiles/r05g_swt_stream32_*.txt, three levels forward | inverse, same box).
// (a) Rows that are not whole quads: the packed kernels below need Nc % 4 == 0 and the tiles' 16-B accesses at 4-B alignment pay
//     per tap -- 2047^2 db10 L2 282 | 375 us on the tiles, 138 | 178 here; 1022^2 db10 L3 156 | 162 -> 64 | 60; 2046^2 db20 L2
//     338 | 471 -> 143 | 232.  Shorter filters: the inverse gains from 10 taps (db5 0.94, db6 0.88, sym8 0.85 of the tiles' time),
//     the forward is level at 16 taps and behind below (db7 1.08-1.11, db5 1.54).
// (b) Small images with long filters: four columns per lane leave a 512^2 level 256 wavefronts of 43 dependent loads each; two
//     columns per lane are four times the wavefronts.  From 18 taps up to 2^20 samples: 512^2 L3 db9 51 | 45 -> 33 | 38, db10
//     54 | 48 -> 34 | 37, db20 73 | 99 -> 41 | 51; 1024^2 db10 67 | 76 -> 62 | 60, db20 90 | 111 -> 79 | 88; 16 taps: level
//     (0.83-1.08 | 1.01-1.10; the inverse of 10-16 taps alone: 1.03-1.35 of the TILES' time, r05g_swt_stream32_inv.txt -- its 1024^2 rows,
//     0.85-0.88, compared with the packed kernels the old rule picked at exactly 2^20 samples, not with the tiles).  From 1.9 M samples on the packed kernels are ahead again (1200 x 1600: 1.06-1.11 | 1.17-1.27;
//     2048^2: 1.4-1.5 | 1.5-1.7).
// Shortest filter per rule, 0 = never (A/B measurements in the lab library: PDWT_SWT_STREAM_RAGGED_FWD / _RAGGED / _SMALL / _SMALL_LOG2).
static bool stream_route32(int hlen, int Nr, int Nc, int f, bool inverse, long long samples, int min_taps) {
    static const int ragged_fwd = env_int("PDWT_SWT_STREAM_RAGGED_FWD", 16), ragged_inv = env_int("PDWT_SWT_STREAM_RAGGED", 10),
                     small_fwd = env_int("PDWT_SWT_STREAM_SMALL", 18), small_inv = env_int("PDWT_SWT_STREAM_SMALL_INV", 18),
                     small_log2 = env_int("PDWT_SWT_STREAM_SMALL_LOG2", 20);
    if (min_taps <= 0 || min_taps >= 100) return false;  // the path is off, or the packed kernels are forced (tests)
    if (hlen < 2 || hlen > kMaxTaps || f < 1 || f >= Nr || f >= Nc) return false;
    const int ragged = inverse ? ragged_inv : ragged_fwd;
    if ((Nc & 3) && ragged > 0 && hlen >= ragged) return true;
    const int small = inverse ? small_inv : small_fwd;
    if (samples <= (1LL << small_log2) && small > 0 && hlen >= small) return true;
    return false;
}

bool swt2_split_supported(int hlen, int Nr, int Nc, int f, bool inverse, long long samples) {
    const Tuning* at = active_tuning();  // the calling plan's snapshot, else the process-wide value
    int min_taps = at ? (inverse ? at->swt_split_inv : at->swt_split_fwd) : split_min(inverse).load(std::memory_order_relaxed);
    if (min_taps <= 0) return false;
    if (stream_route32(hlen, Nr, Nc, f, inverse, samples, min_taps)) return true;
    // (narrow images keep this path too: measured, the tiled inverse is slower still there -- 4096 images of 64^2, db4 L2
    // forward+inverse 1006 us on this path against 1168 tiled, 8192 of 32^2 1047 against 2182)
    // (rows the dilation does not divide used to come here from 10 taps on, when the alternative was three direct passes; since
    // round 5 the LDS tiles take any row count, so such levels follow the ordinary thresholds -- and the caller's tuning value)
    if (min_taps >= 100) {
        min_taps -= 100;  // forced: the same threshold at every size (tests)
    } else if (!inverse) {
        if (samples < (1LL << tune::swt_split_fwd_big_log2) && min_taps < tune::swt_split_fwd_taps) min_taps = (int)tune::swt_split_fwd_taps;  // 14 and 16 taps: from 2 M samples (see above)
    } else {
        // small launches: two launches of one round each cost more than they save until the filter is long -- inverse levels of
        // 256^2 / 512^2: 10 taps 13 | 17-18 us (tiled | split), 16 taps 16 | 21-22, 26 taps 33 | 26-27, 40 taps 31-44 | 22-35;
        // 1024^2: 12 taps 24 | 23, 16 taps 28 | 25, 20 taps 35 | 28 (profiles/r03_swt_split_sweep.txt)
        // (8 taps at dilation 1 and 2 -- the rule below -- already from 1.5 M samples on: a 1080 x 1920 image, db4 L3 forward+inverse
        // 121 -> 105 us, 1200 x 1600 121 -> 102; at 1024^2 and below the tiles stay ahead: 79 against 90 us)
        const bool eight_mid = hlen == 8 && f <= 2 && min_taps <= 10 && samples >= (3LL << 19);
        // (up to AND INCLUDING 2^20 samples since round 5: a 1024^2 image is exactly that, and its 12-16-tap inverse is 46-51 us on
        // the tiles against 66-73 on these kernels -- tools/swt_pitch_probe.py, profiles/r05k_swt_inv_at_2p20.txt)
        if (samples <= (1LL << tune::swt_split_inv_small_log2) && min_taps < tune::swt_split_inv_small_taps) min_taps = (int)tune::swt_split_inv_small_taps;
        else if (samples < (1LL << tune::swt_split_inv_mid_log2) && min_taps < tune::swt_split_inv_mid_taps && !eight_mid) min_taps = (int)tune::swt_split_inv_mid_taps;
        // 8 taps (db4, sym4, bior2.4 ...), dilation 1 and 2, from 2048^2 on: the tiled inverse issues one 16-B load per band and
        // tap at 4-B / 8-B alignment there (52-55 us per 2048^2 level against 33 at dilation 4, where the loads are aligned);
        // the two launches stage aligned quads through LDS: 35.6 / 33.4 us, 4096^2 190.8 / 199.2 -> 140.8 / 164.6
        // (profiles/r04m_swt_short_sweep2.txt).  6 taps: level (2048^2) or behind (4096^2, 1024^2); 4 taps: behind.
        else if (hlen == 8 && f <= 2 && min_taps <= 10) min_taps = 8;
    }
    if ((hlen & 1) || hlen < 4 || hlen > kMaxTaps || hlen < min_taps) return false;
    if ((Nc & 3) || f < 1 || f >= Nr || f >= Nc || Nc < 16) return false;
    if (f != 1 && f != 2 && (f & 3)) return false;
    return true;
}

// ---- the column pass streamed down strips with the filter's history in LDS (swt_colstream_kernels.hpp)

template <int HLEN, bool INV>
static hipError_t run_colstream(const SwtSplitArgs& c, hipStream_t s) {
    if constexpr (HLEN < 10) {
        return hipErrorNotSupported;
    } else {
        // Rules (tools/swt_colstream_ab.py, profiles/r06_swt_colstream.txt): the inverse gains at every size and dilation (2048^2: 7-25 %,
        // 4096^2: 25-45 %); the forward only where the register kernels' shared rows fall out of L1 / L2 -- from 2^23 samples per launch
        // (4096^2: 7-36 %; 2048^2: +-5 %).  Tuning key "swt_colstream": shortest filter, 0 = never, 100 + n = n taps, both directions, every size.
        const Tuning* at = active_tuning();
        int min_taps = at ? at->swt_colstream : get_swt_colstream_min();
        const bool forced = min_taps >= 100;
        if (forced) min_taps -= 100;
        if (min_taps <= 0 || HLEN < min_taps) return hipErrorNotSupported;
        if (!INV && !forced && (long long)c.batch * c.Nr * c.Nc < (1LL << tune::swt_colstream_fwd_log2)) return hipErrorNotSupported;
        constexpr int TXC = 64, TY = 32, NT = 256, M = 8, MINB = 2;
        using G = SwtColStreamGeom<HLEN, INV, TXC, TY>;
        SwtColStreamArgs a;
        for (int k = 0; k < 4; ++k) { a.in[k] = c.in[k]; a.out[k] = c.out[k]; }
        a.Nr = c.Nr; a.Nc = c.Nc; a.f = c.f;
        a.in_bstride = c.in_bstride; a.out_bstride = c.out_bstride;
        a.soft_beta = c.soft_beta;
        a.t = c.t;
        a.wk = swt_walk(c.Nr, c.Nc, c.f, 4);
        // 32-bit byte offsets inside a plane (buffer stores), chains the staging plan can advance with one conditional wrap
        if ((c.Nc & 3) || a.wk.rows_phase < TY || c.batch > 65535 || (long long)c.Nr * c.Nc * 2 * (long long)sizeof(real_t) >= (1LL << 32)) return hipErrorNotSupported;
        a.strips = cdiv(c.Nc, TXC);
        // segments as long as one round of resident workgroups allows (strip_walk_seg; a segment re-loads hlen - 1 rows, it filters none twice)
        const long long units = (long long)a.strips * a.wk.phases * c.batch;
        static std::atomic<bool> big[64] = {};
        constexpr size_t lds = (size_t)G::LDS_REALS * sizeof(real_t);
        auto kern = swt_colstream_kernel<HLEN, INV, TXC, TY, NT, M, MINB>;
        hipError_t e = allow_big_lds(kern, lds, big);
        if (e != hipSuccess) return e;
        static std::atomic<int> slots_cache{0};
        static const int forced_slots = lab_env("PDWT_STRIP_SLOTS") ? atoi(lab_env("PDWT_STRIP_SLOTS")) : 0;  // A/B measurements
        a.seg = strip_walk_seg(a.wk.rows_phase, units, TY, 1, forced_slots > 0 ? forced_slots : resident_slots(kern, NT, lds, &slots_cache));
        a.segs = cdiv(a.wk.rows_phase, a.seg);
        hipLaunchKernelGGL(kern, dim3(8 * cdiv(a.strips * a.segs * a.wk.phases, 8), c.batch), dim3(NT), lds, s, a);
        return hipGetLastError();
    }
}

template <int NT, typename K>
static hipError_t go(K kernel, const SwtSplitArgs& a, long long waves, hipStream_t s, size_t lds_bytes = 0) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), lds_bytes, s, a);
    return hipGetLastError();
}

template <int HLEN>
static hipError_t run_split(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    // column workgroups of 16 wavefronts x 4 rows share their input rows on one CU (split_col_work).  8 rows per wavefront
    // with 8 wavefronts (the same 64 rows, half the L1 traffic) measured no better: 16 taps 31.1 / 21.5 us against 30.5 / 20.0
    constexpr int NT = 256, R = 4, RC = 4, NTC = PDWT_SPLIT_NTC;  // NTC: wavefronts of a column workgroup share their rows (split_col_work)
    const long long plane = (long long)a.Nr * a.Nc;
    SwtSplitArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = batch;
    k.soft_beta = a.soft_beta;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2h(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    const int f = a.f;
    const long long col_items = split_col_waves(batch, a.Nr, a.Nc, f, RC);
    const long long row_items4 = f >= 4 ? split_row_waves(batch, a.Nr, split_row_items4(a.Nc, f, R)) : 0;
    const long long row_lds = split_row_lds_waves(batch, a.Nr, a.Nc);
    hipError_t e;
    if (!inverse) {
        SwtSplitArgs r = k;  // in -> lo, hi (scratch: two planes per image)
        r.in[0] = a.in; r.in_bstride = a.bstride;
        r.out[0] = tmp; r.out[1] = tmp + plane; r.out_bstride = 2 * plane;
        // dilation 1, 2, 4: staged through LDS (coalesced loads and stores); beyond, the runs of f / 4 lanes are whole lines
        if (f == 1) e = go<NT>(swt_row_fwd_lds_kernel<HLEN, 1, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 1>(false, NT));
        else if (f == 2) e = go<NT>(swt_row_fwd_lds_kernel<HLEN, 2, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 2>(false, NT));
        else if (f == 4) e = go<NT>(swt_row_fwd_lds_kernel<HLEN, 4, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 4>(false, NT));
        else e = go<NT>(swt_row_fwd4_kernel<HLEN, R, NT>, r, row_items4, s);
        if (e != hipSuccess) return e;
        SwtSplitArgs c = k;
        c.in[0] = tmp; c.in[1] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0] = a.A; c.out[1] = a.H; c.out[2] = a.V; c.out[3] = a.D; c.out_bstride = a.bstride;
        e = run_colstream<HLEN, false>(c, s);
        if (e != hipErrorNotSupported) { note_family("colstream"); return e; }
        return go<NTC>(swt_col_fwd_kernel<HLEN, RC, NTC>, c, col_items, s);
    }
    SwtSplitArgs c = k;  // A, H, V, D -> interleaved (L', H') (scratch: two planes per image)
    c.in[0] = a.A; c.in[1] = a.H; c.in[2] = a.V; c.in[3] = a.D; c.in_bstride = a.bstride;
    c.out[0] = tmp; c.out_bstride = 2 * plane;
    e = run_colstream<HLEN, true>(c, s);
    if (e != hipErrorNotSupported) note_family("colstream");
    else e = go<NTC>(swt_col_inv_kernel<HLEN, RC, NTC>, c, col_items, s);
    if (e != hipSuccess) return e;
    SwtSplitArgs r = k;
    r.in[0] = tmp; r.in_bstride = 2 * plane;
    r.out[0] = a.out; r.out_bstride = a.bstride;
    if (f == 1) return go<NT>(swt_row_inv_lds_kernel<HLEN, 1, 1, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 1>(true, NT));
    if (f == 2) return go<NT>(swt_row_inv_lds_kernel<HLEN, 2, 1, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 2>(true, NT));
    if (f == 4) return go<NT>(swt_row_inv_lds_kernel<HLEN, 4, 1, NT>, r, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 4>(true, NT));
    return go<NT>(swt_row_inv4_kernel<HLEN, R, NT>, r, row_items4, s);
}

// The row kernels are the (batched) 1D transform as they stand: rows of Nc samples, approximation and detail planes
// separate (the inverse interleaves them while staging, or packs over column pairs at dilation >= 8).
template <int HLEN>
static hipError_t run_split1(const SwtPassArgs& a, bool inverse, hipStream_t s) {
    constexpr int NT = 256, R = 4;
    SwtSplitArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = 1;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2h(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    const int f = a.f;
    const long long row_items4 = f >= 4 ? split_row_waves(1, a.Nr, split_row_items4(a.Nc, f, R)) : 0;
    const long long row_lds = split_row_lds_waves(1, a.Nr, a.Nc);
    k.in[0] = a.in0; k.in[1] = a.in1; k.out[0] = a.out0; k.out[1] = a.out1;
    if (!inverse) {
        if (f == 1) return go<NT>(swt_row_fwd_lds_kernel<HLEN, 1, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 1>(false, NT));
        if (f == 2) return go<NT>(swt_row_fwd_lds_kernel<HLEN, 2, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 2>(false, NT));
        if (f == 4) return go<NT>(swt_row_fwd_lds_kernel<HLEN, 4, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 4>(false, NT));
        return go<NT>(swt_row_fwd4_kernel<HLEN, R, NT>, k, row_items4, s);
    }
    if (f == 1) return go<NT>(swt_row_inv_lds_kernel<HLEN, 1, 2, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 1, 2>(true, NT));
    if (f == 2) return go<NT>(swt_row_inv_lds_kernel<HLEN, 2, 2, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 2, 2>(true, NT));
    if (f == 4) return go<NT>(swt_row_inv_lds_kernel<HLEN, 4, 2, NT>, k, row_lds, s, sizeof(real_t) * swt_row_lds_floats<HLEN, 4, 2>(true, NT));
    return go<NT>(swt_row_inv4p_kernel<HLEN, R, NT>, k, row_items4, s);
}

hipError_t try_launch_swt1_split(const SwtPassArgs& a, bool inverse, hipStream_t s) {
    // shortest filter on this path, 0 = never (A/B measurements): forward from 10 taps, inverse from PDWT_SWT1_SPLIT_INV
    static const int min_fwd = env_int("PDWT_SWT1_SPLIT", 10), min_inv = env_int("PDWT_SWT1_SPLIT_INV", 10);
    int min_taps = inverse ? min_inv : min_fwd;
    // the (batched) 1D inverse of 6 and 8 taps from 2^22 samples on: 2^24 samples x 4 levels db4 208 -> 167 us, db3 176 -> 162;
    // 4096 rows of 4096: db4 252 -> 156, db3 219 -> 155; at 2^20 samples the vec kernels stay ahead (21 against 29 us); 4 taps:
    // behind on one long row (129 -> 147)
    if (inverse && min_taps == 10 && a.hlen >= 6 && (long long)a.Nr * a.Nc >= (1LL << 22)) min_taps = 6;
    // rows of 64 samples: a wavefront of these kernels stages 1024 outputs of ONE row -- the four-samples-per-work-item kernels
    // stay ahead there (65536 rows of 64, three levels forward+inverse: db3 221 -> 125 us, db4 238 -> 156, db5 228 -> 148); from
    // 256 samples on the row kernels win (profiles/r04zl_swt1_short_rows.txt)
    if (a.Nc < 128) return hipErrorNotSupported;
    // rows that are not whole quads: the stream kernels (one or two samples per work item)
    // (rows are filtered along x: the row COUNT does not bound the dilation -- one long row is Nr = 1)
    // (1D, three levels forward | inverse, a row of 2^20 + 1 samples: sym8 44.8 | 46.1 -> 26.6 | 32.3 us, db10 52.5 | 53.5 -> 28.1 | 36.5, db20
    // 86 | 88 -> 39 | 56; 10 taps: level | 7 % behind -- from 12 taps)
    if (!a.along_y && (a.Nc & 3) && min_taps > 0 && a.hlen >= 12 && stream_route32(a.hlen, 1 << 30, a.Nc, a.f, inverse, (long long)a.Nr * a.Nc, 10))
        return stream_pass1d(a, inverse, s);
    if (a.along_y || min_taps <= 0 || (a.hlen & 1) || a.hlen < 4 || a.hlen < min_taps || a.hlen > kMaxTaps) return hipErrorNotSupported;
    if ((a.Nc & 3) || a.Nc < 16 || a.f < 1 || a.f >= a.Nc || (a.f != 1 && a.f != 2 && (a.f & 3))) return hipErrorNotSupported;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(a.in0) || !al16(a.out0) || (inverse ? !al16(a.in1) : !al16(a.out1))) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_split1<h>(a, inverse, s);
        PDWT_SPLIT_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

// scratch: 2 * Nr * Nc * batch elements, 16-B aligned
hipError_t launch_swt2_split(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    if (!swt2_split_supported(a.hlen, a.Nr, a.Nc, a.f, inverse, (long long)batch * a.Nr * a.Nc) || !tmp) return hipErrorNotSupported;
    {
        const Tuning* at = active_tuning();
        const int min_taps = at ? (inverse ? at->swt_split_inv : at->swt_split_fwd) : split_min(inverse).load(std::memory_order_relaxed);
        if (stream_route32(a.hlen, a.Nr, a.Nc, a.f, inverse, (long long)batch * a.Nr * a.Nc, min_taps)) {
            note_family("stream");
            return stream_level2d(a, tmp, inverse, batch, s);
        }
    }
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(tmp) || !al16(a.A) || !al16(a.H) || !al16(a.V) || !al16(a.D) || (a.bstride & 3)) return hipErrorNotSupported;
    if (!al16(inverse ? (const void*)a.out : (const void*)a.in)) return hipErrorNotSupported;
    note_family("packed");
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_split<h>(a, tmp, inverse, batch, s);
        PDWT_SPLIT_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}
#endif

}  // namespace pdwt
