// launch_dwt2.hip -- instantiations + launchers of the fused 2D DWT level kernels (gfx950).
#include <stdlib.h>

#include <atomic>

#include "dwt2_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"
#include "tuning.hpp"

namespace pdwt {

// Tile shape: 64 columns x TY rows of each sub-band per 256-thread workgroup (4 wavefronts,
// one 64-lane wavefront per output row).  Short filters use TY = 16 (about 40 KB of LDS ->
// 4 workgroups per CU); long filters amortise their (hlen-2)-row halo over TY = 32.
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd(const Fwd2DArgs& a, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    const size_t lds = (size_t)fwd2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(dwt2_fwd_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)fwd2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    dim3 grid(cdiv(a.Nc2, TX), cdiv(a.Nr2, TY), batch);
    hipLaunchKernelGGL((dwt2_fwd_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv(const Inv2DArgs& a, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    const size_t lds = (size_t)inv2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(dwt2_inv_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)inv2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    dim3 grid(cdiv(a.Nc, 2 * TX), cdiv(a.Nr, 2 * TY), batch);
    hipLaunchKernelGGL((dwt2_inv_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

#ifdef PDWT_DOUBLE
// fp64 build: of the tuned kernels only the register kernels of dwt2_wave_kernels.hpp and dwt1_reg_kernels.hpp are
// compiled (they are written over real_t); the LDS-tiled packed-fp32 kernels (tuned single-level, tile pyramid, streaming strips, fused
// 1D pyramids, fused SWT groups) are not: those levels run through the generic kernels of this file / launch_dwt1.hip
bool dwt2_wave2_supported(int, int, int) { return false; }
hipError_t launch_dwt2_fwd_wave2(const real_t*, real_t* const[3], real_t* const[4], int, int, int, const FilterBank&, int,
                                 hipStream_t, int) { return hipErrorNotSupported; }
bool dwt2_pyramid_supported(int, int, int, bool) { return false; }
bool dwt2_strip_supported(int, int, int) { return false; }
hipError_t launch_dwt2_fwd_pyr2(const real_t*, real_t* const[3], real_t* const[4], int, int, int, const FilterBank&, int,
                                hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt2_fwd_strip2(const real_t*, real_t* const[3], real_t* const[4], int, int, int, const FilterBank&,
                                  int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt2_inv_strip2(const real_t* const[4], const real_t* const[3], real_t*, int, int, int,
                                  const FilterBank&, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt2_inv_pyr2(const real_t* const[4], const real_t* const[3], real_t*, int, int, int,
                                const FilterBank&, int, hipStream_t) { return hipErrorNotSupported; }
// the register-ring kernels for 12-20 taps keep hlen tap pairs in SGPRs and hlen/2 x 8 running sums in VGPRs: in doubles neither fits
// (hipcc: 256 VGPRs + 300-900 B of scratch per lane), so the fp64 library does not build them
hipError_t try_launch_dwt2_fwd_ring(const Fwd2DArgs&, int, hipStream_t, int, int) { return hipErrorNotSupported; }
hipError_t try_launch_dwt2_inv_ring(const Inv2DArgs&, int, hipStream_t, int, int) { return hipErrorNotSupported; }
int dwt1_fused_max_levels(int) { return 1; }
bool dwt1_fused_supported(int, int, int, bool) { return false; }
hipError_t launch_dwt1_fwd_fused(const real_t*, real_t* const*, real_t*, int, int, int, int, const FilterBank&,
                                 hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt1_inv_fused(const real_t*, const real_t* const*, real_t*, int, int, int, int, const FilterBank&,
                                 hipStream_t) { return hipErrorNotSupported; }
constexpr int kTyLong = 8;   // long filters: a 32-row tile of doubles would not fit the 160 KB of LDS
#else
constexpr int kTyLong = 32;
#endif

#if defined(PDWT_DOUBLE) || !defined(PDWT_LAB_KERNELS)
// levels chained inside one launch (launch_dwt2_chain.hip): an experiment that measured no faster, LAB build only
bool dwt2_chain_supported(int, int, int, int) { return false; }
int dwt2_chain_tiles(int, int, int) { return 0; }
int set_chain_timeout(int) { return 0; }
hipError_t launch_dwt2_fwd_chain(const real_t*, real_t* const*, real_t* const*, int, int, int, int, const FilterBank&, int, unsigned*,
                                 unsigned, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt2_inv_chain(real_t*, real_t* const*, real_t* const*, int, int, int, int, const FilterBank&, int, unsigned*,
                                 unsigned, hipStream_t) { return hipErrorNotSupported; }
#endif

// The wave-per-tile kernels take the levels that are large enough to be bandwidth-bound (>= 2^22 samples
// enter the level: 2048^2 of one image); smaller levels are launch-bound and stay with the LDS tiles
// (numbers in launch_dwt2_wave.hip).  PDWT_NO_WAVE=1 (read once) keeps the LDS tiles everywhere, PDWT_WAVE_MIN /
// pdwt_set_tuning("wave_min_log2") override the threshold (log2 samples): tests and A/B measurements.
// fp64 build: the alternative is the generic kernel, not a tuned LDS tile, so the wave kernels start at 2^16 samples
#ifdef PDWT_DOUBLE
constexpr int kWaveMinDefault = (int)tune::wave_min_log2_f64;
#else
constexpr int kWaveMinDefault = (int)tune::wave_min_log2;
#endif
static std::atomic<int>& wave_min_log2() {
    static std::atomic<int> v{lab_env("PDWT_NO_WAVE") ? 63 : (lab_env("PDWT_WAVE_MIN") ? atoi(lab_env("PDWT_WAVE_MIN")) : kWaveMinDefault)};
    return v;
}
int set_wave_min_log2(int value) {  // pdwt_set_tuning("wave_min_log2")
    if (value < 0) value = 0;
    if (value > 63) value = 63;
    return wave_min_log2().exchange(value);
}
int get_wave_min_log2() { return wave_min_log2().load(std::memory_order_relaxed); }
static std::atomic<int>& wave2_flag() {
    static std::atomic<int> v{lab_env("PDWT_WAVE2") ? 1 : 0};
    return v;
}
static std::atomic<int>& swt_fused_flag() {
    static std::atomic<int> v{lab_env("PDWT_SWT_FUSED") ? atoi(lab_env("PDWT_SWT_FUSED")) : 1};
    return v;
}
int set_swt_fused_enabled(int value) { return swt_fused_flag().exchange(value < 0 ? 0 : (value > 2 ? 2 : value)); }
int get_swt_fused_enabled() { return swt_fused_flag().load(std::memory_order_relaxed); }
static std::atomic<int>& reg1d_flag() {
    static std::atomic<int> v{lab_env("PDWT_REG1D") ? (atoi(lab_env("PDWT_REG1D")) & 15) : 3};
    return v;
}
int set_reg1d_enabled(int value) { return reg1d_flag().exchange(value < 0 ? 0 : (value > 15 ? 15 : value)); }  // four flag bits
int get_reg1d_enabled() { return reg1d_flag().load(std::memory_order_relaxed); }
static std::atomic<int>& chain_flag() {
    static std::atomic<int> v{lab_env("PDWT_CHAIN") ? atoi(lab_env("PDWT_CHAIN")) : 0};  // opt-in: measured no faster, see plan.cpp
    return v;
}
int set_chain_enabled(int value) { return chain_flag().exchange(value < 0 ? 0 : (value > 3 ? 3 : value)); }
int get_chain_enabled() { return chain_flag().load(std::memory_order_relaxed); }
int set_wave2_enabled(int value) { return wave2_flag().exchange(value ? 1 : 0); }
int get_wave2_enabled() { return wave2_flag().load(std::memory_order_relaxed); }
static thread_local const char* g_last_family = "";
void note_family(const char* family) { g_last_family = family; }
const char* last_family() { return g_last_family; }
// a launcher's verdict: hipErrorNotSupported = it declined (try the next one); anything else = it launched (or failed)
static bool took(hipError_t e, const char* family) {
    if (e == hipErrorNotSupported) return false;
    note_family(family);
    return true;
}
static thread_local const Tuning* g_active_tuning = nullptr;
void set_active_tuning(const Tuning* t) { g_active_tuning = t; }
const Tuning* active_tuning() { return g_active_tuning; }
// Register-ring kernels (dwt2_ring_kernels.hpp): 38 % fewer vector instructions than the LDS tile at 16 taps (rocprofv3
// SQ_INSTS_VALU per 4096^2 level: 11.1e6 -> 6.9e6, profiles/r05a_*), but a level is only as fast as its wavefronts are many:
// same-box A/B inside plans (tools/ring_ab.py, profiles/r05c_ring_ab.txt; forward + inverse of the whole plan, tiles -> ring):
//   * batches (>= 2^25 samples in the level), 16 taps: 2 / 4 / 16 x 4096^2 L4 186 -> 175 / 345 -> 316 / 1350 -> 1151 us,
//     4 x 3000 x 4000 247 -> 217, 64 x 1024^2 333 -> 317, 16 x 2048^2 332 -> 310; 12 taps 4 x 4096^2 310 -> 295;
//   * ONE 4096^2 image: level 1 alone 56.1 -> 53.3 us, the four-level plan 87.3 -> 88.8 (2048 wavefronts of 46 rows each: the
//     launch ends with every wavefront's last rows at once); 14 / 18 / 20 taps lose 4-13 % in plans at every batch size
//     (18 and 20 keep the tile's inverse, launch_dwt2_ring.hip); images of 512 columns lose 2x (two strips per image).
// Default: levels of at least 2^25 samples, 12 or 16 taps, rows of at least 1024 columns.  Tuning key "ring_min_log2"
// (63 = never; below the default: every level of 10-20 taps and any width of that size on -- tests and measurements).
constexpr int kRingMinDefault = (int)tune::ring_min_log2;
static std::atomic<int>& ring_min_log2() {
    static std::atomic<int> v{kRingMinDefault};
    return v;
}
int set_ring_min_log2(int value) { return ring_min_log2().exchange(value < 0 ? 0 : (value > 63 ? 63 : value)); }
int get_ring_min_log2() { return ring_min_log2().load(std::memory_order_relaxed); }
static bool ring_kernels_for(long long samples, int hlen, int Nc, bool inverse) {
    const int m = g_active_tuning ? g_active_tuning->ring_min_log2 : ring_min_log2().load(std::memory_order_relaxed);
    if (m >= 63 || samples < (1LL << m)) return false;
    static const int dirs = lab_env("PDWT_RING_DIRS") ? atoi(lab_env("PDWT_RING_DIRS")) : 3;  // A/B measurements: bit 0 forward, bit 1 inverse
    if (!((dirs >> (inverse ? 1 : 0)) & 1)) return false;
    if (m < kRingMinDefault) return hlen >= 10 && hlen <= 20;  // forced: every length the kernels are built for, any width
    return (hlen == 12 || hlen == 16) && Nc >= tune::ring_min_columns;
}

// Strip-streaming kernels for long filters (dwt2_long_kernels.hpp, round 6).  A long filter is arithmetic-bound: the tiles
// re-filter their halo (40 taps, 32 x 8 inverse tile: 1.75x the column arithmetic, 6.3x the staging) where a strip walks down
// the level and filters every row once.  Inside plans (tools/long_ab.py, profiles/r06_long_ab.txt; three levels, forward /
// inverse us, tiles -> strips):
//     db20  4096^2        92 / 113 ->  71 /  73      4 images   349 / 469 -> 246 / 258     16 images  1254 / 2052 -> 954 / 930
//     db16  4096^2        65 /  92 ->  57 /  64                 272 / 373 -> 203 / 227
//     db13  4096^2        51 /  63 ->  53 /  56 (forced)        204 / 252 -> 185 / 201
//     db10  4096^2        45 /  59 ->  49 /  51 (forced)        186 / 243 -> 162 / 184                 717 /  985 -> 639 / 720
//     sym8  (16 taps)     41 /  45 ->  44 /  47 (forced)        155 / 157 -> 158 / 168: tiles and ring kernels stay
//     db20  2048^2 L5     38 /  45 ->  53 /  56 with every level forced; level 1 of the inverse alone: 45 -> 40 (db13: 28 -> 30)
//     db9   (18 taps)     43 /  47 ->  45 /  49 (forced)        172 / 193 -> 159 / 176
//     db14  4096^2        61 /  76 ->  56 /  57;  db15  62 / 68 -> 55 / 59;  db13 forward 51 -> 53: the forward of one image from 28 taps
//     db20  2048^2 L5     38 /  45 ->  53 /  56 with every level forced; level 1 of the inverse alone: 45 -> 40 (db16 38 -> 35, db15 31 -> 32)
// Rules (tuning keys "long_fwd" / "long_inv": the shortest filter, default 18; 0 = never; 100 + n = n taps at every size the
// kernels take: tests): 18 taps from 2^26 samples per launch; 20 taps and more: the inverse from 2^24 samples, the forward
// from 2^25; the forward of 28 taps and more from 2^24; the inverse of 32 taps and more from 2^22.
constexpr int kLongFwdDefault = (int)tune::long_min_taps, kLongInvDefault = (int)tune::long_min_taps;
static std::atomic<int>& long_min_taps(bool inverse) {
    static std::atomic<int> v[2] = {{kLongFwdDefault}, {kLongInvDefault}};
    return v[inverse ? 1 : 0];
}
int set_long_min_taps(int inverse, int taps) { return long_min_taps(inverse != 0).exchange(taps < 0 ? 0 : taps); }
int get_long_min_taps(int inverse) { return long_min_taps(inverse != 0).load(std::memory_order_relaxed); }
static bool long_kernels_for(long long samples, int hlen, bool inverse) {
    const int m = g_active_tuning ? (inverse ? g_active_tuning->long_inv : g_active_tuning->long_fwd) : get_long_min_taps(inverse);
    if (m <= 0) return false;
    if (m >= 100) return hlen >= m - 100;  // forced: every size
    if (hlen < m) return false;
    if (sizeof(real_t) == 8) {  // the fp64 library (profiles/r06_f64_long_ab.txt): the lengths whose tap tables fit the register file
        if (inverse) return hlen <= tune::long_f64_inv_max_taps && samples >= (1LL << tune::long_f64_min_log2);
        return (hlen <= tune::long_f64_fwd_max_taps && samples >= (1LL << tune::long_f64_fwd_log2)) ||
               (hlen <= tune::long_f64_fwd_mid_max_taps && samples >= (1LL << tune::long_f64_min_log2));
    }
    if (samples >= (1LL << tune::long_any_log2)) return true;
    if (hlen < tune::long_taps20) return false;
    if (samples >= (1LL << (inverse ? tune::long_inv_log2 : tune::long_fwd_log2))) return true;
    if (inverse) return hlen >= tune::long_inv_mid_taps && samples >= (1LL << tune::long_inv_mid_log2);
    return hlen >= tune::long_fwd_mid_taps && samples >= (1LL << tune::long_fwd_mid_log2);
}

static int eff_wave_min_log2() { return g_active_tuning ? g_active_tuning->wave_min_log2 : wave_min_log2().load(std::memory_order_relaxed); }
static bool wave_kernels_for(long long samples) {
    const int m = eff_wave_min_log2();
    return m < 63 && samples >= (1LL << m);
}

// fp32: the large levels of ONE cache-resident image (2^22 < samples <= 2^24) run on the LDS tiles: with branch-free staging
// they are ahead of the wave kernels there -- 4096^2 db4: 21.0 / 21.8 us forward / inverse against 21.7 / 22.6,
// 2048^2: 7.0 / 8.1 against 7.8 / 8.8 (profiles/r02y_kbench_tiles.txt) -- while a batch streamed from HBM keeps the
// wave kernels' forward (4 x 4096^2: 99 against 107 us).  "lds_max_log2" (PDWT_LDS_MAX) moves the limit (0 = wave kernels
// wherever they apply, as before).
// Round 4 re-measured the range between one image and the strips (2^24 < samples < 2^26: two or three 4096^2 images, 8-12
// of 2048^2, 32-48 of 1024^2; tools/ab_batchrange.sh, profiles/r04_ab_batchrange.txt).  Up to 2^25 samples the tiles win in both
// directions (forward+inverse db4 2 x 4096^2 173 -> 170 us, 8 x 2048^2 170 -> 161, 32 x 1024^2 165 -> 152; haar 2 x 4096^2
// 151 -> 139): the default limit is 25.  Above, the wave INVERSE stays ahead (3-8 %), and the wave FORWARD only on images of
// 2^24 samples (3 x 4096^2: 110 against 117 us; 12 x 2048^2: 114 against 101, 48 x 1024^2: 116 against 97 on the tiles).
#ifdef PDWT_DOUBLE
constexpr int kLdsMaxDefault = 0;   // no tuned LDS tiles in the fp64 build
#else
constexpr int kLdsMaxDefault = (int)tune::lds_max_log2;
#endif
static std::atomic<int>& lds_max_log2() {
    static std::atomic<int> v{lab_env("PDWT_LDS_MAX") ? atoi(lab_env("PDWT_LDS_MAX")) : kLdsMaxDefault};
    return v;
}
int set_lds_max_log2(int value) { return lds_max_log2().exchange(value < 0 ? 0 : (value > 62 ? 62 : value)); }
int get_lds_max_log2() { return lds_max_log2().load(std::memory_order_relaxed); }
static bool lds_tiles_for(long long samples, int hlen, long long per_image = 0, bool inverse = true) {
    // ... and only above 2^22 samples: in the step the 2048^2 level of cfg2 is 0.3-0.4 us faster on the wave kernels
    // (10.6 / 11.6 against 11.0 / 11.9 us event-timed), the 4096^2 level 1.0 / 0.3 us faster on the tiles
    const int m = g_active_tuning ? g_active_tuning->lds_max_log2 : lds_max_log2().load(std::memory_order_relaxed);
    if (m <= 0 || (hlen & 1) || samples <= (1LL << tune::lds_tiles_above_log2)) return false;
    if (samples <= (1LL << m)) return true;
    // forward levels of smaller images, up to where the strips take over
    return !inverse && m >= kLdsMaxDefault && kLdsMaxDefault > 0 && samples < (1LL << tune::lds_fwd_batch_log2) && per_image < (1LL << tune::lds_fwd_image_log2);
}

// A wavefront of the wave kernels owns a strip of 256 image columns: on a batch of NARROW images most of its lanes idle and
// every wavefront pays the hlen - 2 warm-up rows for a short walk -- 256 images of 256^2, level 2 (128 columns, 2^22 samples
// in total): 39.8 us forward / 53.8 us inverse for 32 MiB of traffic, against ~8 us on the LDS tiles (rocprofv3,
// tools/planprof.sh; the whole db4 L3 plan 155.7 -> see profiles/r04zc_small_batches.txt).  fp32 only (the fp64 library has
// no tuned tile to fall back to); a threshold forced below its default -- tests -- still takes the wave kernels.
static bool narrow_for_wave(int Nc) {
    static const int min_nc = lab_env("PDWT_WAVE_MIN_NC") ? atoi(lab_env("PDWT_WAVE_MIN_NC")) : (int)tune::wave_min_columns;  // two full strips; A/B measurements
    return sizeof(real_t) == 4 && Nc < min_nc && eff_wave_min_log2() >= kWaveMinDefault;
}

Tuning current_tuning() {
    Tuning t;
    t.wave_min_log2 = get_wave_min_log2();
    t.lds_max_log2 = get_lds_max_log2();
    t.swt_split_fwd = get_swt_split_min(0);
    t.swt_split_inv = get_swt_split_min(1);
    t.dwt_split_fwd = get_dwt_split_min(0);
    t.dwt_split_inv = get_dwt_split_min(1);
    t.ring_min_log2 = get_ring_min_log2();
    t.long_fwd = get_long_min_taps(0);
    t.long_inv = get_long_min_taps(1);
    t.swt_colstream = get_swt_colstream_min();
    t.swt_fwdstream = get_swt_fwdstream_min();
    t.swt_invstream = get_swt_invstream_min();
    t.wave2 = get_wave2_enabled();
    t.swt_fused = get_swt_fused_enabled();
    t.chain = get_chain_enabled();
    t.reg1d = get_reg1d_enabled();
    return t;
}

hipError_t launch_dwt2_fwd(const Fwd2DArgs& a, int batch, hipStream_t s) {
    if (long_kernels_for((long long)batch * a.Nr * a.Nc, a.hlen, false)) {
        const hipError_t e = try_launch_dwt2_fwd_long(a, batch, s);
        if (took(e, "long")) return e;
    }
    if (ring_kernels_for((long long)batch * a.Nr * a.Nc, a.hlen, a.Nc, false)) {
        const hipError_t e = try_launch_dwt2_fwd_ring(a, batch, s);
        if (took(e, "ring")) return e;
    }
    if (lds_tiles_for((long long)batch * a.Nr * a.Nc, a.hlen, (long long)a.Nr * a.Nc, false)) {
        const hipError_t e = try_launch_dwt2_fwd_fast(a, batch, s);
        if (took(e, "tile")) return e;
    }
    // (fp32, 4 taps: hipcc schedules the wave forward kernel's row loop with an s_waitcnt vmcnt(1) behind its stores, i.e. one
    // store round trip per four rows -- db2 2048^2 22.6 us against 9.8 us on the LDS tiles, found in round 4 by the reference's
    // own benchmark plan dwt2 db2 2048^2 L9: 44.7 us; the other lengths are level with the tiles there)
    // (a threshold forced below its default -- tests -- still takes it)
    const bool slow4 = sizeof(real_t) == 4 && a.hlen == 4 && eff_wave_min_log2() >= kWaveMinDefault;
    if (wave_kernels_for((long long)batch * a.Nr * a.Nc) && !slow4 && !narrow_for_wave(a.Nc)) {
        const hipError_t e = try_launch_dwt2_fwd_wave(a, batch, s);
        if (took(e, "wave")) return e;
    }
    {
        const hipError_t e = try_launch_dwt2_fwd_fast(a, batch, s);
        if (took(e, "tile")) return e;
    }
    note_family("generic");
    if (a.hlen & 1) return run_fwd<0, 64, kTyLong < 16 ? kTyLong : 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h)                                                       \
    case h:                                                        \
        if constexpr (h <= 12) return run_fwd<h, 64, 16, 256>(a, batch, s);  \
        else return run_fwd<h, 64, kTyLong, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_fwd<0, 64, kTyLong < 16 ? kTyLong : 16, 256>(a, batch, s);
    }
}

hipError_t launch_dwt2_inv(const Inv2DArgs& a, int batch, hipStream_t s) {
    // inverse: the wave kernel wins while the working set sits in the 256 MiB Infinity Cache (4096^2: 22.5 vs
    // 23.6 us) and loses slightly to the LDS tiles on a batch streamed from HBM (8 x 4096^2: 224-232 vs 219 us,
    // profiles/r02b_wbench_b8.txt): 2^26 samples and beyond go to the tiles
    const long long samples = (long long)batch * a.Nr * a.Nc;
    if (long_kernels_for(samples, a.hlen, true)) {
        const hipError_t e = try_launch_dwt2_inv_long(a, batch, s);
        if (took(e, "long")) return e;
    }
    if (ring_kernels_for(samples, a.hlen, a.Nc, true)) {
        const hipError_t e = try_launch_dwt2_inv_ring(a, batch, s);
        if (took(e, "ring")) return e;
    }
    if (lds_tiles_for(samples, a.hlen)) {
        const hipError_t e = try_launch_dwt2_inv_fast(a, batch, s);
        if (took(e, "tile")) return e;
    }
#ifdef PDWT_DOUBLE
    constexpr long long kInvWaveMax = 1LL << 62;  // no tuned LDS tile to hand a large batch to
#else
    constexpr long long kInvWaveMax = 1LL << tune::inv_wave_max_log2;
#endif
    if (wave_kernels_for(samples) && samples < kInvWaveMax && !narrow_for_wave(a.Nc)) {
        const hipError_t e = try_launch_dwt2_inv_wave(a, batch, s);
        if (took(e, "wave")) return e;
    }
    {
        const hipError_t e = try_launch_dwt2_inv_fast(a, batch, s);
        if (took(e, "tile")) return e;
    }
    note_family("generic");
    if (a.hlen & 1) return run_inv<0, 64, kTyLong < 16 ? kTyLong : 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h)                                                       \
    case h:                                                        \
        if constexpr (h <= 12) return run_inv<h, 64, 16, 256>(a, batch, s);  \
        else return run_inv<h, 64, kTyLong, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_inv<0, 64, kTyLong < 16 ? kTyLong : 16, 256>(a, batch, s);
    }
}

}  // namespace pdwt
