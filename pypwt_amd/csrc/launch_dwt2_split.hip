// launch_dwt2_split.hip -- launchers of the two-launch DECIMATED level (dwt2_split_kernels.hpp): long filters.
//
// EXPERIMENT (round 4), compiled into libpypwt_amd_lab.so only (-DPDWT_LAB_KERNELS); in the product the entry points
// decline.  Built because the review of round 3 asked for the register-blocked two-launch scheme of the SWT on the
// decimated transform; measured on MI355X it does NOT beat LDS tiles of the right shape (launch_dwt2_fast.hip):
//   db20 (40 taps)  2048^2 L5  forward 92 (old tiles) | 91 (split) | 53 us (32 x 32 / 32 x 16 tiles), inverse 120 | 82 | 62
//                   4096^2 level 1: forward 87.4 | 90.7 | 56.9, inverse 132 | 92 | 90
//   db13 (26 taps)  4096^2 level 1: forward 58 | 62 | 38.6, inverse 75 | 80 | 46
// (event-timed launches; profiles/r04d_dwt_split_sweep_first.txt, r04i_tilesweep.txt, r04j_tilesweep2.txt).  Why: a level of
// a decimated transform is a quarter of the SWT's planes, so each of the two launches is ONE round of at most a wavefront
// or four per SIMD whose load, arithmetic and store phases coincide (rocprofv3: 14.4 + 12.3 us for the two forward launches
// of a 2048^2 level of 40 taps, wavefronts resident ~60 % of the time, VALU 25 % busy), and every small level pays two
// launch latencies (2 x 5-6 us) where one tile launch now takes 5-6.  Kept with its emulation and GPU parity tests
// (tests/test_emu_tiles.py::test_emu_dwt_split_*, tests/test_gpu_parity.py::test_dwt_two_launch_levels).
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#if !defined(PDWT_DOUBLE) && defined(PDWT_LAB_KERNELS)
#include "dwt2_split_kernels.hpp"
#endif
#ifdef PDWT_DOUBLE
#include "dwt2_stream_kernels.hpp"
#endif

namespace pdwt {

static int env_int_d(const char* name, int dflt) {
    const char* e = lab_env(name);
    return e ? atoi(e) : dflt;
}

// Shortest (even) filter whose decimated 2D levels run as a row launch + a column launch through scratch instead of one
// LDS-tiled launch.  Tuning keys "dwt_split_fwd" / "dwt_split_inv" (environment PDWT_DWT_SPLIT_FWD / _INV): 0 = never
// (default), 100 + n = n taps at EVERY size (tests).  The product keeps the value and does nothing with it.
#ifdef PDWT_DOUBLE
constexpr int kDsplitDefault = 28;  // fp64 library: the any-length stream kernels (dwt2_stream_kernels.hpp), see below
#else
constexpr int kDsplitDefault = 0;
#endif
static std::atomic<int>& dsplit_min(bool inverse) {
    static std::atomic<int> fwd{env_int_d("PDWT_DWT_SPLIT_FWD", kDsplitDefault)}, inv{env_int_d("PDWT_DWT_SPLIT_INV", kDsplitDefault)};
    return inverse ? inv : fwd;
}
int set_dwt_split_min(int inverse, int taps) { return dsplit_min(inverse != 0).exchange(taps < 0 ? 0 : taps); }
int get_dwt_split_min(int inverse) { return dsplit_min(inverse != 0).load(std::memory_order_relaxed); }

#if defined(PDWT_DOUBLE)
// ---- fp64 library (round 5): a row launch + a column launch of the any-length stream kernels.  Over doubles the LDS tiles of a
// 22-40-tap level are 32 x 32 outputs behind 38 halo rows and columns.  Measured (tools/f64dwt_ab.py, profiles/r05i_f64_dwt_stream_ab.txt,
// three levels, tiles | stream, us): the INVERSE gains from 28 taps on large levels -- db20 4096^2 694 | 274, 2048^2 189 | 91, 1024^2
// 62 | 50; db16 4096^2 518 | 255, 2048^2 138 | 84, 1024^2 46 | 49; db14 4096^2 275 | 258, 2048^2 80 | 84; 26 taps and below: tiles
// (db13 4096^2 165 | 237) -- the FORWARD does not (db20 4096^2 261 | 348: its row launch reads the image one double per lane at a
// 64-B lane stride).  Tuning keys as in the fp32 lab library: 0 = never, n = from n taps where the rule below allows, 100 + n = n taps
// at every size and in both directions' own key (tests).
bool dwt2_split_supported(int hlen, int Nr, int Nc, bool inverse, long long samples) {
    const Tuning* at = active_tuning();
    int min_taps = at ? (inverse ? at->dwt_split_inv : at->dwt_split_fwd) : dsplit_min(inverse).load(std::memory_order_relaxed);
    if (min_taps <= 0) return false;
    if (min_taps >= 100) {
        min_taps -= 100;
    } else {
        if (!inverse) return false;
        const int need = samples >= (1LL << 24) ? 28 : (samples >= (1LL << 22) ? 30 : (samples >= (1LL << 20) ? 36 : 99));
        if (min_taps < need) min_taps = need;
    }
    if ((hlen & 1) || hlen < 2 || hlen > kMaxTaps || hlen < min_taps) return false;
    return !(Nr & 1) && !(Nc & 1) && Nr >= 2 && Nc >= 2;
}

static void dstream_taps(DwtStreamArgs& k, const FilterBank& fb, int hlen, bool syn) {
    for (int j = 0; j < kStreamTaps; ++j) k.t[0][j] = k.t[1][j] = k.t[2][j] = k.t[3][j] = 0;
    if (!syn) {
        for (int j = 0; j < hlen; ++j) {
            k.t[0][kStreamPadL + j] = fb.lo[hlen - 1 - j];
            k.t[1][kStreamPadL + j] = fb.hi[hlen - 1 - j];
        }
        return;
    }
    for (int j = 0; j < hlen / 2; ++j)
        for (int par = 0; par < 2; ++par) {
            k.t[par][kStreamPadL + j] = fb.lo[hlen - 1 - (2 * j + 1 - par)];
            k.t[2 + par][kStreamPadL + j] = fb.hi[hlen - 1 - (2 * j + 1 - par)];
        }
}

template <bool SYN, bool ALONG_Y, int NC>
static hipError_t go_dstream(const DwtStreamArgs& k, hipStream_t s) {
    constexpr int R = 4, NT = ALONG_Y ? 512 : 256;
    const int positions = SYN ? (ALONG_Y ? k.in_rows : k.in_cols) : (ALONG_Y ? k.out_rows : k.out_cols);
    const long long waves = dwt_stream_waves(ALONG_Y, k.problems, k.batch, k.in_rows, k.in_cols, positions, R, NC);
    hipLaunchKernelGGL((dwt_stream_kernel<SYN, ALONG_Y, NC, R, NT>), dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), 0, s, k);
    return hipGetLastError();
}
template <bool SYN>
static hipError_t go_dstream_y(const DwtStreamArgs& k, bool pairs, hipStream_t s) {
    return pairs ? go_dstream<SYN, true, 2>(k, s) : go_dstream<SYN, true, 1>(k, s);
}
static bool al_pair_d(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (2 * sizeof(real_t) - 1)) == 0; }

// scratch: Nr * Nc * batch elements
hipError_t launch_dwt2_split_fwd(const Fwd2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    if (!tmp || !dwt2_split_supported(a.hlen, a.Nr, a.Nc, false, (long long)batch * a.Nr * a.Nc)) return hipErrorNotSupported;
    if (a.Nr2 * 2 != a.Nr || a.Nc2 * 2 != a.Nc) return hipErrorNotSupported;
    const long long plane = (long long)a.Nr * a.Nc2;
    DwtStreamArgs k{};
    k.batch = batch; k.hlen = a.hlen;
    dstream_taps(k, a.fb, a.hlen, false);
    DwtStreamArgs r = k;  // rows: in (Nr x Nc) -> lo, hi (Nr x Nc / 2) in scratch
    r.problems = 1;
    r.in_rows = a.Nr; r.in_cols = a.Nc; r.out_rows = a.Nr; r.out_cols = a.Nc2;
    r.in[0][0] = a.in; r.in_bstride = a.in_bstride;
    r.out[0][0] = tmp; r.out[0][1] = tmp + plane; r.out_bstride = 2 * plane;
    hipError_t e = go_dstream<false, false, 1>(r, s);
    if (e != hipSuccess) return e;
    DwtStreamArgs c = k;  // columns: lo -> A, H ; hi -> V, D (Nr / 2 x Nc / 2)
    c.problems = 2;
    c.in_rows = a.Nr; c.in_cols = a.Nc2; c.out_rows = a.Nr2; c.out_cols = a.Nc2;
    c.in[0][0] = tmp; c.in[1][0] = tmp + plane; c.in_bstride = 2 * plane;
    c.out[0][0] = a.A; c.out[0][1] = a.H; c.out[1][0] = a.V; c.out[1][1] = a.D; c.out_bstride = a.out_bstride;
    const bool pairs = !(a.Nc2 & 1) && !(a.out_bstride & 1) && al_pair_d(tmp) && al_pair_d(a.A) && al_pair_d(a.H) && al_pair_d(a.V) && al_pair_d(a.D);
    return go_dstream_y<false>(c, pairs, s);
}

hipError_t launch_dwt2_split_inv(const Inv2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    if (!tmp || !dwt2_split_supported(a.hlen, a.Nr, a.Nc, true, (long long)batch * a.Nr * a.Nc)) return hipErrorNotSupported;
    if (a.Nrc * 2 != a.Nr || a.Ncc * 2 != a.Nc) return hipErrorNotSupported;
    const long long plane = (long long)a.Nr * a.Ncc;
    DwtStreamArgs k{};
    k.batch = batch; k.hlen = a.hlen;
    dstream_taps(k, a.fb, a.hlen, true);
    DwtStreamArgs c = k;  // columns: (A, H) -> L' ; (V, D) -> H' (Nr x Nc / 2) in scratch
    c.problems = 2;
    c.in_rows = a.Nrc; c.in_cols = a.Ncc; c.out_rows = a.Nr; c.out_cols = a.Ncc;
    c.in[0][0] = a.A; c.in[0][1] = a.H; c.in[1][0] = a.V; c.in[1][1] = a.D; c.in_bstride = a.in_bstride;
    c.out[0][0] = tmp; c.out[1][0] = tmp + plane; c.out_bstride = 2 * plane;
    const bool pairs = !(a.Ncc & 1) && !(a.in_bstride & 1) && al_pair_d(tmp) && al_pair_d(a.A) && al_pair_d(a.H) && al_pair_d(a.V) && al_pair_d(a.D);
    hipError_t e = go_dstream_y<true>(c, pairs, s);
    if (e != hipSuccess) return e;
    DwtStreamArgs r = k;  // rows: (L', H') -> out (Nr x Nc)
    r.problems = 1;
    r.in_rows = a.Nr; r.in_cols = a.Ncc; r.out_rows = a.Nr; r.out_cols = a.Nc;
    r.in[0][0] = tmp; r.in[0][1] = tmp + plane; r.in_bstride = 2 * plane;
    r.out[0][0] = a.out; r.out_bstride = a.out_bstride;
    return go_dstream<true, false, 1>(r, s);
}
#elif !defined(PDWT_LAB_KERNELS)
bool dwt2_split_supported(int, int, int, bool, long long) { return false; }
hipError_t launch_dwt2_split_fwd(const Fwd2DArgs&, real_t*, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_dwt2_split_inv(const Inv2DArgs&, real_t*, int, hipStream_t) { return hipErrorNotSupported; }
#else

// filter lengths the split kernels are built for
#define PDWT_DSPLIT_HLENS(X) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)

static inline v2f mk2d(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}

// (Nr, Nc): the level's image side (forward: its input, inverse: its output); samples: over the whole batch
bool dwt2_split_supported(int hlen, int Nr, int Nc, bool inverse, long long samples) {
    const Tuning* at = active_tuning();
    int min_taps = at ? (inverse ? at->dwt_split_inv : at->dwt_split_fwd) : dsplit_min(inverse).load(std::memory_order_relaxed);
    if (min_taps <= 0) return false;
    if (min_taps >= 100) {
        min_taps -= 100;  // forced: the same threshold at every size (tests)
    } else if (samples > (1LL << 24) && hlen < 26) {
        // beyond one cache-resident 4096^2 image the scratch planes go through HBM: twice the bytes of the fused level
        return false;
    }
    if ((hlen & 1) || hlen < 10 || hlen > kMaxTaps || hlen < min_taps) return false;
    if ((Nr & 1) || (Nc & 7) || Nr < 2 || Nc < 16) return false;
    if (samples >= (1LL << 33)) return false;  // the kernels index their wavefronts with 32 bits
    return true;
}

template <int NT, typename K>
static hipError_t god(K kernel, const DwtSplitArgs& a, long long waves, hipStream_t s, size_t lds_bytes = 0) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), lds_bytes, s, a);
    return hipGetLastError();
}

// rows per work item of the column kernels: blocks of 8 while that leaves about a wavefront per SIMD, else 4 / 2 (small
// levels are latency chains: more, shorter wavefronts)
static int pick_r(long long out_rows, long long cols, int batch, int rmax) {
    const long long groups = (long long)batch * ((cols / 4 + 63) / 64);
    if (rmax >= 8 && groups * ((out_rows + 7) / 8) >= 1024) return 8;
    if (groups * ((out_rows + 3) / 4) >= 512) return 4;
    return 2;
}

template <int HLEN>
static hipError_t run_dsplit_fwd(const Fwd2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    constexpr int NT = 256, NTC = 512;
    const long long plane = (long long)a.Nr * a.Nc2;
    DwtSplitArgs k{};
    k.batch = batch;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2d(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    DwtSplitArgs r = k;  // in -> lo, hi (scratch: two planes [Nr][Nc2] per image)
    r.rows = a.Nr; r.cols = a.Nc;
    r.in[0] = a.in; r.in_bstride = a.in_bstride;
    r.out[0] = tmp; r.out[1] = tmp + plane; r.out_bstride = 2 * plane;
    // (workgroups of 64 / 256 / 512 / 1024 threads measured the same: profiles/r04h_dsplit_nt.txt)
    hipError_t e = god<NT>(dwt_row_fwd_kernel<HLEN, NT>, r, dwt_row_waves(batch, a.Nr, a.Nc), s,
                           sizeof(real_t) * dwt_row_lds_floats<HLEN>(false, NT));
    if (e != hipSuccess) return e;
    DwtSplitArgs c = k;
    c.rows = a.Nr; c.cols = a.Nc2;
    c.in[0] = tmp; c.in[1] = tmp + plane; c.in_bstride = 2 * plane;
    c.out[0] = a.A; c.out[1] = a.H; c.out[2] = a.V; c.out[3] = a.D; c.out_bstride = a.out_bstride;
    // 8 rows x 32+ taps: hipcc gives up unrolling the 2048+ multiply-adds of the body (the tap tests become run-time branches)
    constexpr int RBIG = HLEN <= 30 ? 8 : 4;
    // (2 / 4 / 8 rows per chunk of loads measured the same: profiles/r04f_dsplit_ab_db20.txt)
    switch (pick_r(a.Nr2, a.Nc2, batch, RBIG)) {
        case 8: return god<NTC>(dwt_col_fwd_kernel<HLEN, RBIG, NTC, 4>, c, dwt_col_waves(batch, a.Nr2, a.Nc2, RBIG), s);
        case 4: return god<NTC>(dwt_col_fwd_kernel<HLEN, 4, NTC, 4>, c, dwt_col_waves(batch, a.Nr2, a.Nc2, 4), s);
        default: return god<NT>(dwt_col_fwd_kernel<HLEN, 2, NT, 4>, c, dwt_col_waves(batch, a.Nr2, a.Nc2, 2), s);
    }
}

template <int HLEN>
static hipError_t run_dsplit_inv(const Inv2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    constexpr int NT = 256, NTC = 512;
    const long long plane2 = (long long)a.Nr * a.Nc;  // interleaved (t1, t2): [Nr][Ncc][2] per image
    DwtSplitArgs k{};
    k.batch = batch;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2d(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    DwtSplitArgs c = k;
    c.rows = a.Nrc; c.cols = a.Ncc;
    c.in[0] = a.A; c.in[1] = a.H; c.in[2] = a.V; c.in[3] = a.D; c.in_bstride = a.in_bstride;
    c.out[0] = tmp; c.out_bstride = plane2;
    hipError_t e;
    switch (pick_r(a.Nr, a.Ncc, batch, 8)) {
        case 8: e = god<NTC>(dwt_col_inv_kernel<HLEN, 8, NTC, 4>, c, dwt_col_waves(batch, a.Nr, a.Ncc, 8), s); break;
        case 4: e = god<NTC>(dwt_col_inv_kernel<HLEN, 4, NTC, 4>, c, dwt_col_waves(batch, a.Nr, a.Ncc, 4), s); break;
        default: e = god<NT>(dwt_col_inv_kernel<HLEN, 2, NT, 4>, c, dwt_col_waves(batch, a.Nr, a.Ncc, 2), s); break;
    }
    if (e != hipSuccess) return e;
    DwtSplitArgs r = k;
    r.rows = a.Nr; r.cols = a.Ncc;
    r.in[0] = tmp; r.in_bstride = plane2;
    r.out[0] = a.out; r.out_bstride = a.out_bstride;
    return god<NT>(dwt_row_inv_kernel<HLEN, NT>, r, dwt_row_waves(batch, a.Nr, a.Nc), s, sizeof(real_t) * dwt_row_lds_floats<HLEN>(true, NT));
}

static bool al16d(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scratch: Nr * Nc * batch elements, 16-B aligned
hipError_t launch_dwt2_split_fwd(const Fwd2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    if (!tmp || !dwt2_split_supported(a.hlen, a.Nr, a.Nc, false, (long long)batch * a.Nr * a.Nc)) return hipErrorNotSupported;
    if (a.Nr2 * 2 != a.Nr || a.Nc2 * 2 != a.Nc) return hipErrorNotSupported;
    if (!al16d(tmp) || !al16d(a.in) || !al16d(a.A) || !al16d(a.H) || !al16d(a.V) || !al16d(a.D) || (a.in_bstride & 3) || (a.out_bstride & 3))
        return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_dsplit_fwd<h>(a, tmp, batch, s);
        PDWT_DSPLIT_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t launch_dwt2_split_inv(const Inv2DArgs& a, real_t* tmp, int batch, hipStream_t s) {
    if (!tmp || !dwt2_split_supported(a.hlen, a.Nr, a.Nc, true, (long long)batch * a.Nr * a.Nc)) return hipErrorNotSupported;
    if (a.Nrc * 2 != a.Nr || a.Ncc * 2 != a.Nc) return hipErrorNotSupported;
    if (!al16d(tmp) || !al16d(a.out) || !al16d(a.A) || !al16d(a.H) || !al16d(a.V) || !al16d(a.D) || (a.in_bstride & 3) || (a.out_bstride & 3))
        return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_dsplit_inv<h>(a, tmp, batch, s);
        PDWT_DSPLIT_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}
#endif

}  // namespace pdwt
