// launch_swt_fused.hip -- launchers of the multi-level 2D SWT kernels for 2-tap filter banks (swt2_fused_kernels.hpp).
#include "launch.hpp"
#include "launch_util.hpp"
#include "swt2_fused_kernels.hpp"
#include "swt2_fused4_kernels.hpp"

#include <cstdlib>

namespace pdwt {

// Levels l0 .. l0+K-1 of an (Nr, Nc) plane in one launch: 2-tap filters, rows of at least one strip, first dilation 1 or 8
// (levels 1.. or 4..: the lane shifts of a group must stay inside a wavefront), at least one unrolled group of rows per
// chain of rows, planes of at most 1 GiB (the dropped-store offsets of swt2_fused_kernels.hpp start at 2^30).  Rows of
// whole 16-B groups with a row count the first dilation divides run the aligned instantiations, every other size the
// GEN ones (SwtWalk: 4-B aligned 16-B accesses, chains of rows instead of phases).
// 4-tap banks (swt2_fused4_kernels.hpp): pairs of levels (1, 2) and (3, 4); `inverse` selects the direction asked about.
static bool swt4_inverse_built() { return true; }
bool swt2_fused_supported(int hlen, int Nr, int Nc, int l0, int K, bool inverse) {
    if (hlen == 4) {
        if (K != 2 || (l0 != 1 && l0 != 3) || (inverse && !swt4_inverse_built())) return false;
        const int f0 = 1 << (l0 - 1);
        return Nc >= 256 && (long long)Nr * Nc * (long long)sizeof(real_t) <= (1LL << 30) && Nr / f0 >= 8;
    }
    if (hlen != 2 || K < 2 || K > kSwtFusedMaxLevels || (l0 != 1 && l0 != 4)) return false;
    const int f0 = 1 << (l0 - 1);
    return Nc >= 256 && (long long)Nr * Nc * (long long)sizeof(real_t) <= (1LL << 30) && Nr / f0 >= (1 << K);
}

// Phase rows per wavefront.  A segment reads 2^K - 1 rows it does not own (the inverse: of all 3 K + 1 planes), so
// long segments move fewer bytes; but a wavefront is alone on its SIMD (up to 286 VGPRs) and a row costs it about a
// microsecond of latency, so the launch needs at least ~768 wavefronts.  Measured on one 2048^2 image, levels 1-3
// (9 strips): 8 / 16 / 24 / 32 / 64 rows = 66 / 52 / 48 / 51 / 68 us inverse, 40 / 42 / 37 / 39 / 55 us forward
// (profiles/r02w_*).  Hence: the longest segment that still gives 768 wavefronts.  PDWT_SWT_SEG overrides (tuning).
static int fused_seg_rows(int K, int f0, int rows_phase, int strips, int batch) {  // f0: the number of phases / chains
    const int P = 1 << K;  // SwtFusedGeom::P
    static const int forced = [] { const char* e = lab_env("PDWT_SWT_SEG"); return e ? atoi(e) : 0; }();
    int seg;
    if (forced > 0) {
        seg = (forced + P - 1) / P * P;
    } else {
        auto waves = [&](int s) { return (long long)batch * f0 * strips * ((rows_phase + s - 1) / s); };
        seg = 256;
        while (seg > P && waves(seg) < 768) seg -= P;
        // a batch beyond the Infinity Cache (more than 2^26 samples per launch): every row is an HBM round trip for its
        // wavefront, long walks of few wavefronts starve the memory system (4 x 2048^2, levels 4-5: 189 us at 88-row
        // walks) -- at most 32 rows
        if ((long long)batch * rows_phase * f0 * strips * 256 > (1LL << 26) && seg > 32) seg = 32 / P * P;
    }
    if (seg > rows_phase) seg = (rows_phase + P - 1) / P * P;
    return seg;
}

template <int K, int F0, int C>
static hipError_t run_inv(SwtFusedArgs& a, int batch, bool gen, hipStream_t s) {
    constexpr int NT = 64;
    a.wk = swt_walk(a.Nr, a.Nc, F0, C);
    a.strips = swt_walk_strips(a.wk, a.Nc, C * SwtInvGeom<K, F0, C>::V);
    a.seg_rows = fused_seg_rows(K, a.wk.phases, a.wk.rows_phase, a.strips, batch);
    a.segs = cdiv(a.wk.rows_phase, a.seg_rows);
    const long long waves = (long long)batch * a.wk.phases * a.segs * a.strips;
    const dim3 grid(8u * (unsigned)cdivll(waves, 8));  // swt_fused_wave
    if (gen) hipLaunchKernelGGL((swt2_inv_fused_kernel<K, F0, 4, C, NT, true>), grid, dim3(NT), 0, s, a, waves);
    else hipLaunchKernelGGL((swt2_inv_fused_kernel<K, F0, 4, C, NT, false>), grid, dim3(NT), 0, s, a, waves);
    return hipGetLastError();
}

template <int K, int F0>
static hipError_t run(SwtFusedArgs& a, bool inverse, int batch, bool gen, hipStream_t s) {
    using G = SwtFusedGeom<K, F0>;
    constexpr int NT = 64;  // one wavefront per workgroup: up to 512 VGPRs each, placed on any free SIMD
    if (inverse) {
        // 16-B lanes.  8-B lanes (128-column strips: twice the wavefronts per segment length, half the registers) were
        // built to afford longer segments for ONE image and measured no better: 2048^2 levels 1-3 49.1-51.8 us at
        // 24-64 rows against 47.8 us (16-B lanes, 24 rows), levels 4-5 39.9-42.1 against 35.1-37.0 us
        // (profiles/r02y_bench_cfg4_lanes_sweep.txt).  PDWT_SWT_CPL=2 selects them for re-measurement.
#if defined(PDWT_DOUBLE)
        return run_inv<K, F0, 2>(a, batch, gen || (a.Nc % 2) != 0, s);  // fp64: two doubles per lane (the four-column lanes need > 512 registers)
#elif defined(PDWT_LAB_KERNELS)
        static const int forced = [] { const char* e = lab_env("PDWT_SWT_CPL"); return e ? atoi(e) : 0; }();
        return forced == 2 ? run_inv<K, F0, 2>(a, batch, gen, s) : run_inv<K, F0, 4>(a, batch, gen, s);
#else
        return run_inv<K, F0, 4>(a, batch, gen, s);
#endif
    }
    a.wk = swt_walk(a.Nr, a.Nc, F0, 4);
    a.strips = swt_walk_strips(a.wk, a.Nc, 4 * G::V);
    a.seg_rows = fused_seg_rows(K, a.wk.phases, a.wk.rows_phase, a.strips, batch);
    a.segs = cdiv(a.wk.rows_phase, a.seg_rows);
    const long long waves = (long long)batch * a.wk.phases * a.segs * a.strips;
    const unsigned grid = 8u * (unsigned)cdivll(cdivll(waves, NT / 64), 8);  // XCD-contiguous wavefront ranges (swt_fused_wave)
    if (gen) hipLaunchKernelGGL((swt2_fwd_fused_kernel<K, F0, NT, true>), dim3(grid), dim3(NT), 0, s, a, waves);
    else hipLaunchKernelGGL((swt2_fwd_fused_kernel<K, F0, NT, false>), dim3(grid), dim3(NT), 0, s, a, waves);
    return hipGetLastError();
}

template <int F0>
static hipError_t run4(Swt4Args& a, bool inverse, int batch, bool gen, hipStream_t s) {
    using G = Swt4Geom<F0>;
    constexpr int NT = 64;
    a.wk = swt_walk(a.Nr, a.Nc, F0, 4);
    a.strips = swt_walk_strips(a.wk, a.Nc, 4 * (inverse ? G::Vi : G::Vf));
    const int rows_phase = a.wk.rows_phase, phases = a.wk.phases;
    // phase rows per wavefront: a segment walks 9 rows it does not own (one input plane: cheap), a wavefront wants ~1 us per
    // row: the longest multiple of 8 that still gives ~768 wavefronts
    static const int forced = [] { const char* e = lab_env("PDWT_SWT_SEG"); return e ? atoi(e) : 0; }();
    int seg = 256;
    auto waves_of = [&](int sg) { return (long long)batch * phases * a.strips * cdiv(rows_phase, sg); };
    while (seg > 8 && waves_of(seg) < 768) seg -= 8;
    if (forced > 0) seg = (forced + 7) / 8 * 8;
    if (seg > rows_phase) seg = (rows_phase + 7) / 8 * 8;
    a.seg_rows = seg;
    a.segs = cdiv(rows_phase, seg);
    const long long waves = waves_of(seg);
    const unsigned grid = 8u * (unsigned)cdivll(cdivll(waves, NT / 64), 8);
    if (inverse) {
        static const int nri = [] { const char* e = lab_env("PDWT_SWT4_NRI"); return e ? atoi(e) : 4; }();  // load slots (A/B measurements)
        if (gen) hipLaunchKernelGGL((swt4_inv_fused_kernel<F0, 4, NT, true>), dim3(grid), dim3(NT), 0, s, a, waves);
        else if (nri == 2) hipLaunchKernelGGL((swt4_inv_fused_kernel<F0, 2, NT, false>), dim3(grid), dim3(NT), 0, s, a, waves);
        else hipLaunchKernelGGL((swt4_inv_fused_kernel<F0, 4, NT, false>), dim3(grid), dim3(NT), 0, s, a, waves);
    } else if (gen) {
        hipLaunchKernelGGL((swt4_fwd_fused_kernel<F0, NT, true>), dim3(grid), dim3(NT), 0, s, a, waves);
    } else {
        hipLaunchKernelGGL((swt4_fwd_fused_kernel<F0, NT, false>), dim3(grid), dim3(NT), 0, s, a, waves);
    }
    return hipGetLastError();
}

// in / out: the approximation planes on either side of the group; det[3 k + {0,1,2}] = H, V, D of level l0 + k;
// beta[k]: soft threshold the inverse applies to level l0 + k's details as it loads them (nullptr: none)
hipError_t launch_swt2_fused(const real_t* in, real_t* out, real_t* const* det, int Nr, int Nc, int l0, int K, bool inverse,
                             int hlen, const FilterBank& fb, const real_t* beta, int batch, hipStream_t s) {
    if (!swt2_fused_supported(hlen, Nr, Nc, l0, K, inverse)) return hipErrorNotSupported;  // a stale schedule falls back, never truncates taps
    // planes that do not start on 16 B (images of a batch whose size is not a multiple of four samples) take the GEN kernels too
    const int f0 = 1 << (l0 - 1);
    bool gen = swt_walk_general(Nr, Nc, f0) || (batch > 1 && ((long long)Nr * Nc) % 4 != 0);
    for (int k = 0; k < 3 * K; k++) gen = gen || (reinterpret_cast<uintptr_t>(det[k]) & 15);
    gen = gen || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15);
    static const bool force_gen = [] { const char* e = lab_env("PDWT_SWT_GEN"); return e && atoi(e) != 0; }();  // A/B on aligned sizes
    gen = gen || force_gen;
    if (gen) note_family("anysize");
    if (hlen == 4) {
        Swt4Args b;
        b.in = in; b.out = out; b.Nr = Nr; b.Nc = Nc; b.bstride = (long long)Nr * Nc;
        for (int k = 0; k < 2; k++) {
            b.H[k] = det[3 * k]; b.V[k] = det[3 * k + 1]; b.D[k] = det[3 * k + 2];
            b.beta[k] = beta ? beta[k] : real_t(0);
        }
        for (int j = 0; j < 4; j++) { b.lo[j] = fb.lo[j]; b.hi[j] = fb.hi[j]; }
        return l0 == 1 ? run4<1>(b, inverse, batch, gen, s) : run4<4>(b, inverse, batch, gen, s);
    }
    SwtFusedArgs a;
    a.in = in; a.out = out; a.Nr = Nr; a.Nc = Nc; a.bstride = (long long)Nr * Nc;
    for (int k = 0; k < kSwtFusedMaxLevels; k++) {
        a.H[k] = k < K ? det[3 * k] : nullptr;
        a.V[k] = k < K ? det[3 * k + 1] : nullptr;
        a.D[k] = k < K ? det[3 * k + 2] : nullptr;
        a.beta[k] = (beta && k < K) ? beta[k] : real_t(0);
    }
    a.lo[0] = fb.lo[0]; a.lo[1] = fb.lo[1]; a.hi[0] = fb.hi[0]; a.hi[1] = fb.hi[1];
    if (l0 == 1 && K == 2) return run<2, 1>(a, inverse, batch, gen, s);
    if (l0 == 1 && K == 3) return run<3, 1>(a, inverse, batch, gen, s);
    if (l0 == 4 && K == 2) return run<2, 8>(a, inverse, batch, gen, s);
    if (l0 == 4 && K == 3) return run<3, 8>(a, inverse, batch, gen, s);
    return hipErrorNotSupported;
}

}  // namespace pdwt
