// launch_dwt2_wave.hip -- launchers of the wave-per-tile 2D DWT level kernels (dwt2_wave_kernels.hpp).
//
// try_launch_* return hipErrorNotSupported when the level does not meet the kernels' preconditions
// (even filter length <= 8, row length a multiple of 4 (forward) / coefficient row length even and
// Nc == 2 Ncc (inverse), 16-B aligned buffers); the caller then falls back to the LDS-tiled kernels.
//
// Geometry: a wavefront owns a strip of 256 image columns and `seg` output rows (forward) or coefficient
// rows (inverse).  The host picks `seg` so that a level has about ONE wavefront per SIMD (1024 on the
// chip) and never less than one unrolled group: short segments re-read the hlen-2 rows they share with
// their neighbour more often, long ones leave SIMDs idle.  Measured (tools/wbench.hip, db4, one image,
// profiles/r02a_wbench_b1.txt): 4096^2 forward 24.8 / 23.1 / 22.1 / 21.7 / 27.8 us with 8192 / 4096 / 2048 /
// 1024 / 512 wavefronts (LDS tiles: 25.2 one tile per workgroup, 23.8 streaming; float4 copy 19.6);
// inverse 28.2 / 24.9 / 22.7 / 22.7 / 29.0 us (LDS tiles 23.6).  Register prefetch (3 rows in flight per
// wavefront), not occupancy, covers the memory latency.  Below 2048^2 a level is launch-bound (a
// 2 MB float4 copy takes 3.3 us, an empty kernel 2.5) and the LDS tiles' shorter per-workgroup critical
// path wins (1024^2: 4.3 vs 4.8 us): the host keeps them there (launch_dwt2.hip).
// The predicate-free kernels (GUARD = false) need whole strips and whole groups; anything else
// runs the guarded twins.
#include "dwt2_wave_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

namespace pdwt {

static void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kWaveTarget = 1024;  // wavefronts per launch the segment length aims for (1 per SIMD)

// rows per wavefront: multiple of `unit`, about strips * ceil(rows / seg) * batch = kWaveTarget, at most 64
static int pick_seg(int rows, int strips, int batch, int unit, int hint) {
    // A level that lives in the Infinity Cache: about one wavefront per SIMD, at most 64 rows each.  A level of a batch
    // beyond it (a wavefront moves 4 KiB per output row / coefficient row pair: 2^17 of them = 512 MiB; half that
    // when the batch is large, i.e. every level is cold): every row then costs a wavefront an HBM round trip and
    // 16-row walks of many wavefronts win -- 4 x 4096^2: 99 us at 16 rows against 105-108 at 32-64; the 512^2
    // inverse level of 128 images: 67 us against 161 us in 67-row walks (profiles/r02y_wbench_b{2,4}.txt,
    // r02z_bench_cfg5_shard.json vs r02y_bench_cfg5.json).
    const long long work = (long long)rows * strips * batch;
    const bool cold = work >= (1LL << 17) || (work >= (1LL << 16) && batch > 2);
    long long seg = hint > 0 ? hint : (cold ? 16 : work / kWaveTarget);
    seg = (seg / unit) * unit;
    if (seg < unit) seg = unit;
    if (hint <= 0 && seg > 64) seg = (64 / unit) * unit;
    return (int)seg;
}

template <int HLEN>
static hipError_t run_fwd(const Fwd2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    constexpr int NT = 256, G2 = FwdWaveGeom<HLEN>::GR / 2;
    FwdWaveArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.Nr2 = g.Nr2; a.Nc2 = g.Nc2;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Nc, 256);
    a.seg_out = pick_seg(g.Nr2, a.strips, batch, G2, seg_hint);
    a.segs = cdiv(g.Nr2, a.seg_out);
    interleave(a.fb, g.fb);
    const int nblk = cdiv(a.strips * a.segs, NT / 64);
    const dim3 grid(8 * cdiv(nblk, 8), batch);
    const bool plain = (g.Nc % 256) == 0 && (g.Nr2 % G2) == 0;
    if (plain) hipLaunchKernelGGL((dwt2_fwd_wave_kernel<HLEN, false, NT>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((dwt2_fwd_wave_kernel<HLEN, true, NT>), grid, dim3(NT), 0, s, a);
    return hipGetLastError();
}

template <int HLEN>
static hipError_t run_inv(const Inv2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    constexpr int NT = 256, GR = InvWaveGeom<HLEN>::GR;
    InvWaveArgs a;
    a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D; a.out = g.out;
    a.Nrc = g.Nrc; a.Ncc = g.Ncc; a.Nr = g.Nr; a.Nc = g.Nc;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Ncc, 128);
    a.seg_pairs = pick_seg(g.Nrc, a.strips, batch, GR, seg_hint);
    a.segs = cdiv(g.Nrc, a.seg_pairs);
    interleave(a.fb, g.fb);
    for (int d = 0; d < HLEN / 2; d++) {
        a.pl[d].x = g.fb.lo[HLEN - 2 - 2 * d]; a.pl[d].y = g.fb.lo[HLEN - 1 - 2 * d];
        a.ph[d].x = g.fb.hi[HLEN - 2 - 2 * d]; a.ph[d].y = g.fb.hi[HLEN - 1 - 2 * d];
    }
    const int nblk = cdiv(a.strips * a.segs, NT / 64);
    const dim3 grid(8 * cdiv(nblk, 8), batch);
    const bool plain = (g.Ncc % 128) == 0 && (g.Nrc % GR) == 0 && g.Nr == 2 * g.Nrc;
    if (plain) hipLaunchKernelGGL((dwt2_inv_wave_kernel<HLEN, false, NT>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((dwt2_inv_wave_kernel<HLEN, true, NT>), grid, dim3(NT), 0, s, a);
    return hipGetLastError();
}

#if !defined(PDWT_DOUBLE) && defined(PDWT_LAB_KERNELS)
// ---- two forward levels per wavefront (dwt2_fwd2_wave): same contract as launch_dwt2_fwd_pyr2.  An experiment that measured
// slower than two launches: compiled into the LAB build only (libpypwt_amd_lab.so, -DPDWT_LAB_KERNELS), not into the product
bool dwt2_wave2_supported(int hlen, int N0r, int N0c) {
    return !(hlen & 1) && hlen >= 2 && hlen <= 8 && (N0r % 4) == 0 && (N0c % 16) == 0 && N0r >= 4 && N0c >= 16 &&
           (long long)N0r * N0c < (1LL << 30);  // 32-bit byte offsets inside one image
}

template <int HLEN>
static hipError_t run_fwd2(FwdWave2Args& a, int batch, int seg_hint, hipStream_t s) {
    constexpr int NT = 256;
    a.strips = cdiv(a.N0c, 240);
    const int N2r = a.N0r / 4;
    // level-2 rows per wavefront: about one wavefront per SIMD; every segment recomputes 3 hlen - 6 image rows
    long long seg = seg_hint > 0 ? seg_hint : cdivll((long long)N2r * a.strips * batch, kWaveTarget);
    if (seg < 2) seg = 2;
    if (seg_hint <= 0 && seg > 32) seg = 32;
    a.seg2_out = (int)seg;
    a.segs = cdiv(N2r, a.seg2_out);
    const int nblk = cdiv(a.strips * a.segs, NT / 64);
    hipLaunchKernelGGL((dwt2_fwd2_wave_kernel<HLEN, NT>), dim3(8 * cdiv(nblk, 8), batch), dim3(NT), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_dwt2_fwd_wave2(const float* in, float* const det1[3], float* const band2[4], int N0r, int N0c,
                                 int hlen, const FilterBank& fb, int batch, hipStream_t s, int seg_hint) {
    if (!dwt2_wave2_supported(hlen, N0r, N0c)) return hipErrorNotSupported;
    if (!aligned16(in) || !aligned16(det1[0]) || !aligned16(det1[1]) || !aligned16(det1[2]) || !aligned16(band2[0]) ||
        !aligned16(band2[1]) || !aligned16(band2[2]) || !aligned16(band2[3]))
        return hipErrorNotSupported;
    FwdWave2Args a;
    a.in = in; a.H1 = det1[0]; a.V1 = det1[1]; a.D1 = det1[2];
    a.A2 = band2[0]; a.H2 = band2[1]; a.V2 = band2[2]; a.D2 = band2[3];
    a.N0r = N0r; a.N0c = N0c;
    a.in_bstride = (long long)N0r * N0c;
    a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    interleave(a.fb, fb);
    switch (hlen) {
        case 2: return run_fwd2<2>(a, batch, seg_hint, s);
        case 4: return run_fwd2<4>(a, batch, seg_hint, s);
        case 6: return run_fwd2<6>(a, batch, seg_hint, s);
        case 8: return run_fwd2<8>(a, batch, seg_hint, s);
    }
    return hipErrorNotSupported;
}
#elif !defined(PDWT_DOUBLE)
bool dwt2_wave2_supported(int, int, int) { return false; }
hipError_t launch_dwt2_fwd_wave2(const float*, float* const[3], float* const[4], int, int, int, const FilterBank&, int, hipStream_t,
                                 int) { return hipErrorNotSupported; }
#endif

hipError_t try_launch_dwt2_fwd_wave(const Fwd2DArgs& a, int batch, hipStream_t s, int seg_hint) {
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > 8) return hipErrorNotSupported;
    if ((a.Nc & 3) || (a.in_bstride & 3) || (a.out_bstride & 1) || a.Nc2 * 2 != a.Nc) return hipErrorNotSupported;
    if ((long long)a.Nc * (long long)sizeof(real_t) >= (1LL << 31)) return hipErrorNotSupported;  // 32-bit byte offsets inside a row
    if (!aligned16(a.in) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    switch (a.hlen) {
        case 2: return run_fwd<2>(a, batch, seg_hint, s);
        case 4: return run_fwd<4>(a, batch, seg_hint, s);
        case 6: return run_fwd<6>(a, batch, seg_hint, s);
        case 8: return run_fwd<8>(a, batch, seg_hint, s);
    }
    return hipErrorNotSupported;
}

hipError_t try_launch_dwt2_inv_wave(const Inv2DArgs& a, int batch, hipStream_t s, int seg_hint) {
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > 8) return hipErrorNotSupported;
    if ((a.Ncc & 1) || a.Nc != 2 * a.Ncc || (a.in_bstride & 1) || (a.out_bstride & 3)) return hipErrorNotSupported;
    if (a.Nr > 2 * a.Nrc || a.Nr < 2 * a.Nrc - 1) return hipErrorNotSupported;
    if ((long long)a.Nc * (long long)sizeof(real_t) >= (1LL << 31)) return hipErrorNotSupported;
    if (!aligned16(a.out) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    switch (a.hlen) {
        case 2: return run_inv<2>(a, batch, seg_hint, s);
        case 4: return run_inv<4>(a, batch, seg_hint, s);
        case 6: return run_inv<6>(a, batch, seg_hint, s);
        case 8: return run_inv<8>(a, batch, seg_hint, s);
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
