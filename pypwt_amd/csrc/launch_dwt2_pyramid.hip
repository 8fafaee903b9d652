// launch_dwt2_pyramid.hip -- launchers of the two-levels-per-launch 2D DWT kernels (gfx950).
//
// Used by the host only for SMALL level pairs (plan.cpp: pyramid_pairs), where a launch's fixed cost
// dominates: kbench on MI355X, db4, one image (profiles/r01e_kbench_pyramid.txt):
//     level input   two launches   one pyramid launch
//       512^2         8.8 us            5.4 us   (forward)      8.4 -> 5.3 us (inverse)
//      1024^2         9.3 us            5.5 us                   9.0 -> 6.2 us
//      2048^2        13.8 us           12.2 us                  14.4 -> 12.7 us
//      4096^2        33.6 us           40.3 us  (slower: 1.9x recomputed halo, half the occupancy)
#include "dwt2_pyramid_kernels.hpp"
#include "dwt2_strip_kernels.hpp"
#include <cstdlib>

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

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool dwt2_pyramid_supported(int hlen, int N0r, int N0c, bool inverse) {
    // even filters of at most 16 taps (LDS), even sizes at both levels (exact periodization), 16-B aligned rows at level l+1: rows
    // of N0c % 8 == 0 samples.  (Until round 5: N0c % 16 == 0 -- the forward always stored the bands of level l+2 in pairs, and the
    // inverse now stages them in pairs where their rows are not whole quads: a 1000 x 1000 image ran level launches for want of it,
    // dwt2 db4 1000^2 L3 forward+inverse 25.9 us against 18.5 for 1024^2.)
    (void)inverse;
    return !(hlen & 1) && hlen >= 2 && hlen <= 16 && (N0r % 4) == 0 && (N0c % 8) == 0 && N0r >= 4 && N0c >= 16;
}
// the streaming strips carry (hlen - 2) rows per level in LDS: filters of at most 8 taps
bool dwt2_strip_supported(int hlen, int N0r, int N0c) { return hlen <= 8 && (N0c % 16) == 0 && dwt2_pyramid_supported(hlen, N0r, N0c, true); }

template <int HLEN>
static hipError_t run_fwd(FwdPyr2Args& a, int batch, hipStream_t s) {
    constexpr int TX2 = 32, TY2 = 8, NT = 512;
    constexpr size_t lds = (size_t)Pyr2Geom<HLEN, TX2, TY2>::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 150 * 1024, "fits the LDS (filters of more than 8 taps opt in to more than 64 KiB)");
    static std::atomic<bool> big[64] = {};
    const hipError_t e = allow_big_lds(dwt2_fwd_pyr2_kernel<HLEN, TX2, TY2, NT>, lds, big);
    if (e != hipSuccess) return e;
    a.tiles_x = cdiv(a.N0c / 4, TX2);
    a.tiles_y = cdiv(a.N0r / 4, TY2);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    hipLaunchKernelGGL((dwt2_fwd_pyr2_kernel<HLEN, TX2, TY2, NT>), dim3(8 * chunk, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv_t(InvPyr2Args& a, int batch, hipStream_t s) {
    constexpr size_t lds = (size_t)InvPyr2Geom<HLEN, TX, TY>::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 150 * 1024, "fits the LDS");
    static std::atomic<bool> big[64] = {};
    const hipError_t e = allow_big_lds(dwt2_inv_pyr2_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    a.tiles_x = cdiv(a.N0c, 2 * TX);
    a.tiles_y = cdiv(a.N0r, 2 * TY);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    hipLaunchKernelGGL((dwt2_inv_pyr2_kernel<HLEN, TX, TY, NT>), dim3(8 * chunk, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

// 128x32-sample tiles of 512 threads from 2^20 samples on (1024^2 db4: 5.5 us against 6.1 us with 128x16 tiles of 256
// threads; 512^2: 4.95 against 4.85 us -- profiles/r02y_kbench_tiles.txt)
template <int HLEN>
static hipError_t run_inv(InvPyr2Args& a, int batch, hipStream_t s) {
    if ((long long)batch * a.N0r * a.N0c >= (1LL << tune::pyr2_inv_big_log2)) return run_inv_t<HLEN, 64, 16, 512>(a, batch, s);
    return run_inv_t<HLEN, 64, 8, 256>(a, batch, s);
}

// in (N0r,N0c) -> details of level l (det1 = H,V,D) and all four bands of level l+1 (band2 = A,H,V,D)
hipError_t launch_dwt2_fwd_pyr2(const float* in, float* const det1[3], float* const band2[4], int N0r, int N0c,
                                int hlen, const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_pyramid_supported(hlen, N0r, N0c, false)) return hipErrorNotSupported;
    if (!al16(in) || !al16(det1[0]) || !al16(det1[1]) || !al16(det1[2]) || !al16(band2[0]) || !al16(band2[1]) ||
        !al16(band2[2]) || !al16(band2[3]))
        return hipErrorNotSupported;
    FwdPyr2Args a;
    a.in = in; a.H1 = det1[0]; a.V1 = det1[1]; a.D1 = det1[2];
    a.A2 = band2[0]; a.H2 = band2[1]; a.V2 = band2[2]; a.D2 = band2[3];
    a.N0r = N0r; a.N0c = N0c;
    a.in_bstride = (long long)N0r * N0c;
    a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    interleave(a.fb, fb);
    switch (hlen) {
        case 2: return run_fwd<2>(a, batch, s);
        case 4: return run_fwd<4>(a, batch, s);
        case 6: return run_fwd<6>(a, batch, s);
        case 8: return run_fwd<8>(a, batch, s);
        case 10: return run_fwd<10>(a, batch, s);
        case 12: return run_fwd<12>(a, batch, s);
        case 14: return run_fwd<14>(a, batch, s);
        case 16: return run_fwd<16>(a, batch, s);
    }
    return hipErrorNotSupported;
}

// Streaming strips (dwt2_strip_kernels.hpp): same contract as launch_dwt2_fwd_pyr2, for LARGE inputs.
// kbench, MI355X, db4, 8 images of 4096^2 (profiles/r02a_kbench_strip.txt): levels 1+2 as two launches
// 284 us, as one strip launch 232 us (A_1 never goes to HBM); one image: 32 us vs 36 us (slower: the
// launch then has too few workgroups for its longer per-workgroup critical path), so the host only uses it
// when at least 2^26 samples enter the pair (plan.cpp: strip_pairs_fwd).
template <int HLEN>
static hipError_t run_fwd_strip(FwdStrip2Args& a, int batch, hipStream_t s) {
    constexpr int TX2 = 32, NT = 256, PF = 2;
    constexpr size_t lds = (size_t)Strip2Geom<HLEN, TX2>::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 64 * 1024, "fits the default dynamic-LDS limit");
    a.strips = cdiv(a.N0c / 4, TX2);
    int seg2 = 128;  // longer segments amortise the 3(hlen-2)-row warm-up; keep >= 4 workgroups per CU
    while (seg2 > 16 && (long long)a.strips * cdiv(a.N0r / 4, seg2) * batch < 1024) seg2 >>= 1;
    a.seg2 = seg2;
    a.segs = cdiv(a.N0r / 4, seg2);
    hipLaunchKernelGGL((dwt2_fwd_strip2_kernel<HLEN, TX2, NT, PF>), dim3(a.strips * a.segs, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_dwt2_fwd_strip2(const float* in, float* const det1[3], float* const band2[4], int N0r, int N0c,
                                  int hlen, const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_strip_supported(hlen, N0r, N0c)) return hipErrorNotSupported;
    if (!al16(in) || !al16(det1[0]) || !al16(det1[1]) || !al16(det1[2]) || !al16(band2[0]) || !al16(band2[1]) ||
        !al16(band2[2]) || !al16(band2[3]))
        return hipErrorNotSupported;
    FwdStrip2Args a;
    a.in = in; a.H1 = det1[0]; a.V1 = det1[1]; a.D1 = det1[2];
    a.A2 = band2[0]; a.H2 = band2[1]; a.V2 = band2[2]; a.D2 = band2[3];
    a.N0r = N0r; a.N0c = N0c;
    a.in_bstride = (long long)N0r * N0c;
    a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    interleave(a.fb, fb);
    switch (hlen) {
        case 2: return run_fwd_strip<2>(a, batch, s);
        case 4: return run_fwd_strip<4>(a, batch, s);
        case 6: return run_fwd_strip<6>(a, batch, s);
        case 8: return run_fwd_strip<8>(a, batch, s);
    }
    return hipErrorNotSupported;
}

#ifdef PDWT_LAB_KERNELS  // the inverse strips never beat two launches: LAB build only
template <int HLEN>
static hipError_t run_inv_strip(InvStrip2Args& a, int batch, hipStream_t s) {
    constexpr int TX = 64, NT = 256;
    constexpr size_t lds = (size_t)InvStrip2Geom<HLEN, TX>::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 64 * 1024, "fits the default dynamic-LDS limit");
    a.strips = cdiv(a.N0c / 2, TX);
    int seg = 512;  // output rows per workgroup; one 16-row warm-up chunk per segment
    static const int forced = lab_env("PDWT_ISTRIP_SEG") ? atoi(lab_env("PDWT_ISTRIP_SEG")) : 0;  // A/B measurements
    if (forced > 0) seg = forced;
    while (seg > 64 && (long long)a.strips * cdiv(a.N0r, seg) * batch < 1024) seg >>= 1;
    a.seg_rows = seg;
    a.segs = cdiv(a.N0r, seg);
    hipLaunchKernelGGL((dwt2_inv_strip2_kernel<HLEN, TX, NT>), dim3(a.strips * a.segs, batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

// same contract as launch_dwt2_inv_pyr2, streaming-strip kernel for large inputs
hipError_t launch_dwt2_inv_strip2(const float* const band2[4], const float* const det1[3], float* out, int N0r,
                                  int N0c, int hlen, const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_strip_supported(hlen, N0r, N0c)) return hipErrorNotSupported;
    if (!al16(out) || !al16(det1[0]) || !al16(det1[1]) || !al16(det1[2]) || !al16(band2[0]) || !al16(band2[1]) ||
        !al16(band2[2]) || !al16(band2[3]))
        return hipErrorNotSupported;
    InvStrip2Args a;
    a.A2 = band2[0]; a.H2 = band2[1]; a.V2 = band2[2]; a.D2 = band2[3];
    a.H1 = det1[0]; a.V1 = det1[1]; a.D1 = det1[2];
    a.out = out; a.N0r = N0r; a.N0c = N0c;
    a.out_bstride = (long long)N0r * N0c;
    a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    interleave(a.fb, fb);
    switch (hlen) {
        case 2: return run_inv_strip<2>(a, batch, s);
        case 4: return run_inv_strip<4>(a, batch, s);
        case 6: return run_inv_strip<6>(a, batch, s);
        case 8: return run_inv_strip<8>(a, batch, s);
    }
    return hipErrorNotSupported;
}

#else
hipError_t launch_dwt2_inv_strip2(const float* const[4], const float* const[3], float*, int, int, int, const FilterBank&, int,
                                  hipStream_t) { return hipErrorNotSupported; }
#endif  // PDWT_LAB_KERNELS

hipError_t launch_dwt2_inv_pyr2(const float* const band2[4], const float* const det1[3], float* out, int N0r, int N0c,
                                int hlen, const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_pyramid_supported(hlen, N0r, N0c, true)) return hipErrorNotSupported;
    if (!al16(out) || !al16(det1[0]) || !al16(det1[1]) || !al16(det1[2]) || !al16(band2[0]) || !al16(band2[1]) ||
        !al16(band2[2]) || !al16(band2[3]))
        return hipErrorNotSupported;
    InvPyr2Args a;
    a.A2 = band2[0]; a.H2 = band2[1]; a.V2 = band2[2]; a.D2 = band2[3];
    a.H1 = det1[0]; a.V1 = det1[1]; a.D1 = det1[2];
    a.out = out; a.N0r = N0r; a.N0c = N0c;
    a.out_bstride = (long long)N0r * N0c;
    a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    interleave(a.fb, fb);
    switch (hlen) {
        case 2: return run_inv<2>(a, batch, s);
        case 4: return run_inv<4>(a, batch, s);
        case 6: return run_inv<6>(a, batch, s);
        case 8: return run_inv<8>(a, batch, s);
        case 10: return run_inv<10>(a, batch, s);
        case 12: return run_inv<12>(a, batch, s);
        case 14: return run_inv<14>(a, batch, s);
        case 16: return run_inv<16>(a, batch, s);
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
