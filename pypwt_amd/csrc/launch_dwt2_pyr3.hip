// launch_dwt2_pyr3.hip -- launchers of the three-levels-per-launch 2D DWT kernels for small images
// (dwt2_pyr3_kernels.hpp).  Compiled in both builds (the kernels are written over real_t).
#include "dwt2_pyr3_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <cstdlib>

namespace pdwt {

constexpr size_t kPyr3LdsMax = 150 * 1024;

template <int HLEN, int T>
static constexpr size_t pyr3_fwd_lds() { return (size_t)Pyr3FwdGeom<HLEN, T>::LDS * sizeof(real_t); }
template <int HLEN, int T>
static constexpr size_t pyr3_inv_lds() { return (size_t)Pyr3InvGeom<HLEN, 8 * T>::LDS * sizeof(real_t); }

// even filters of at most 16 taps whose smallest tile fits the LDS (fp64: at most 8 taps), even sizes at all three levels
template <int HLEN>
static constexpr bool pyr3_fits() { return pyr3_fwd_lds<HLEN, 2>() <= kPyr3LdsMax && pyr3_inv_lds<HLEN, 2>() <= kPyr3LdsMax; }
bool dwt2_pyr3_supported(int hlen, int N0r, int N0c) {
    if ((N0r % 8) || (N0c % 8) || N0r < 8 || N0c < 8) return false;
    switch (hlen) {
        case 2: return pyr3_fits<2>();
        case 4: return pyr3_fits<4>();
        case 6: return pyr3_fits<6>();
        case 8: return pyr3_fits<8>();
        case 10: return pyr3_fits<10>();
        case 12: return pyr3_fits<12>();
        case 14: return pyr3_fits<14>();
        case 16: return pyr3_fits<16>();
    }
    return false;
}

// Tile edge T at the third level (a workgroup covers an 8T x 8T block of the input): the launch is a latency chain of
// seven barrier-separated phases, so what matters is how much each workgroup has to do, as long as there are about as
// many workgroups as CUs -- the largest T that still gives ~256 of them.  Measured (tools/smalltime.py,
// profiles/r02y_pyr3_sweep.txt, forward+inverse, us): 512^2 db2 L3  T=8: 12.2  T=4: 9.9  T=2: 10.9 (two launches each
// way: 15.6); 256^2 haar L3  7.9 / 8.0 / 7.3 (15.5); 64 x 128^2 db4 L3  20.3 / 24.0 / 34.6 (27.0).
// PDWT_PYR3_T / PDWT_PYR3_NT override (tuning).
static int pyr3_tile(int N0r, int N0c, int batch) {
    static const int forced = [] { const char* e = lab_env("PDWT_PYR3_T"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 4 || forced == 8) return forced;
    for (int T = 8; T > 2; T /= 2)
        if ((long long)cdiv(N0c / 8, T) * cdiv(N0r / 8, T) * batch >= 200) return T;
    return 2;
}
static int pyr3_threads(int T) {
    static const int forced = [] { const char* e = lab_env("PDWT_PYR3_NT"); return e ? atoi(e) : 0; }();
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    return T == 8 ? 1024 : 512;
}

template <int HLEN, int T, int NT>
static hipError_t run_fwd_t(Pyr3Args& a, int batch, hipStream_t s) {
    constexpr size_t lds = pyr3_fwd_lds<HLEN, T>();
    if constexpr (lds > kPyr3LdsMax) {
        if constexpr (T > 2) return run_fwd_t<HLEN, T / 2, NT>(a, batch, s);  // long filters, fp64: the next smaller tile
        else return hipErrorNotSupported;
    } else {
        static std::atomic<bool> big[64] = {};
        const hipError_t e = allow_big_lds(dwt2_fwd_pyr3_kernel<HLEN, T, NT>, lds, big);
        if (e != hipSuccess) return e;
        a.tiles_x = cdiv(a.N0c / 8, T);
        a.tiles_y = cdiv(a.N0r / 8, T);
        hipLaunchKernelGGL((dwt2_fwd_pyr3_kernel<HLEN, T, NT>), dim3(a.tiles_x * a.tiles_y, batch), dim3(NT), lds, s, a);
        return hipGetLastError();
    }
}
template <int HLEN, int T, int NT>
static hipError_t run_inv_t(Pyr3Args& a, int batch, hipStream_t s) {
    constexpr size_t lds = pyr3_inv_lds<HLEN, T>();
    if constexpr (lds > kPyr3LdsMax) {
        if constexpr (T > 2) return run_inv_t<HLEN, T / 2, NT>(a, batch, s);
        else return hipErrorNotSupported;
    } else {
        constexpr int T0 = 8 * T;
        static std::atomic<bool> big[64] = {};
        const hipError_t e = allow_big_lds(dwt2_inv_pyr3_kernel<HLEN, T0, NT>, lds, big);
        if (e != hipSuccess) return e;
        a.tiles_x = cdiv(a.N0c, T0);
        a.tiles_y = cdiv(a.N0r, T0);
        hipLaunchKernelGGL((dwt2_inv_pyr3_kernel<HLEN, T0, NT>), dim3(a.tiles_x * a.tiles_y, batch), dim3(NT), lds, s, a);
        return hipGetLastError();
    }
}

#define PDWT_PYR3_DISPATCH(fn)                                                                                     \
    const int T = pyr3_tile(a.N0r, a.N0c, batch), NT = pyr3_threads(T);                                            \
    if (T == 2) return NT == 1024 ? fn<HLEN, 2, 1024>(a, batch, s) : NT == 512 ? fn<HLEN, 2, 512>(a, batch, s) : fn<HLEN, 2, 256>(a, batch, s); \
    if (T == 4) return NT == 1024 ? fn<HLEN, 4, 1024>(a, batch, s) : NT == 512 ? fn<HLEN, 4, 512>(a, batch, s) : fn<HLEN, 4, 256>(a, batch, s); \
    return NT == 1024 ? fn<HLEN, 8, 1024>(a, batch, s) : NT == 512 ? fn<HLEN, 8, 512>(a, batch, s) : fn<HLEN, 8, 256>(a, batch, s);

template <int HLEN>
static hipError_t run_fwd(Pyr3Args& a, int batch, hipStream_t s) { PDWT_PYR3_DISPATCH(run_fwd_t) }
template <int HLEN>
static hipError_t run_inv(Pyr3Args& a, int batch, hipStream_t s) { PDWT_PYR3_DISPATCH(run_inv_t) }
#undef PDWT_PYR3_DISPATCH

// in (N0r, N0c) -> det[k] = (H, V, D) of the three levels (finest first), out = the approximation of the third
hipError_t launch_dwt2_fwd_pyr3(const real_t* in, real_t* const det[9], real_t* out, int N0r, int N0c, int hlen,
                                const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_pyr3_supported(hlen, N0r, N0c)) return hipErrorNotSupported;
    Pyr3Args a;
    a.in = in; a.out = out; a.N0r = N0r; a.N0c = N0c; a.fb = fb;
    for (int k = 0; k < 9; k++) a.det[k / 3][k % 3] = det[k];
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

// app = the approximation of the third level, det as above -> out (N0r, N0c)
hipError_t launch_dwt2_inv_pyr3(const real_t* app, real_t* const det[9], real_t* out, int N0r, int N0c, int hlen,
                                const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_pyr3_supported(hlen, N0r, N0c)) return hipErrorNotSupported;
    Pyr3Args a;
    a.in = app; a.out = out; a.N0r = N0r; a.N0c = N0c; a.fb = fb;
    for (int k = 0; k < 9; k++) a.det[k / 3][k % 3] = det[k];
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
