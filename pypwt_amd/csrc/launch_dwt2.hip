// launch_dwt2.hip -- instantiations + launchers of the fused 2D DWT level kernels (gfx950).
#include "dwt2_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

namespace pdwt {

// Tile shape: 64 columns x TY rows of each sub-band per 256-thread workgroup (4 wavefronts,
// one 64-lane wavefront per output row).  Short filters use TY = 16 (about 40 KB of LDS ->
// 4 workgroups per CU); long filters amortise their (hlen-2)-row halo over TY = 32.
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd(const Fwd2DArgs& a, int batch, hipStream_t s) {
    static bool big[64] = {};
    const size_t lds = (size_t)fwd2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(float);
    hipError_t e = allow_big_lds(dwt2_fwd_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)fwd2d_lds_floats<TX, TY>(a.hlen) * sizeof(float);
    dim3 grid(cdiv(a.Nc2, TX), cdiv(a.Nr2, TY), batch);
    hipLaunchKernelGGL((dwt2_fwd_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv(const Inv2DArgs& a, int batch, hipStream_t s) {
    static bool big[64] = {};
    const size_t lds = (size_t)inv2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(float);
    hipError_t e = allow_big_lds(dwt2_inv_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)inv2d_lds_floats<TX, TY>(a.hlen) * sizeof(float);
    dim3 grid(cdiv(a.Nc, 2 * TX), cdiv(a.Nr, 2 * TY), batch);
    hipLaunchKernelGGL((dwt2_inv_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

hipError_t launch_dwt2_fwd(const Fwd2DArgs& a, int batch, hipStream_t s) {
    {
        const hipError_t e = try_launch_dwt2_fwd_fast(a, batch, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.hlen & 1) return run_fwd<0, 64, 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h)                                                       \
    case h:                                                        \
        if constexpr (h <= 12) return run_fwd<h, 64, 16, 256>(a, batch, s);  \
        else return run_fwd<h, 64, 32, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_fwd<0, 64, 16, 256>(a, batch, s);
    }
}

hipError_t launch_dwt2_inv(const Inv2DArgs& a, int batch, hipStream_t s) {
    {
        const hipError_t e = try_launch_dwt2_inv_fast(a, batch, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.hlen & 1) return run_inv<0, 64, 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h)                                                       \
    case h:                                                        \
        if constexpr (h <= 12) return run_inv<h, 64, 16, 256>(a, batch, s);  \
        else return run_inv<h, 64, 32, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_inv<0, 64, 16, 256>(a, batch, s);
    }
}

}  // namespace pdwt
