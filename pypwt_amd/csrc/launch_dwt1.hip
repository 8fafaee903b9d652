// launch_dwt1.hip -- instantiations + launchers of the (batched) 1D DWT level kernels (gfx950).
#include "dwt1_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

namespace pdwt {

// 256 threads produce 1024 (long rows) or 256 (short rows) outputs of one row; the grid is
// flattened over (row, tile) so that both one 2^24-sample row and thousands of short rows
// give the dispatcher >> 256 workgroups.
template <int HLEN, int TXO, int NT>
static hipError_t run_fwd(const Fwd1DArgs& a, hipStream_t s) {
    const int tiles = cdiv(a.Nc2, TXO);
    const size_t lds = (size_t)fwd1d_lds_floats<TXO>(a.hlen) * sizeof(real_t);
    hipLaunchKernelGGL((dwt1_fwd_kernel<HLEN, TXO, NT>), dim3((unsigned)((long long)tiles * a.rows)), dim3(NT),
                       lds, s, a, tiles);
    return hipGetLastError();
}

template <int HLEN, int TXO, int NT>
static hipError_t run_inv(const Inv1DArgs& a, hipStream_t s) {
    const int tiles = cdiv(a.Nc, 2 * TXO);
    const size_t lds = (size_t)inv1d_lds_floats<TXO>(a.hlen) * sizeof(real_t);
    hipLaunchKernelGGL((dwt1_inv_kernel<HLEN, TXO, NT>), dim3((unsigned)((long long)tiles * a.rows)), dim3(NT),
                       lds, s, a, tiles);
    return hipGetLastError();
}

hipError_t launch_dwt1_fwd(const Fwd1DArgs& a, hipStream_t s) {
    const bool wide = a.Nc2 >= 2048;
    if (a.hlen & 1) return wide ? run_fwd<0, 1024, 256>(a, s) : run_fwd<0, 256, 256>(a, s);
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return wide ? run_fwd<h, 1024, 256>(a, s) : run_fwd<h, 256, 256>(a, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return wide ? run_fwd<0, 1024, 256>(a, s) : run_fwd<0, 256, 256>(a, s);
    }
}

hipError_t launch_dwt1_inv(const Inv1DArgs& a, hipStream_t s) {
    const bool wide = a.Nc >= 4096;
    if (a.hlen & 1) return wide ? run_inv<0, 1024, 256>(a, s) : run_inv<0, 256, 256>(a, s);
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return wide ? run_inv<h, 1024, 256>(a, s) : run_inv<h, 256, 256>(a, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return wide ? run_inv<0, 1024, 256>(a, s) : run_inv<0, 256, 256>(a, s);
    }
}

}  // namespace pdwt
