// launch_swt.hip -- instantiations + launchers of the undecimated (a-trous) kernels (gfx950).
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#include "swt_kernels.hpp"

namespace pdwt {

// 64 columns x 16 rows of one dilation phase per 256-thread workgroup; blockIdx.y enumerates
// (row tile, phase).
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd(const Swt2DArgs& a, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    const size_t lds = (size_t)swt2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(swt2_fwd_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)swt2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    const int M = cdiv(a.Nr, a.f);  // rows of the longest dilation phase (f need not divide Nr: the tiles wrap rows, not phase indices)
    dim3 grid(cdiv(a.Nc, TX), cdiv(M, TY) * a.f, batch);
    hipLaunchKernelGGL((swt2_fwd_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv(const Swt2DArgs& a, int batch, hipStream_t s) {
    static std::atomic<bool> big[64] = {};
    const size_t lds = (size_t)swt2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(swt2_inv_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)swt2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    const int M = cdiv(a.Nr, a.f);  // rows of the longest dilation phase (f need not divide Nr: the tiles wrap rows, not phase indices)
    dim3 grid(cdiv(a.Nc, TX), cdiv(M, TY) * a.f, batch);
    hipLaunchKernelGGL((swt2_inv_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

// vectorised twins (launch_swt_vec.hip): hipErrorNotSupported when their preconditions do not hold
hipError_t try_launch_swt2_vec(const Swt2DArgs& a, bool inverse, int batch, hipStream_t s);

hipError_t launch_swt2_fwd(const Swt2DArgs& a, int batch, hipStream_t s) {
    {
        const hipError_t e = try_launch_swt2_vec(a, false, batch, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.hlen & 1) return run_fwd<0, 64, 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_fwd<h, 64, 16, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_fwd<0, 64, 16, 256>(a, batch, s);
    }
}

hipError_t launch_swt2_inv(const Swt2DArgs& a, int batch, hipStream_t s) {
    {
        const hipError_t e = try_launch_swt2_vec(a, true, batch, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.hlen & 1) return run_inv<0, 64, 16, 256>(a, batch, s);
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_inv<h, 64, 16, 256>(a, batch, s);
        PDWT_EVEN_HLENS(X)
#undef X
        default:
            return run_inv<0, 64, 16, 256>(a, batch, s);
    }
}

// four samples per work item (swt1_*_vec_kernel): rows filtered along x, Nc % 4 == 0, even compile-time filter length
static bool swt1_vec_ok(const SwtPassArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(real_t) - 1)) == 0; };
    static const bool off = lab_env("PDWT_NO_SWT1_VEC") != nullptr;  // A/B measurements
    // (fp64: the kernel keeps all taps of both bands of a work item in registers -- beyond 30 taps of doubles the inverse spills
    // to scratch, tools/spillscan.py; those lengths take the one-output-per-thread kernel)
    if (sizeof(real_t) == 8 && a.hlen > 30) return false;
    return !off && !a.along_y && !(a.Nc & 3) && !(a.hlen & 1) && a.hlen >= 2 && a.hlen <= kMaxTaps && al(a.in0) &&
           (!a.in1 || al(a.in1)) && al(a.out0) && (!a.out1 || al(a.out1));
}

// register-blocked row kernels of swt_split_kernels.hpp (launch_swt_split.hip) for filters of >= 10 taps
hipError_t try_launch_swt1_split(const SwtPassArgs& a, bool inverse, hipStream_t s);

hipError_t launch_swt_pass_fwd(const SwtPassArgs& a, hipStream_t s) {
    if (a.images == 1) {  // (several planes per launch: the direct kernel only)
        const hipError_t e = try_launch_swt1_split(a, false, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.images == 1 && swt1_vec_ok(a)) {
        const unsigned grid = (unsigned)cdivll((long long)a.Nr * (a.Nc >> 2), 256);
        switch (a.hlen) {
#define X(h)                                                                                      \
    case h:                                                                                       \
        hipLaunchKernelGGL((swt1_fwd_vec_kernel<h, 256>), dim3(grid), dim3(256), 0, s, a);        \
        return hipGetLastError();
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    const long long total = (long long)a.Nr * a.Nc * a.images;
    hipLaunchKernelGGL((swt_pass_fwd_kernel<256>), dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_swt_pass_inv(const SwtPassArgs& a, hipStream_t s) {
    if (a.images == 1) {
        const hipError_t e = try_launch_swt1_split(a, true, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.images == 1 && swt1_vec_ok(a)) {
        const unsigned grid = (unsigned)cdivll((long long)a.Nr * (a.Nc >> 2), 256);
        switch (a.hlen) {
#define X(h)                                                                                      \
    case h:                                                                                       \
        if constexpr (sizeof(real_t) == 4 || h <= 30) {                                           \
            hipLaunchKernelGGL((swt1_inv_vec_kernel<h, 256>), dim3(grid), dim3(256), 0, s, a);    \
            return hipGetLastError();                                                             \
        }                                                                                         \
        break;
            PDWT_EVEN_HLENS(X)
#undef X
        }
    }
    const long long total = (long long)a.Nr * a.Nc * a.images;
    hipLaunchKernelGGL((swt_pass_inv_kernel<256>), dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace pdwt
