// launch_swt.hip -- instantiations + launchers of the undecimated (a-trous) kernels (gfx950).
#include "launch.hpp"
#include "launch_util.hpp"
#include "swt_kernels.hpp"

namespace pdwt {

// 64 columns x 16 rows of one dilation phase per 256-thread workgroup; blockIdx.y enumerates
// (row tile, phase).
template <int HLEN, int TX, int TY, int NT>
static hipError_t run_fwd(const Swt2DArgs& a, int batch, hipStream_t s) {
    static bool big[64] = {};
    const size_t lds = (size_t)swt2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(swt2_fwd_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)swt2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    const int M = a.Nr / a.f;
    dim3 grid(cdiv(a.Nc, TX), cdiv(M, TY) * a.f, batch);
    hipLaunchKernelGGL((swt2_fwd_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

template <int HLEN, int TX, int TY, int NT>
static hipError_t run_inv(const Swt2DArgs& a, int batch, hipStream_t s) {
    static bool big[64] = {};
    const size_t lds = (size_t)swt2d_lds_floats<TX, TY>(HLEN ? HLEN : kMaxTaps) * sizeof(real_t);
    hipError_t e = allow_big_lds(swt2_inv_kernel<HLEN, TX, TY, NT>, lds, big);
    if (e != hipSuccess) return e;
    const size_t use = (size_t)swt2d_lds_floats<TX, TY>(a.hlen) * sizeof(real_t);
    const int M = a.Nr / a.f;
    dim3 grid(cdiv(a.Nc, TX), cdiv(M, TY) * a.f, batch);
    hipLaunchKernelGGL((swt2_inv_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), use, s, a);
    return hipGetLastError();
}

// vectorised twins: 128 columns x 16 rows of one phase, four columns per thread (16-B accesses)
template <int HLEN, bool INV>
static hipError_t run_vec(const Swt2DArgs& a, int batch, hipStream_t s) {
    constexpr int TX = 128, TY = 16, NT = 256;
    static bool big[64] = {};
    constexpr size_t lds = (size_t)swt2d_vec_lds_floats<TX, TY>(HLEN) * sizeof(real_t);
    const int M = a.Nr / a.f;
    dim3 grid(cdiv(a.Nc, TX), cdiv(M, TY) * a.f, batch);
    if (INV) {
        hipError_t e = allow_big_lds(swt2_inv_vec_kernel<HLEN, TX, TY, NT>, lds, big);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_inv_vec_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, s, a);
    } else {
        hipError_t e = allow_big_lds(swt2_fwd_vec_kernel<HLEN, TX, TY, NT>, lds, big);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_fwd_vec_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, s, a);
    }
    return hipGetLastError();
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(real_t) - 1)) == 0; }

// preconditions of the vectorised kernels: even compile-time filter length, whole 4-column groups,
// planes aligned for 4-element accesses
static bool vec_ok(const Swt2DArgs& a, bool inverse) {
    if ((a.hlen & 1) || a.hlen < 2 || a.hlen > kMaxTaps || (a.Nc & 3) || (a.bstride & 3)) return false;
    if (!al16(a.A) || !al16(a.H) || !al16(a.V) || !al16(a.D)) return false;
    return al16(inverse ? (const void*)a.out : (const void*)a.in);
}

hipError_t launch_swt2_fwd(const Swt2DArgs& a, int batch, hipStream_t s) {
    if (vec_ok(a, false)) {
        switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_vec<h, false>(a, batch, s);
            PDWT_EVEN_HLENS(X)
#undef X
        }
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
    if (vec_ok(a, true)) {
        switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_vec<h, true>(a, batch, s);
            PDWT_EVEN_HLENS(X)
#undef X
        }
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

hipError_t launch_swt_pass_fwd(const SwtPassArgs& a, hipStream_t s) {
    const long long total = (long long)a.Nr * a.Nc;
    hipLaunchKernelGGL((swt_pass_fwd_kernel<256>), dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_swt_pass_inv(const SwtPassArgs& a, hipStream_t s) {
    const long long total = (long long)a.Nr * a.Nc;
    hipLaunchKernelGGL((swt_pass_inv_kernel<256>), dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace pdwt
