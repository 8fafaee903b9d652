// launch_swt_tail.hip -- launchers of the whole-transform-per-workgroup SWT kernels for batches of tiny images
// (swt2_tail_kernels.hpp).  Compiled in both builds (the kernels are written over real_t).
#include "launch.hpp"
#include "launch_util.hpp"
#include "swt2_tail_kernels.hpp"

#include <atomic>

namespace pdwt {

// images of at most 4096 samples (any sizes, powers of two or not), any filter length, at most 12 levels
bool swt2_tail_supported(int hlen, int Nr, int Nc, int L) {
    if (hlen < 1 || hlen > kMaxTaps || L < 1 || L > kSwtTailMaxLevels || Nr < 2 || Nc < 2) return false;
    return (long long)Nr * Nc <= kSwtTailMaxSamples;
}

// forward: in = the images -> det[3 (l - 1) + b] = band b of level l, out = A_L
// inverse: in = A_L, det as above (beta[l - 1]: soft threshold applied to level l's details as they are read) -> out = the images
hipError_t launch_swt2_tail(const real_t* in, real_t* const* det, real_t* out, int Nr, int Nc, int L, int hlen, bool inverse,
                            const FilterBank& fb, const real_t* beta, int batch, hipStream_t s) {
    if (!swt2_tail_supported(hlen, Nr, Nc, L)) return hipErrorNotSupported;
    SwtTailArgs a;
    a.in = in; a.out = out; a.R = Nr; a.C = Nc; a.L = L; a.hlen = hlen; a.fb = fb;
    for (int l = 0; l < kSwtTailMaxLevels; l++) {
        for (int b = 0; b < 3; b++) a.det[l][b] = l < L ? det[3 * l + b] : nullptr;
        a.beta[l] = (beta && l < L) ? beta[l] : (real_t)0;
    }
    constexpr int NT = 256;
    auto lg2 = [](int v) { int lg = 0; while ((1 << lg) < v) ++lg; return (1 << lg) == v ? lg : -1; };
    const bool pow2 = lg2(Nr) >= 0 && lg2(Nc) >= 0;
    a.lgC = pow2 ? lg2(Nc) : -1;
    a.lgR = pow2 ? lg2(Nr) : -1;
    const size_t lds = swt_tail_lds_elems(Nr * Nc, inverse) * sizeof(real_t);
    static std::atomic<bool> big[4][64] = {};
    hipError_t e = hipSuccess;
#define PDWT_SWT_TAIL_GO(kernel, slot)                                         \
    e = allow_big_lds(kernel, lds, big[slot]);                                 \
    if (e != hipSuccess) return e;                                             \
    hipLaunchKernelGGL(kernel, dim3(batch), dim3(NT), lds, s, a);
    // power-of-two sizes: the mask / shift kernels; any other size: the general ones (tabulated offsets)
    if (inverse) {
        if (pow2) { PDWT_SWT_TAIL_GO((swt2_inv_tail_p2_kernel<NT>), 0) } else { PDWT_SWT_TAIL_GO((swt2_inv_tail_kernel<NT, false>), 1) }
    } else {
        if (pow2) { PDWT_SWT_TAIL_GO((swt2_fwd_tail_p2_kernel<NT>), 2) } else { PDWT_SWT_TAIL_GO((swt2_fwd_tail_kernel<NT, false>), 3) }
    }
#undef PDWT_SWT_TAIL_GO
    return hipGetLastError();
}

}  // namespace pdwt
