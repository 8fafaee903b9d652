// launch_swt_tail.hip -- launchers of the whole-transform-per-workgroup SWT kernels for batches of tiny images
// (swt2_tail_kernels.hpp).  Compiled in both builds (the kernels are written over real_t).
#include "launch.hpp"
#include "launch_util.hpp"
#include "swt2_tail_kernels.hpp"

#include <atomic>

namespace pdwt {

static int exact_log2_swt(int v) {
    int lg = 0;
    while ((1 << lg) < v) ++lg;
    return (1 << lg) == v ? lg : -1;
}

// power-of-two images of at most 4096 samples, any filter length, at most 12 levels
bool swt2_tail_supported(int hlen, int Nr, int Nc, int L) {
    if (hlen < 1 || hlen > kMaxTaps || L < 1 || L > kSwtTailMaxLevels || Nr < 2 || Nc < 2) return false;
    if ((long long)Nr * Nc > kSwtTailMaxSamples) return false;
    return exact_log2_swt(Nr) >= 0 && exact_log2_swt(Nc) >= 0;
}

// forward: in = the images -> det[3 (l - 1) + b] = band b of level l, out = A_L
// inverse: in = A_L, det as above (beta[l - 1]: soft threshold applied to level l's details as they are read) -> out = the images
hipError_t launch_swt2_tail(const real_t* in, real_t* const* det, real_t* out, int Nr, int Nc, int L, int hlen, bool inverse,
                            const FilterBank& fb, const real_t* beta, int batch, hipStream_t s) {
    if (!swt2_tail_supported(hlen, Nr, Nc, L)) return hipErrorNotSupported;
    SwtTailArgs a;
    a.in = in; a.out = out; a.lgR = exact_log2_swt(Nr); a.lgC = exact_log2_swt(Nc); a.L = L; a.hlen = hlen; a.fb = fb;
    for (int l = 0; l < kSwtTailMaxLevels; l++) {
        for (int b = 0; b < 3; b++) a.det[l][b] = l < L ? det[3 * l + b] : nullptr;
        a.beta[l] = (beta && l < L) ? beta[l] : (real_t)0;
    }
    constexpr int NT = 256;
    const size_t lds = swt_tail_lds_elems(Nr * Nc, inverse) * sizeof(real_t);
    static std::atomic<bool> big[2][64] = {};
    if (inverse) {
        const hipError_t e = allow_big_lds(swt2_inv_tail_kernel<NT>, lds, big[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_inv_tail_kernel<NT>), dim3(batch), dim3(NT), lds, s, a);
    } else {
        const hipError_t e = allow_big_lds(swt2_fwd_tail_kernel<NT>, lds, big[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((swt2_fwd_tail_kernel<NT>), dim3(batch), dim3(NT), lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace pdwt
