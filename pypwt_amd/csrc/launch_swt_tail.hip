// launch_swt_tail.hip -- launchers of the whole-transform-per-workgroup SWT kernels for batches of tiny images
// (swt2_tail_kernels.hpp).  Compiled in both builds (the kernels are written over real_t).
#include "launch.hpp"
#include "launch_util.hpp"
#include "swt2_tail_kernels.hpp"

#include <atomic>
#include <cstdlib>

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
    auto lg2 = [](int v) { int lg = 0; while ((1 << lg) < v) ++lg; return (1 << lg) == v ? lg : -1; };
    const bool pow2 = lg2(Nr) >= 0 && lg2(Nc) >= 0;
    a.lgC = pow2 ? lg2(Nc) : -1;
    a.lgR = pow2 ? lg2(Nr) : -1;
    const int n = Nr * Nc;
    const size_t lds = swt_tail_lds_elems(n, inverse) * sizeof(real_t);
    // 256 threads per image; ONE wavefront per image for large batches of images of at most 256 samples (no barrier waits, four
    // times the images in flight) -- forward+inverse us: 65536 x 16^2 haar L3 737 -> 413, 262144 x 8^2 haar L2 1731 -> 535, 5000 x
    // 12x20 haar L2 57 -> 35; images of 784-1024 samples lose (every SWT level is full size: 16384 x 32^2 haar L3 401 -> 511,
    // profiles/r04zr_swt_tail_one_wavefront.txt).  PDWT_TAIL_WAVE_MAX = largest such image in samples (A/B measurements).
    static const int wave_max = lab_env("PDWT_TAIL_WAVE_MAX") ? atoi(lab_env("PDWT_TAIL_WAVE_MAX")) : kSwtTailWaveSamples;
    const bool wave = n <= wave_max && n <= kSwtTailWaveSamples && batch >= 2048;
    const bool few = n <= 1024;  // 256 threads x 4 trips
    static std::atomic<bool> big[12][64] = {};
    hipError_t e = hipSuccess;
#define PDWT_SWT_TAIL_GO(kernel, slot, NT)                                     \
    e = allow_big_lds(kernel, lds, big[slot]);                                 \
    if (e != hipSuccess) return e;                                             \
    hipLaunchKernelGGL(kernel, dim3(batch), dim3(NT), lds, s, a);
    // power-of-two sizes: the mask / shift kernels; any other size: the general ones (tabulated offsets)
    if (wave) {
        if (inverse) {
            if (pow2) { PDWT_SWT_TAIL_GO((swt2_inv_tail_p2_kernel<64, 4>), 4, 64) } else { PDWT_SWT_TAIL_GO((swt2_inv_tail_kernel<64, false, 4>), 5, 64) }
        } else {
            if (pow2) { PDWT_SWT_TAIL_GO((swt2_fwd_tail_p2_kernel<64, 4>), 6, 64) } else { PDWT_SWT_TAIL_GO((swt2_fwd_tail_kernel<64, false, 4>), 7, 64) }
        }
    } else if (few) {
        if (inverse) {
            if (pow2) { PDWT_SWT_TAIL_GO((swt2_inv_tail_p2_kernel<256, 4>), 8, 256) } else { PDWT_SWT_TAIL_GO((swt2_inv_tail_kernel<256, false, 4>), 9, 256) }
        } else {
            if (pow2) { PDWT_SWT_TAIL_GO((swt2_fwd_tail_p2_kernel<256, 4>), 10, 256) } else { PDWT_SWT_TAIL_GO((swt2_fwd_tail_kernel<256, false, 4>), 11, 256) }
        }
    } else if (inverse) {
        if (pow2) { PDWT_SWT_TAIL_GO((swt2_inv_tail_p2_kernel<256, 16>), 0, 256) } else { PDWT_SWT_TAIL_GO((swt2_inv_tail_kernel<256, false, 16>), 1, 256) }
    } else {
        if (pow2) { PDWT_SWT_TAIL_GO((swt2_fwd_tail_p2_kernel<256, 16>), 2, 256) } else { PDWT_SWT_TAIL_GO((swt2_fwd_tail_kernel<256, false, 16>), 3, 256) }
    }
#undef PDWT_SWT_TAIL_GO
    return hipGetLastError();
}

}  // namespace pdwt
