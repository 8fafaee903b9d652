// launch_dwt2_tail.hip -- launchers of the all-remaining-levels kernel for small approximations (dwt2_tail_kernels.hpp).
// Compiled in both builds (the kernels are written over real_t).
#include "dwt2_tail_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <atomic>
#include <cstdlib>

namespace pdwt {

static int exact_log2(int v) {
    int lg = 0;
    while ((1 << lg) < v) ++lg;
    return (1 << lg) == v ? lg : -1;
}

// (R0, C0) enter the group's finest level; K levels follow.  Even filter lengths; any sizes (the general kernels halve by
// ceil-halving like the level kernels); the LDS planes of tail_geometry fit one CU.
constexpr size_t kTailLdsBytes = 160 * 1024;
bool dwt2_tail_supported(int hlen, int R0, int C0, int K) {
    if (hlen < 2 || (hlen & 1) || hlen > kMaxTaps || K < 1 || K > kTailMaxLevels || R0 < 2 || C0 < 2) return false;
    if ((long long)R0 * C0 > kTailMaxSamples) return false;
    TailArgs a;
    a.R0 = R0; a.C0 = C0; a.K = K;
    return tail_geometry(a, false) * sizeof(real_t) <= kTailLdsBytes && tail_geometry(a, true) * sizeof(real_t) <= kTailLdsBytes;
}
// the most levels (at most Kmax) the launch can take from (R0, C0) on; 0 = none
int dwt2_tail_max_levels(int hlen, int R0, int C0, int Kmax) {
    int K = Kmax < kTailMaxLevels ? Kmax : kTailMaxLevels;
    while (K >= 1 && !dwt2_tail_supported(hlen, R0, C0, K)) --K;
    return K;
}

template <int HLEN, int NT>
static hipError_t run_tail_general(TailArgs& a, bool inverse, int batch, hipStream_t s) {
    const size_t lds = tail_geometry(a, inverse) * sizeof(real_t);
    static std::atomic<bool> big[2][64] = {};
    if (inverse) {
        const hipError_t e = allow_big_lds(dwt2_inv_tail_kernel<HLEN, NT>, lds, big[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_inv_tail_kernel<HLEN, NT>), dim3(batch), dim3(NT), lds, s, a);
    } else {
        const hipError_t e = allow_big_lds(dwt2_fwd_tail_kernel<HLEN, NT>, lds, big[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_fwd_tail_kernel<HLEN, NT>), dim3(batch), dim3(NT), lds, s, a);
    }
    return hipGetLastError();
}
template <int HLEN, int NT>
static hipError_t run_tail_p2(const TailArgs& a, bool inverse, int batch, hipStream_t s) {
    const size_t lds = tail_lds_elems(a.R0 * a.C0) * sizeof(real_t);
    static std::atomic<bool> big[2][64] = {};
    if (inverse) {
        const hipError_t e = allow_big_lds(dwt2_inv_tail_p2_kernel<HLEN, NT>, lds, big[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_inv_tail_p2_kernel<HLEN, NT>), dim3(batch), dim3(NT), lds, s, a);
    } else {
        const hipError_t e = allow_big_lds(dwt2_fwd_tail_p2_kernel<HLEN, NT>, lds, big[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_fwd_tail_p2_kernel<HLEN, NT>), dim3(batch), dim3(NT), lds, s, a);
    }
    return hipGetLastError();
}

// a thread stages kTailTrips values: 1024 threads for planes of more than 4096 samples, 256 below (fewer idle wavefronts at
// every barrier of the small levels), ONE wavefront per image for the tiniest images of a large batch (no barrier waits at all and
// four times the images in flight per CU; PDWT_TAIL_WAVE_MAX = largest such image in samples, A/B measurements).
// Power-of-two sizes: the mask / shift kernels; any other size: the general ones.
template <int HLEN>
static hipError_t run_tail_nt(TailArgs& a, bool inverse, int batch, hipStream_t s) {
    static const int wave_max = lab_env("PDWT_TAIL_WAVE_MAX") ? atoi(lab_env("PDWT_TAIL_WAVE_MAX")) : 1024;
    const int n0 = a.R0 * a.C0;
    const bool big = n0 > 4096, pow2 = a.lgR >= 0 && a.lgC >= 0;
    // ... once the batch alone fills the chip's 8192 wavefront slots (earlier for images of at most 256 samples, which leave three of
    // a workgroup's four wavefronts idle): forward+inverse us, 256 threads -> one wavefront: 65536 x 16^2 haar L4 240 -> 101,
    // 262144 x 8^2 haar L3 790 -> 267, 20000 x 28^2 db2 L3 154 -> 132, 16384 x 32^2 db2 L3 83 -> 73, 5000 x 12x20 haar L2 26 -> 17;
    // but 1200 x 24x40 db2 L2 17 -> 24 (profiles/r04zr_tail_one_wavefront.txt)
    const bool wave = n0 <= wave_max && n0 <= 64 * kTailTrips && (batch >= 8192 || (n0 <= 256 && batch >= 2048));
    if (pow2) return big ? run_tail_p2<HLEN, 1024>(a, inverse, batch, s) : wave ? run_tail_p2<HLEN, 64>(a, inverse, batch, s) : run_tail_p2<HLEN, 256>(a, inverse, batch, s);
    return big ? run_tail_general<HLEN, 1024>(a, inverse, batch, s) : wave ? run_tail_general<HLEN, 64>(a, inverse, batch, s) : run_tail_general<HLEN, 256>(a, inverse, batch, s);
}

// forward: in = A_{l-1} -> det[3 k + b] (band b of the group's k-th level, finest first), out = A_{l-1+K}
// inverse: in = A_{l-1+K}, det as above -> out = A_{l-1}
hipError_t launch_dwt2_tail(const real_t* in, real_t* const* det, real_t* out, int R0, int C0, int K, int hlen, bool inverse,
                            const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_tail_supported(hlen, R0, C0, K)) return hipErrorNotSupported;
    TailArgs a;
    a.in = in; a.out = out; a.R0 = R0; a.C0 = C0; a.lgR = exact_log2(R0); a.lgC = exact_log2(C0); a.K = K; a.hlen = hlen; a.fb = fb;
    tail_geometry(a, inverse);  // (every field set, whichever kernel runs)
    for (int k = 0; k < kTailMaxLevels; k++)
        for (int b = 0; b < 3; b++) a.det[k][b] = k < K ? det[3 * k + b] : nullptr;
    switch (hlen) {
        case 2: return run_tail_nt<2>(a, inverse, batch, s);
        case 4: return run_tail_nt<4>(a, inverse, batch, s);
        case 6: return run_tail_nt<6>(a, inverse, batch, s);
        case 8: return run_tail_nt<8>(a, inverse, batch, s);
    }
    return run_tail_nt<0>(a, inverse, batch, s);  // run-time filter length
}

}  // namespace pdwt
