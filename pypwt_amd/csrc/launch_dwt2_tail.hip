// launch_dwt2_tail.hip -- launchers of the all-remaining-levels kernel for small approximations (dwt2_tail_kernels.hpp).
// Compiled in both builds (the kernels are written over real_t).
#include "dwt2_tail_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <atomic>

namespace pdwt {

static int exact_log2(int v) {
    int lg = 0;
    while ((1 << lg) < v) ++lg;
    return (1 << lg) == v ? lg : -1;
}

// (R0, C0) enter the group's finest level; K levels follow.  Even filter lengths; every level's input has even sizes (so each
// level halves both sizes exactly); the two LDS planes of the first level fit one CU.
bool dwt2_tail_supported(int hlen, int R0, int C0, int K) {
    if (hlen < 2 || (hlen & 1) || hlen > kMaxTaps || K < 1 || K > kTailMaxLevels || R0 < 2 || C0 < 2) return false;
    if ((long long)R0 * C0 > kTailMaxSamples) return false;
    for (int k = 0; k < K; k++)
        if (((R0 >> k) & 1) || ((C0 >> k) & 1) || (R0 >> k) < 2 || (C0 >> k) < 2) return false;
    return true;
}
// the most levels (at most Kmax) the launch can take from (R0, C0) on; 0 = none
int dwt2_tail_max_levels(int hlen, int R0, int C0, int Kmax) {
    int K = Kmax < kTailMaxLevels ? Kmax : kTailMaxLevels;
    while (K >= 1 && !dwt2_tail_supported(hlen, R0, C0, K)) --K;
    return K;
}

template <int HLEN, int NT>
static hipError_t run_tail_general(const TailArgs& a, bool inverse, int batch, hipStream_t s) {
    const size_t lds = tail_lds_elems(a.R0 * a.C0) * sizeof(real_t);
    static std::atomic<bool> big[2][64] = {};
    if (inverse) {
        const hipError_t e = allow_big_lds(dwt2_inv_tail_kernel<HLEN, NT, false>, lds, big[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_inv_tail_kernel<HLEN, NT, false>), dim3(batch), dim3(NT), lds, s, a);
    } else {
        const hipError_t e = allow_big_lds(dwt2_fwd_tail_kernel<HLEN, NT, false>, lds, big[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwt2_fwd_tail_kernel<HLEN, NT, false>), dim3(batch), dim3(NT), lds, s, a);
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
// every barrier of the small levels).  Power-of-two sizes: the mask / shift kernels; any other size: the general ones.
template <int HLEN>
static hipError_t run_tail_nt(const TailArgs& a, bool inverse, int batch, hipStream_t s) {
    const bool big = a.R0 * a.C0 > 4096, pow2 = a.lgR >= 0 && a.lgC >= 0;
    if (pow2) return big ? run_tail_p2<HLEN, 1024>(a, inverse, batch, s) : run_tail_p2<HLEN, 256>(a, inverse, batch, s);
    return big ? run_tail_general<HLEN, 1024>(a, inverse, batch, s) : run_tail_general<HLEN, 256>(a, inverse, batch, s);
}

// forward: in = A_{l-1} -> det[3 k + b] (band b of the group's k-th level, finest first), out = A_{l-1+K}
// inverse: in = A_{l-1+K}, det as above -> out = A_{l-1}
hipError_t launch_dwt2_tail(const real_t* in, real_t* const* det, real_t* out, int R0, int C0, int K, int hlen, bool inverse,
                            const FilterBank& fb, int batch, hipStream_t s) {
    if (!dwt2_tail_supported(hlen, R0, C0, K)) return hipErrorNotSupported;
    TailArgs a;
    a.in = in; a.out = out; a.R0 = R0; a.C0 = C0; a.lgR = exact_log2(R0); a.lgC = exact_log2(C0); a.K = K; a.hlen = hlen; a.fb = fb;
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
