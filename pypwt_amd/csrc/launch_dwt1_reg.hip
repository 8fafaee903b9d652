// launch_dwt1_reg.hip -- launchers of the register-resident 1D DWT kernels (dwt1_reg_kernels.hpp): up to three
// levels per launch, lane shifts instead of LDS.  Compiled in both builds (the kernels are written over real_t).
#include "dwt1_reg_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <cstdlib>

namespace pdwt {

static void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}

// ---- up to three levels per launch in registers (dwt1_reg_kernels.hpp)
bool dwt1_reg_supported(int hlen, int N0, int K) {
    const int q = (4 << K) > 16 ? (4 << K) : 16;  // 16-B loads of whole lanes; every store unit inside its row
    static const int nmin = [] { const char* e = lab_env("PDWT_REG1_MIN"); return e ? atoi(e) : 2048; }();  // tuning
    return !(hlen & 1) && hlen >= 2 && hlen <= kReg1MaxHlen && K >= 1 && K <= kReg1MaxLevels && N0 >= nmin &&
           (N0 % q) == 0 && (long long)N0 * (long long)sizeof(real_t) < (1LL << 31);  // row byte offsets below kReg1Dropped
}

static int reg1_blocks_per_wave(long long total_blocks) {
    // Measured on 2^24 samples (18397 blocks; profiles/r02u_*): 1 / 2 / 5 / 9 / 18 blocks per wavefront = 27.0 / 27.5 / 26.7 /
    // 25.1 / 25.1 us forward and 32.1 (5) / 25.8 (9) / 25.3 (18) us inverse -- long runs keep the prefetch pipeline
    // full -- while the 2300 blocks of the second launch (levels 4-6) take 9.0 us at 1, 12.3 at 9 and 18.4 at 18:
    // a launch that small wants every SIMD busy.  Hence: about 2048 wavefronts, one block each below 4096 blocks.
    // PDWT_REG1_BPW overrides (tuning).
    static const int forced = [] { const char* e = lab_env("PDWT_REG1_BPW"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    if (total_blocks <= 4096) return 1;
    const long long b = (total_blocks + 2047) / 2048;
    return (int)(b > 64 ? 64 : b);
}

template <int HLEN, int K>
static hipError_t run_fwd_reg(Fwd1DRegArgs& a, hipStream_t s) {
    constexpr int NT = 256;
    reg1_fwd_blocks(HLEN, K, a.N0, &a.nblk, &a.nplain);
    a.bpw = reg1_blocks_per_wave((long long)a.nblk * a.rows);
    a.wpr = cdiv(a.nblk, a.bpw);
    const long long waves = (long long)a.wpr * a.rows;
    hipLaunchKernelGGL((dwt1_fwd_reg_kernel<HLEN, K, NT>), dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_dwt1_fwd_reg(const real_t* in, real_t* const* det, real_t* app, int rows, int N0, int K, int hlen,
                               const FilterBank& fb, hipStream_t s) {
    if (!dwt1_reg_supported(hlen, N0, K)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(app) & 15)) return hipErrorNotSupported;
    Fwd1DRegArgs a;
    a.in = in; a.app = app; a.rows = rows; a.N0 = N0;
    for (int k = 0; k < kReg1MaxLevels; k++) {
        a.det[k] = k < K ? det[k] : nullptr;
        if (k < K && (reinterpret_cast<uintptr_t>(det[k]) & 15)) return hipErrorNotSupported;
    }
    interleave(a.fb, fb);
#define Y(h, k) if (hlen == h && K == k) return run_fwd_reg<h, k>(a, s);
#define X(h) Y(h, 1) Y(h, 2) Y(h, 3)
    X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
#undef Y
    return hipErrorNotSupported;
}

template <int HLEN, int K>
static hipError_t run_inv_reg(Inv1DRegArgs& a, hipStream_t s) {
    constexpr int NT = 256;
    reg1_inv_blocks(HLEN, K, a.N0, &a.nblk, &a.nplain);
    a.bpw = reg1_blocks_per_wave((long long)a.nblk * a.rows);
    a.wpr = cdiv(a.nblk, a.bpw);
    const long long waves = (long long)a.wpr * a.rows;
    hipLaunchKernelGGL((dwt1_inv_reg_kernel<HLEN, K, NT>), dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_dwt1_inv_reg(const real_t* app, const real_t* const* det, real_t* out, int rows, int N0, int K, int hlen,
                               const FilterBank& fb, hipStream_t s) {
    if (!dwt1_reg_supported(hlen, N0, K)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(app) & 15)) return hipErrorNotSupported;
    Inv1DRegArgs a;
    a.app = app; a.out = out; a.rows = rows; a.N0 = N0;
    for (int k = 0; k < kReg1MaxLevels; k++) {
        a.det[k] = k < K ? det[k] : nullptr;
        if (k < K && (reinterpret_cast<uintptr_t>(det[k]) & 15)) return hipErrorNotSupported;
    }
    interleave(a.fb, fb);
#define Y(h, k) if (hlen == h && K == k) return run_inv_reg<h, k>(a, s);
#define X(h) Y(h, 1) Y(h, 2) Y(h, 3)
    X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
#undef Y
    return hipErrorNotSupported;
}

}  // namespace pdwt
