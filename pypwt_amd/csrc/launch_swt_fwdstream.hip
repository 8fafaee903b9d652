// launch_swt_fwdstream.hip -- launcher of the one-launch forward a-trous level (swt_fwdstream_kernels.hpp); a translation unit of its
// own: the filter lengths x dilations compile beside the other launchers.
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#include "tuning.hpp"
#include "swt_kernels_args.hpp"
#include "swt_fwdstream_kernels.hpp"

namespace pdwt {

// shortest filter on these kernels (tuning key "swt_fwdstream"; 0 = never, 100 + n = n taps at every size they take)
static std::atomic<int>& fwdstream_min() {
    static std::atomic<int> v{(int)tune::swt_fwdstream_taps};
    return v;
}
int set_swt_fwdstream_min(int taps) { return fwdstream_min().exchange(taps < 0 ? 0 : taps); }
int get_swt_fwdstream_min() { return fwdstream_min().load(std::memory_order_relaxed); }

// The fp64 library (the reference's DOUBLEPRECISION build, pdwt/src/filters.h:16-30) runs the same kernels for 6-20 taps at dilations 1-8:
// steps of 16 rows (an element is 8 bytes: 36 KB of history + 11-20 KB of staged rows at 20 taps), the 40 tap registers of fp32's 40 taps
constexpr bool kF64 = sizeof(real_t) == 8;
constexpr int kFwdStreamMaxTaps = kF64 ? 20 : kMaxTaps, kFwdStreamMaxF = kF64 ? 8 : 16;

static inline v2f mk2h(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}

template <int HLEN, int F>
static hipError_t run(const Swt2DArgs& g, int batch, hipStream_t s) {
    // dilation 8: 47-sample phases of 8 columns each -- steps of 16 rows keep the staged rows + the history at 53 KB (two workgroups per CU)
    // (dilation 16: phases of 4 columns: the staged row is 64 + 15 (hlen - 1) columns wide for 64 outputs)
    if constexpr (HLEN > kFwdStreamMaxTaps || F > kFwdStreamMaxF || (kF64 && F >= 4 && HLEN > 18)) return hipErrorNotSupported;  // (fp64, 20 taps, dilations 4 and 8: 12-44 B of scratch)
    else {
    constexpr bool kShort = F >= 8 || kF64;
    constexpr int TXC = 64, TY = kShort ? 16 : 32, NT = 256, KB = kShort ? 4 : 8, M = kShort ? 4 : 8, MINB = 2;
    using G = SwtFwdStreamGeom<HLEN, F, TXC, TY>;
    SwtFwdStreamArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.bstride = g.bstride;
    a.wk = swt_walk(g.Nr, g.Nc, F, 4);
    if (a.wk.rows_phase < TY) return hipErrorNotSupported;
    for (int j = 0; j < HLEN; ++j) a.t.t[j] = mk2h(g.fb.lo[HLEN - 1 - j], g.fb.hi[HLEN - 1 - j]);
    a.strips = cdiv(g.Nc, TXC);
    // Segments (strip_walk_seg): a segment filters hlen - 1 warm-up rows along x that it does not own, so long ones are cheaper -- as long as
    // one round of resident workgroups still covers the launch: long filters fit two workgroups per CU (512 per launch), up to 12 taps
    // (30-45 KB of LDS, < 100 VGPRs) four and more (2048^2, four levels forward, 512 -> 1024 workgroups: db4 87.0 -> 82.8 us, db5 90.5 -> 84.8;
    // db10 112 -> 119, db20 209 -> 237)
    const long long units = (long long)a.strips * a.wk.phases * batch;
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)G::LDS_REALS * sizeof(real_t);
    auto kern = swt_fwdstream_kernel<HLEN, F, TXC, TY, NT, KB, M, MINB>;
    hipError_t e = allow_big_lds(kern, lds, big);
    if (e != hipSuccess) return e;
    static std::atomic<int> slots_cache{0};
    static const int forced_slots = lab_env("PDWT_STRIP_SLOTS") ? atoi(lab_env("PDWT_STRIP_SLOTS")) : 0;  // A/B measurements
    const int slots = forced_slots > 0 ? forced_slots : resident_slots(kern, NT, lds, &slots_cache);
    a.seg = strip_walk_seg(a.wk.rows_phase, units, TY, G::W, slots);
    a.segs = cdiv(a.wk.rows_phase, a.seg);
    hipLaunchKernelGGL(kern, dim3(8 * cdiv(a.strips * a.segs * a.wk.phases, 8), batch), dim3(NT), lds, s, a);
    return hipGetLastError();
    }
}

#ifndef PDWT_FWDSTREAM_HLENS
#define PDWT_FWDSTREAM_HLENS(X) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
#endif

bool swt2_fwd_stream_takes(const Swt2DArgs& a, int batch) {
    const Tuning* at = active_tuning();
    int min_taps = at ? at->swt_fwdstream : get_swt_fwdstream_min();
    const bool forced = min_taps >= 100;
    if (forced) min_taps -= 100;
    if (min_taps <= 0 || a.hlen < min_taps || a.hlen < 6 || (a.hlen & 1) || a.hlen > kFwdStreamMaxTaps) return false;
    if ((a.f != 1 && a.f != 2 && a.f != 4 && a.f != 8 && a.f != 16) || a.f > kFwdStreamMaxF || (kF64 && a.f >= 4 && a.hlen > 18)) return false;
    if (batch < 1 || batch > 65535) return false;
    // rows that are not whole 16-B groups: the staged window of a strip may cross the row end once (swt_stage_pad)
    if ((a.Nc & 3) && a.Nc < 64 + (a.hlen - 1) * a.f + 4) return false;
    if ((long long)a.Nr * a.Nc * (long long)sizeof(real_t) >= (1LL << 32)) return false;  // 32-bit byte offsets inside a plane
    if (swt_walk(a.Nr, a.Nc, a.f, 4).rows_phase < (a.f >= 8 || kF64 ? 16 : 32)) return false;     // chains of at least one step
    return forced || (long long)batch * a.Nr * a.Nc >= (1LL << tune::swt_fwdstream_log2);
}

hipError_t try_launch_swt2_fwd_stream(const Swt2DArgs& a, int batch, hipStream_t s) {
    if (!swt2_fwd_stream_takes(a, batch)) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h)                                            \
    case h:                                             \
        if (a.f == 1) return run<h, 1>(a, batch, s);    \
        if (a.f == 2) return run<h, 2>(a, batch, s);    \
        if (a.f == 4) return run<h, 4>(a, batch, s);    \
        if (a.f == 8) return run<h, 8>(a, batch, s);    \
        return run<h, 16>(a, batch, s);
        PDWT_FWDSTREAM_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}


}  // namespace pdwt
