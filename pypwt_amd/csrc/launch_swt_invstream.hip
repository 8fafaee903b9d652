// launch_swt_invstream.hip -- launcher of the one-launch inverse a-trous level (swt_invstream_kernels.hpp); a translation unit of its
// own: the filter lengths x dilations compile beside the other launchers.
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#include "tuning.hpp"
#include "swt_kernels_args.hpp"
#include "swt_invstream_kernels.hpp"

namespace pdwt {

// shortest filter on these kernels (tuning key "swt_invstream"; 0 = never, 100 + n = n taps at every size they take)
static std::atomic<int>& invstream_min() {
    static std::atomic<int> v{(int)tune::swt_invstream_taps};
    return v;
}
int set_swt_invstream_min(int taps) { return invstream_min().exchange(taps < 0 ? 0 : taps); }
int get_swt_invstream_min() { return invstream_min().load(std::memory_order_relaxed); }

// The fp64 library runs the same kernels for 6-16 taps at dilations 1-4, in steps of 16 rows (launch_swt_fwdstream.hip)
constexpr bool kF64 = sizeof(real_t) == 8;
constexpr int kInvStreamMaxTaps = kF64 ? 16 : 28, kInvStreamMaxF = kF64 ? 4 : 8;

static inline v2f mk2h(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}

template <int HLEN, int F>
static hipError_t run(const Swt2DArgs& g, int batch, hipStream_t s) {
    // the staged rows are pairs (8 B per sample): dilations 4 and 8 walk in steps of 16 rows (two workgroups per CU)
    if constexpr (HLEN > kInvStreamMaxTaps || F > kInvStreamMaxF) return hipErrorNotSupported;
    else {
    constexpr bool kShort = F >= 4 || (F == 2 && HLEN > 36) || kF64;
    constexpr int TXC = 64, TY = kShort ? 16 : 32, NT = 256, KB = kShort ? 4 : 8, M = kShort ? 4 : 8, MINB = 2;
    using G = SwtInvStreamGeom<HLEN, F, TXC, TY>;
    SwtInvStreamArgs a;
    a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D; a.out = g.out;
    a.Nr = g.Nr; a.Nc = g.Nc; a.bstride = g.bstride;
    a.soft_beta = g.soft_beta;
    a.wk = swt_walk(g.Nr, g.Nc, F, 4);
    if (a.wk.rows_phase < TY) return hipErrorNotSupported;
    for (int j = 0; j < HLEN; ++j) a.t.t[j] = mk2h(g.fb.lo[HLEN - 1 - j], g.fb.hi[HLEN - 1 - j]);
    a.strips = cdiv(g.Nc, TXC);
    // Segments as in launch_swt_fwdstream.hip (strip_walk_seg)
    const long long units = (long long)a.strips * a.wk.phases * batch;
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)G::LDS_REALS * sizeof(real_t);
    auto kern = swt_invstream_kernel<HLEN, F, TXC, TY, NT, KB, M, MINB>;
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

// Built for 6-28 taps.  With the first LDS layout the two launches won from 22 taps on (and at dilation 8 from 18); with the lanes on consecutive rows and
// conflict-free pitches (swt_fwdstream_kernels.hpp) one launch wins up to 28 taps -- 2048^2 per level db11 39-49 -> 34-45 us, db13 / db14 41-52 -> 35-50,
// 1024^2 db11 21 -> 15-17, 4096^2 db13 153 -> 120 -- and loses from 32 (db16 45-56 -> 50-65): profiles/r06_swt_invstream.txt.  The row synthesis here is
// the 80-FMA pass, fed by 8-B LDS reads, and the loads of four planes in flight beside the tap registers spill at 40 taps
#ifndef PDWT_INVSTREAM_HLENS
#define PDWT_INVSTREAM_HLENS(X) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28)
#endif

bool swt2_inv_stream_takes(const Swt2DArgs& a, int batch) {
    const Tuning* at = active_tuning();
    int min_taps = at ? at->swt_invstream : get_swt_invstream_min();
    const bool forced = min_taps >= 100;
    if (forced) min_taps -= 100;
    if (min_taps <= 0 || a.hlen < min_taps || a.hlen < 6 || (a.hlen & 1) || a.hlen > kMaxTaps) return false;
    if (a.f != 1 && a.f != 2 && a.f != 4 && a.f != 8) return false;
    if (batch < 1 || batch > 65535) return false;
    // rows that are not whole 16-B groups: the staged window of a strip may cross the row end once (swt_stage_pad)
    if ((a.Nc & 3) && a.Nc < 64 + (a.hlen - 1) * a.f + 4) return false;
    if ((long long)a.Nr * a.Nc * (long long)sizeof(real_t) >= (1LL << 32)) return false;  // 32-bit byte offsets inside a plane
    if (a.hlen > kInvStreamMaxTaps || a.f > kInvStreamMaxF || (a.hlen > tune::swt_invstream_max_taps && !forced) || (a.f == 8 && a.hlen > tune::swt_invstream_f8_max_taps && !forced)) return false;
    if (swt_walk(a.Nr, a.Nc, a.f, 4).rows_phase < (a.f >= 4 || kF64 ? 16 : 32)) return false;     // chains of at least one step
    return forced || (long long)batch * a.Nr * a.Nc >= (1LL << tune::swt_invstream_log2);
}

hipError_t try_launch_swt2_inv_stream(const Swt2DArgs& a, int batch, hipStream_t s) {
    if (!swt2_inv_stream_takes(a, batch)) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h)                                            \
    case h:                                             \
        if (a.f == 1) return run<h, 1>(a, batch, s);    \
        if (a.f == 2) return run<h, 2>(a, batch, s);    \
        if (a.f == 4) return run<h, 4>(a, batch, s);    \
        return run<h, 8>(a, batch, s);
        PDWT_INVSTREAM_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}


}  // namespace pdwt
