// launch_swt_split.hip -- launchers of the two-launch a-trous level (swt_split_kernels.hpp).
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "launch.hpp"
#include "launch_util.hpp"
#ifndef PDWT_DOUBLE
#include "swt_split_kernels.hpp"
#endif

namespace pdwt {

#ifdef PDWT_DOUBLE
bool swt2_split_supported(int, int, int, int, bool) { return false; }
int set_swt_split_min(int, int) { return 0; }
hipError_t launch_swt2_split(const Swt2DArgs&, real_t*, bool, int, hipStream_t) { return hipErrorNotSupported; }
#else

// filter lengths the split kernels are built for
#define PDWT_SPLIT_HLENS(X) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)

static inline v2f mk2h(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Where the two launches beat the LDS-tiled level kernel (2048^2 levels, tools/swtsweep.py, profiles/r03_swt_split_sweep.txt):
// the inverse from 12 taps on (12 taps 80 -> 47 us per level, 16 taps 93 -> 51, 40 taps 150-290 -> 72-87; at 10 taps both
// take 45), the forward from 18 taps on (18 taps 47-61 -> 41, 26 taps 91-111 -> 43, 40 taps 127-158 -> 50; at 16 taps the
// tiled kernel's 35-42 is level with the 40 of two launches).  Tuning keys "swt_split_fwd" / "swt_split_inv" (environment
// PDWT_SWT_SPLIT_FWD / _INV): the shortest filter that takes this path, 0 = never.
static std::atomic<int>& split_min(bool inverse) {
    static std::atomic<int> fwd{env_int("PDWT_SWT_SPLIT_FWD", 18)}, inv{env_int("PDWT_SWT_SPLIT_INV", 12)};
    return inverse ? inv : fwd;
}
int set_swt_split_min(int inverse, int taps) { return split_min(inverse != 0).exchange(taps < 0 ? 0 : taps); }

bool swt2_split_supported(int hlen, int Nr, int Nc, int f, bool inverse) {
    const int min_taps = split_min(inverse).load(std::memory_order_relaxed);
    if (min_taps <= 0 || (hlen & 1) || hlen < 10 || hlen > kMaxTaps || hlen < min_taps) return false;
    if ((Nc & 3) || f < 1 || f >= Nr || f >= Nc || Nc < 16) return false;
    if (f != 1 && f != 2 && (f & 3)) return false;
    return true;
}

template <int NT, typename K>
static hipError_t go(K kernel, const SwtSplitArgs& a, long long waves, hipStream_t s) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)cdivll(waves, NT / 64)), dim3(NT), 0, s, a);
    return hipGetLastError();
}

template <int HLEN>
static hipError_t run_split(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    constexpr int NT = 256, NTC = 1024, R = 4;  // NTC: wavefronts of a column workgroup share their rows (split_col_work)
    const long long plane = (long long)a.Nr * a.Nc;
    SwtSplitArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = batch;
    k.soft_beta = a.soft_beta;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2h(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    const int f = a.f;
    const long long col_items = split_col_waves(batch, a.Nr, a.Nc, f, R);
    const long long row_items4 = f >= 4 ? split_row_waves(batch, a.Nr, split_row_items4(a.Nc, f, R)) : 0;
    const long long row_items1 = split_row_waves(batch, a.Nr, split_row_items1(a.Nc));
    hipError_t e;
    if (!inverse) {
        SwtSplitArgs r = k;  // in -> lo, hi (scratch: two planes per image)
        r.in[0] = a.in; r.in_bstride = a.bstride;
        r.out[0] = tmp; r.out[1] = tmp + plane; r.out_bstride = 2 * plane;
        if (f == 1) e = go<NT>(swt_row_fwd1_kernel<HLEN, 1, NT>, r, row_items1, s);
        else if (f == 2) e = go<NT>(swt_row_fwd1_kernel<HLEN, 2, NT>, r, row_items1, s);
        else e = go<NT>(swt_row_fwd4_kernel<HLEN, R, NT>, r, row_items4, s);
        if (e != hipSuccess) return e;
        SwtSplitArgs c = k;
        c.in[0] = tmp; c.in[1] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0] = a.A; c.out[1] = a.H; c.out[2] = a.V; c.out[3] = a.D; c.out_bstride = a.bstride;
        return go<NTC>(swt_col_fwd_kernel<HLEN, R, NTC>, c, col_items, s);
    }
    SwtSplitArgs c = k;  // A, H, V, D -> interleaved (L', H') (scratch: two planes per image)
    c.in[0] = a.A; c.in[1] = a.H; c.in[2] = a.V; c.in[3] = a.D; c.in_bstride = a.bstride;
    c.out[0] = tmp; c.out_bstride = 2 * plane;
    e = go<NTC>(swt_col_inv_kernel<HLEN, R, NTC>, c, col_items, s);
    if (e != hipSuccess) return e;
    SwtSplitArgs r = k;
    r.in[0] = tmp; r.in_bstride = 2 * plane;
    r.out[0] = a.out; r.out_bstride = a.bstride;
    if (f == 1) return go<NT>(swt_row_inv1_kernel<HLEN, 1, NT>, r, row_items1, s);
    if (f == 2) return go<NT>(swt_row_inv1_kernel<HLEN, 2, NT>, r, row_items1, s);
    return go<NT>(swt_row_inv4_kernel<HLEN, R, NT>, r, row_items4, s);
}

// scratch: 2 * Nr * Nc * batch elements, 16-B aligned
hipError_t launch_swt2_split(const Swt2DArgs& a, real_t* tmp, bool inverse, int batch, hipStream_t s) {
    if (!swt2_split_supported(a.hlen, a.Nr, a.Nc, a.f, inverse) || !tmp) return hipErrorNotSupported;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(tmp) || !al16(a.A) || !al16(a.H) || !al16(a.V) || !al16(a.D) || (a.bstride & 3)) return hipErrorNotSupported;
    if (!al16(inverse ? (const void*)a.out : (const void*)a.in)) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) \
    case h:  \
        return run_split<h>(a, tmp, inverse, batch, s);
        PDWT_SPLIT_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}
#endif

}  // namespace pdwt
