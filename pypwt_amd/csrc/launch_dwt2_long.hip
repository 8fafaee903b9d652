// launch_dwt2_long.hip -- launchers of the strip-streaming 2D DWT level kernels for long filters (dwt2_long_kernels.hpp).
//
// try_launch_* return hipErrorNotSupported when the level does not meet the kernels' preconditions; the caller then falls
// back to the LDS tiles.
//
// Shapes (tools/longbench.hip on MI355X, profiles/r06_longbench_*.txt): a strip of 64 coefficient columns, 16 coefficient
// rows per step, 256 threads; forward: 4 output columns per row item, 4 output rows per column item; inverse: 4 coefficient
// columns per row item, 8 output row pairs per column item.  56-58 KB of LDS at 40 taps: two workgroups per CU, up to 256 VGPRs
// each, which is what lets the tap tables live in vector registers.
// The fp64 library: strips of 32 columns and 128 threads (an element is 8 bytes: 61-64 KB of LDS at 40 taps, two workgroups of two
// wavefronts per CU -- one wavefront per SIMD with up to 512 registers, of which the tap tables take 160-168).
#include "launch.hpp"
#include "launch_util.hpp"

#include "dwt2_long_kernels.hpp"

namespace pdwt {

static v2f mk2_host(real_t x, real_t y) {
    v2f r;
    r.x = x;
    r.y = y;
    return r;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(real_t) - 1)) == 0; }  // a 4-element group

constexpr bool kLongF64 = sizeof(real_t) == 8;
constexpr int kLongTXC = kLongF64 ? 32 : 64, kLongTY = 16, kLongNT = kLongF64 ? 128 : 256, kLongMinB = kLongF64 ? 1 : 2;

// Rows per segment: the walk of a workgroup re-filters the D history rows of its segment (hlen - 2 input rows forward,
// hlen / 2 coefficient rows inverse), so long segments are cheaper -- but the chip wants two workgroups per CU (512): one
// 4096^2 level is 32 strips x 16 segments of 128 rows, a batch of 16 images one segment per strip.
// (strip_walk_seg, launch_util.hpp: the whole number of steps with the fewest rounds of 512 resident workgroups x steps per workgroup; the first
// version rounded ceil(rows / segments) up to whole steps, which halves the workgroups just past a power of two)
static int long_seg(int rows, int strips, int batch, int hint, int ty, int warm) {
    if (hint > 0) return cdiv(hint, ty) * ty;
    return strip_walk_seg(rows, (long long)strips * (batch > 0 ? batch : 1), ty, warm, 512);
}

template <int HLEN>
static hipError_t run_fwd(const Fwd2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    // fp64, 40 taps: 160 tap registers + the staged rows in flight leave 68 B of scratch and no gain over the tiles
    // (profiles/r06_f64_long_ab.txt: 269 against 263 us): not built, the library holds no kernel that spills
    if constexpr (kLongF64 && HLEN > 38) return hipErrorNotSupported;
    else {
    // fp64: smaller register blocks, and from 32 taps on steps of 8 output rows (half the staged groups in flight): 40 taps keep 68 B of
    // scratch (16-row steps: 296 B and 608 us per 4096^2 L3 forward against 266 on the tiles)
    constexpr int TXC = kLongTXC, TY = kLongF64 && HLEN >= 32 ? 8 : kLongTY, NT = kLongNT, KB = kLongF64 ? 2 : 4, M = kLongF64 ? 2 : 4, XB = 1,
                  MINB = kLongMinB;
    using G = FwdLongGeom<HLEN, TXC, TY>;
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)G::LDS_REALS * sizeof(real_t);
    auto kern = dwt2_fwd_long_kernel<HLEN, TXC, TY, NT, KB, M, XB, MINB>;
    hipError_t e = allow_big_lds(kern, lds, big);
    if (e != hipSuccess) return e;
    if (g.Nr < G::R2) return hipErrorNotSupported;  // the staged rows advance by R2 with ONE conditional wrap
    FwdLongArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.Nr2 = g.Nr2; a.Nc2 = g.Nc2;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Nc2, TXC);
    a.seg = long_seg(g.Nr2, a.strips, batch, seg_hint, TY, G::W);
    a.segs = cdiv(g.Nr2, a.seg);
    for (int i = 0; i < kMaxTaps; i++) a.fb.t[i] = mk2_host(g.fb.lo[i], g.fb.hi[i]);
    hipLaunchKernelGGL(kern, dim3(8 * cdiv(a.strips * a.segs, 8), batch), dim3(NT), lds, s, a);
    return hipGetLastError();
    }
}

template <int HLEN>
static hipError_t run_inv(const Inv2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    constexpr int TXC = kLongTXC, TY = kLongTY, NT = kLongNT, KB = 4, M = 8, XB = 1, MINB = kLongMinB;
    using G = InvLongGeom<HLEN, TXC, TY>;
    static std::atomic<bool> big[64] = {};
    constexpr size_t lds = (size_t)G::LDS_REALS * sizeof(real_t);
    auto kern = dwt2_inv_long_kernel<HLEN, TXC, TY, NT, KB, M, XB, MINB>;
    hipError_t e = allow_big_lds(kern, lds, big);
    if (e != hipSuccess) return e;
    if (g.Nrc < TY) return hipErrorNotSupported;
    InvLongArgs a;
    a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D; a.out = g.out;
    a.Nrc = g.Nrc; a.Ncc = g.Ncc; a.Nr = g.Nr; a.Nc = g.Nc;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Ncc, TXC);
    a.seg = long_seg(g.Nrc, a.strips, batch, seg_hint, TY, G::W);
    a.segs = cdiv(g.Nrc, a.seg);
    long_syn_tables<HLEN>(a, g.fb.lo, g.fb.hi);
    hipLaunchKernelGGL(kern, dim3(8 * cdiv(a.strips * a.segs, 8), batch), dim3(NT), lds, s, a);
    return hipGetLastError();
}

#define PDWT_LONG_HLENS(X) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)

// 32-bit byte offsets inside a plane (buffer stores with a scalar row offset)
static bool plane_fits(long long rows, long long cols) { return rows * cols * (long long)sizeof(real_t) < (1LL << 32); }

hipError_t try_launch_dwt2_fwd_long(const Fwd2DArgs& a, int batch, hipStream_t s, int seg_hint) {
    if ((a.hlen & 1) || a.hlen < kLongMinHlen || a.hlen > kMaxTaps) return hipErrorNotSupported;
    if ((a.Nr & 1) || (a.Nc & 3) || (a.in_bstride & 3) || a.Nc2 * 2 != a.Nc || a.Nr2 * 2 != a.Nr) return hipErrorNotSupported;
    if (batch < 1 || batch > 65535 || !plane_fits(a.Nr2, a.Nc2) || !aligned16(a.in)) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) case h: return run_fwd<h>(a, batch, seg_hint, s);
        PDWT_LONG_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t try_launch_dwt2_inv_long(const Inv2DArgs& a, int batch, hipStream_t s, int seg_hint) {
    if ((a.hlen & 1) || a.hlen < kLongMinHlen || a.hlen > kMaxTaps) return hipErrorNotSupported;
    if ((a.Ncc & 3) || a.Nc != 2 * a.Ncc || (a.in_bstride & 3)) return hipErrorNotSupported;
    if (a.Nr > 2 * a.Nrc || a.Nr < 2 * a.Nrc - 1) return hipErrorNotSupported;
    if (batch < 1 || batch > 65535 || !plane_fits(a.Nr, a.Nc)) return hipErrorNotSupported;
    if (!aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D)) return hipErrorNotSupported;
    switch (a.hlen) {
#define X(h) case h: return run_inv<h>(a, batch, seg_hint, s);
        PDWT_LONG_HLENS(X)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
