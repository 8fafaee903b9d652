// launch_dwt2_ring.hip -- launchers of the register-ring 2D DWT level kernels for long filters (dwt2_ring_kernels.hpp).
//
// try_launch_* return hipErrorNotSupported when the level does not meet the kernels' preconditions (even filter length
// 10-20, row length a multiple of 4, 16-B aligned buffers); the caller then falls back to the LDS-tiled kernels.
//
// Geometry: a wavefront owns a strip of 64 CPL image columns and `seg` output rows (forward) / coefficient rows (inverse);
// 256-thread workgroups of four independent wavefronts, each with its own two LDS rows.
#include "dwt2_ring_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

namespace pdwt {

static void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Rows per wavefront.  The hlen - 2 image rows two vertically adjacent segments share are filtered along x by both, so long
// segments save arithmetic; but a level needs its wavefronts: measured (tools/ringbench.hip, 16 taps, forward + inverse):
// 4096^2 58.7 / 46.1 / 53.1 us with 8 / 16 / 32 rows (4096 / 2048 / 1024 wavefronts), 4 x 4096^2 246.8 / 224.4 us with
// 32 / 64 rows (4096 / 2048 wavefronts): about two wavefronts per SIMD (2048 on the chip) in every case, in whole groups
// (`unit` rows: no finished row is dropped by the range check, which costs more than a store -- a diagnostic build whose
// stores were ALL out of range ran 41 us instead of 24).
constexpr int kRingWaves = 2048;
static int ring_seg(int rows, int strips, int batch, int unit, int hint) {
    if (hint > 0) return hint;
    long long seg = (long long)rows * strips * batch / kRingWaves;
    seg = (seg + unit / 2) / unit * unit;
    if (seg < unit) seg = unit;
    if (seg > 16 * unit) seg = 16 * unit;
    return (int)seg;
}

template <int HLEN, int CPL>
static hipError_t run_fwd(const Fwd2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    using G = FwdRingGeom<HLEN, CPL>;
    constexpr int NT = 256;
    constexpr int MINB = CPL == 2 ? (HLEN <= 18 ? 4 : 3) : (HLEN <= 16 ? 3 : 2);  // wavefronts per SIMD the register budget is held to (no scratch)
    FwdWaveArgs a;
    a.in = g.in; a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D;
    a.Nr = g.Nr; a.Nc = g.Nc; a.Nr2 = g.Nr2; a.Nc2 = g.Nc2;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Nc, G::W);
    a.seg_out = ring_seg(g.Nr2, a.strips, batch, G::NS, seg_hint);
    a.segs = cdiv(g.Nr2, a.seg_out);
    interleave(a.fb, g.fb);
    const int nblk = cdiv(a.strips * a.segs, NT / 64);
    const dim3 grid(8 * cdiv(nblk, 8), batch);
    const size_t lds = (size_t)(NT / 64) * G::LDS_REALS * sizeof(real_t);
    hipLaunchKernelGGL((dwt2_fwd_ring_kernel<HLEN, CPL, NT, MINB>), grid, dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <int HLEN, int CPL>
static hipError_t run_inv(const Inv2DArgs& g, int batch, int seg_hint, hipStream_t s) {
    using G = InvRingGeom<HLEN, CPL>;
    constexpr int NT = 256;
    constexpr int MINB = CPL == 2 ? 3 : 2;  // the inverse holds four bands per row: more registers
    InvRingArgs a;
    a.A = g.A; a.H = g.H; a.V = g.V; a.D = g.D; a.out = g.out;
    a.Nrc = g.Nrc; a.Ncc = g.Ncc; a.Nr = g.Nr; a.Nc = g.Nc;
    a.in_bstride = g.in_bstride; a.out_bstride = g.out_bstride;
    a.strips = cdiv(g.Ncc, G::WC);
    a.seg_pairs = ring_seg(g.Nrc, a.strips, batch, G::H2, seg_hint);
    a.segs = cdiv(g.Nrc, a.seg_pairs);
    interleave(a.fb, g.fb);
    for (int d = 0; d < HLEN / 2; d++) {
        a.pl[d].x = g.fb.lo[HLEN - 2 - 2 * d]; a.pl[d].y = g.fb.lo[HLEN - 1 - 2 * d];
        a.ph[d].x = g.fb.hi[HLEN - 2 - 2 * d]; a.ph[d].y = g.fb.hi[HLEN - 1 - 2 * d];
    }
    const int nblk = cdiv(a.strips * a.segs, NT / 64);
    const dim3 grid(8 * cdiv(nblk, 8), batch);
    const size_t lds = (size_t)(NT / 64) * G::LDS_REALS * sizeof(real_t);
    hipLaunchKernelGGL((dwt2_inv_ring_kernel<HLEN, CPL, NT, MINB>), grid, dim3(NT), lds, s, a);
    return hipGetLastError();
}

// The inverse keeps TWO tap tables in scalar registers -- (lo, hi)[j] pairs for the row synthesis, consecutive-tap pairs of each
// filter for the column synthesis: 4 hlen SGPRs.  Up to 16 taps they fit; at 18 / 20 hipcc spills 25 / 212 of them into VGPR
// lanes (and 20 B of scratch at 20 taps) and the kernel is level with the LDS tile (tools/ringbench.hip: 38.5 against 39.4 us):
// those lengths keep the tile for the inverse.
constexpr int kRingMaxHlenInv = 16;

hipError_t try_launch_dwt2_inv_ring(const Inv2DArgs& a, int batch, hipStream_t s, int cpl, int seg_hint) {
    if (batch < 1 || batch > 65535) return hipErrorNotSupported;  // the batch is grid.y: the tiles take what does not fit
    if ((a.hlen & 1) || a.hlen < kRingMinHlen || a.hlen > kRingMaxHlenInv) return hipErrorNotSupported;
    if ((a.Ncc & 1) || a.Nc != 2 * a.Ncc || (a.in_bstride & 1) || (a.out_bstride & 3)) return hipErrorNotSupported;
    if (a.Nr > 2 * a.Nrc || a.Nr < 2 * a.Nrc - 1) return hipErrorNotSupported;
    if ((long long)a.Nc * (long long)sizeof(real_t) >= (1LL << 31)) return hipErrorNotSupported;
    if (!aligned16(a.out) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    switch (a.hlen) {
#ifdef PDWT_LAB_KERNELS  // two image columns per lane: measured behind four everywhere (A/B builds only)
#define X(h) case h: return cpl == 2 ? run_inv<h, 2>(a, batch, seg_hint, s) : run_inv<h, 4>(a, batch, seg_hint, s);
#else
#define X(h) case h: return cpl == 4 ? run_inv<h, 4>(a, batch, seg_hint, s) : hipErrorNotSupported;
#endif
        X(10) X(12) X(14) X(16)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t try_launch_dwt2_fwd_ring(const Fwd2DArgs& a, int batch, hipStream_t s, int cpl, int seg_hint) {
    if (batch < 1 || batch > 65535) return hipErrorNotSupported;
    if ((a.hlen & 1) || a.hlen < kRingMinHlen || a.hlen > kRingMaxHlen) return hipErrorNotSupported;
    if ((a.Nc & 3) || (a.in_bstride & 3) || (a.out_bstride & 1) || a.Nc2 * 2 != a.Nc) return hipErrorNotSupported;
    if ((long long)a.Nc * (long long)sizeof(real_t) >= (1LL << 31)) return hipErrorNotSupported;  // 32-bit byte offsets inside a row
    if (!aligned16(a.in) || !aligned16(a.A) || !aligned16(a.H) || !aligned16(a.V) || !aligned16(a.D))
        return hipErrorNotSupported;
    switch (a.hlen) {
#ifdef PDWT_LAB_KERNELS
#define X(h) case h: return cpl == 2 ? run_fwd<h, 2>(a, batch, seg_hint, s) : run_fwd<h, 4>(a, batch, seg_hint, s);
#else
#define X(h) case h: return cpl == 4 ? run_fwd<h, 4>(a, batch, seg_hint, s) : hipErrorNotSupported;
#endif
        X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
