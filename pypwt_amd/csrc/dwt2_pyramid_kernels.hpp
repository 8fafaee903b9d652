// dwt2_pyramid_kernels.hpp -- TWO consecutive 2D DWT levels in one launch (gfx950).
//
// Why: a dependent kernel launch on MI355X costs ~2.5 us plus ~1.5-2 us of pipeline fill before a
// level kernel streams at full rate (rocprofv3: 4.5-5 us for the 512^2 and 1024^2 levels of the
// 4096^2 benchmark, whose data would take < 1 us to move; profiles/r01c_rocprofv3_summary.txt).  For
// small levels that fixed cost IS the runtime, so the deep levels are fused pairwise: one workgroup
// stages the input of level l, computes its tile of level l (details to HBM, approximation kept in
// LDS) and, from that approximation tile, its tile of level l+1.
//
// Forward geometry per workgroup (H = filter length, C = H/2-1, E = C & 1):
//   level l+1 tile        TY2 x TX2 outputs at (oy2, ox2)
//   level l   region      R1Y x R1X = (2 TY2 + H-2) x (2 TX2 + H-2 + 2E) at (2 oy2 - C, 2 ox2 - C - E)
//                         (E widens it to an even column origin: 8-B aligned detail stores);
//                         the workgroup OWNS (writes details for) the central 2TY2 x 2TX2 block
//   input     region      (2 R1Y + H-2) x (2 R1X + H-2), staged with 16-B loads
// The level-l halo is recomputed by neighbouring workgroups ((R1Y R1X)/(4 TY2 TX2) = 1.9x for db4 at
// 4x32): cheap for small levels, which is where this kernel is used (host: launch_dwt2_pyramid.hip).
//
// Exactness: approximation values at "virtual" positions outside the image are computed from
// periodically wrapped input; that equals the reference's per-level periodization
// (pdwt/src/separable.cu:114-121) when the input AND the intermediate level have even sizes
// (no repeated-last-sample extension): the host requires rows % 4 == 0 and cols % 8 == 0.
#pragma once

#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

struct FwdPyr2Args {
    const float* in;             // level l-1 approximation (or the image): (N0r, N0c)
    float *H1, *V1, *D1;         // level l details: (N0r/2, N0c/2)
    float *A2, *H2, *V2, *D2;    // level l+1 bands: (N0r/4, N0c/4)
    int N0r, N0c;
    long long in_bstride, l1_bstride, l2_bstride;
    int tiles_x, tiles_y;        // level l+1 tile grid
    FilterBankI fb;              // (dec_lo, dec_hi)
};

template <int HLEN, int TX2, int TY2>
struct Pyr2Geom {
    static constexpr int H = HLEN;
    static constexpr int C = H / 2 - 1;
    static constexpr int E = C & 1;
    static constexpr int R1X = 2 * TX2 + H - 2 + 2 * E;
    static constexpr int R1Y = 2 * TY2 + H - 2;
    static constexpr int S1 = ((R1X + 3) & ~3) + 4;            // row stride of the A(l) tile in LDS
    static constexpr int R0X = 2 * R1X + H - 2;
    static constexpr int R0Y = 2 * R1Y + H - 2;
    static constexpr int PADL = (((C - 2 * E) % 4) + 4) % 4;   // (4 ox2 - 3C - 2E) mod 4
    static constexpr int RXA = (PADL + R0X + 3) & ~3;
    static constexpr int NV1 = (PADL + H + 2 + 3) & ~3;
    static constexpr int NV2 = (E + H + 2 + 3) & ~3;
    static constexpr int SIN = R0Y * RXA + 8;                   // staged input (+ slack for the last b128)
    static constexpr int LDS_FLOATS = SIN + 2 * R0Y * R1X;
    static_assert(R1Y * S1 + 2 * R1Y * TX2 <= SIN, "level l+1 buffers alias the staged input");
    static_assert((TX2 & 1) == 0 && (R1X & 1) == 0 && (R1Y & 1) == 0, "pairs of columns / rows");
};

template <int HLEN, int TX2, int TY2, int NT>
PDWT_DEVICE void dwt2_fwd_pyr2_tile(const FwdPyr2Args& a, int bx, int by, int bz, float* smem) {
    using G = Pyr2Geom<HLEN, TX2, TY2>;
    constexpr int H = G::H, C = G::C, E = G::E, R1X = G::R1X, R1Y = G::R1Y, S1 = G::S1, R0Y = G::R0Y, PADL = G::PADL,
                  RXA = G::RXA, NV1 = G::NV1, NV2 = G::NV2;
    constexpr int V4 = RXA / 4;

    float* sIn = smem;                                         // R0Y x RXA
    v2f* tLH1 = reinterpret_cast<v2f*>(smem + G::SIN);         // R0Y x R1X (L,H) pairs
    float* sA1 = smem;                                         // R1Y x S1   (aliases sIn, dead after phase 2)
    v2f* tLH2 = reinterpret_cast<v2f*>(smem + ((R1Y * S1 + 3) & ~3));  // R1Y x TX2 pairs (aliases sIn)

    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2;
    const int ox2 = bx * TX2, oy2 = by * TY2;
    const int r1x0 = 2 * ox2 - C - E, r1y0 = 2 * oy2 - C;
    const int xa = 2 * r1x0 - C - PADL;  // multiple of 4
    const int y0 = 2 * r1y0 - C;

    // ---- phase 1: stage the input region with 16-B loads (periodic wrap; interior tiles skip it).  Branch-free: a
    // constant number of trips, indices past the end clamped (those threads re-write the last quad with the same
    // value), so a thread's loads are all in flight together -- as a loop with an exit test per trip every load was
    // waited for before the next one was issued (s_waitcnt vmcnt(0) inside the loop: four round trips instead of one).
    PDWT_FOR_THREADS(tid, NT) {
        const float* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
        const bool interior = xa >= 0 && xa + RXA <= a.N0c && y0 >= 0 && y0 + R0Y <= a.N0r;
        constexpr int TOTAL = R0Y * V4, TRIPS = (TOTAL + NT - 1) / NT;
        if (interior) {
            const float* base = in + (long long)y0 * a.N0c + xa;
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4, g = idx - r * V4;
                *reinterpret_cast<v4f*>(sIn + r * RXA + 4 * g) = *reinterpret_cast<const v4f*>(base + (long long)r * a.N0c + 4 * g);
            }
        } else if (a.N0r >= R0Y && a.N0c >= RXA) {  // one conditional add/sub wraps every index
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < TOTAL ? idx : TOTAL - 1;
                const int r = idx / V4, g = idx - r * V4;
                int sy = y0 + r, sx = xa + 4 * g;
                sy = sy < 0 ? sy + a.N0r : (sy >= a.N0r ? sy - a.N0r : sy);
                sx = sx < 0 ? sx + a.N0c : (sx >= a.N0c ? sx - a.N0c : sx);
                *reinterpret_cast<v4f*>(sIn + r * RXA + 4 * g) = *reinterpret_cast<const v4f*>(in + (long long)sy * a.N0c + sx);
            }
        } else {
            for (int idx = tid; idx < R0Y * V4; idx += NT) {
                const int r = idx / V4, g = idx - r * V4;
                const int sy = wrap_periodic(y0 + r, a.N0r);
                const int sx = wrap_periodic(xa + 4 * g, a.N0c);
                *reinterpret_cast<v4f*>(sIn + r * RXA + 4 * g) = *reinterpret_cast<const v4f*>(in + (long long)sy * a.N0c + sx);
            }
        }
        if (tid < 8) sIn[R0Y * RXA + tid] = 0.f;
    }
    PDWT_SYNC();

    // ---- phase 2: level-l row analysis, two adjacent outputs per work item -> (L,H) pairs
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int HT = R1X / 2;
        for (int idx = tid; idx < R0Y * HT; idx += NT) {
            const int r = idx / HT, t = idx - r * HT;
            float v[NV1];
v4f w[NV1 / 4];
#pragma unroll
for (int q = 0; q < NV1 / 4; ++q) w[q] = lds_load16(sIn + r * RXA + 4 * t + 4 * q);
#pragma unroll
for (int q = 0; q < NV1 / 4; ++q) {
    lds_pin(w[q]);
                v[4 * q + 0] = w[q].x; v[4 * q + 1] = w[q].y; v[4 * q + 2] = w[q].z; v[4 * q + 3] = w[q].w;
}
            v2f acc0 = mk2(0.f, 0.f), acc1 = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const v2f tap = a.fb.t[H - 1 - j];
                acc0 = fma2(bc(v[PADL + j]), tap, acc0);
                acc1 = fma2(bc(v[PADL + 2 + j]), tap, acc1);
            }
            f32x4 o;
            o.x = acc0.x; o.y = acc0.y; o.z = acc1.x; o.w = acc1.y;
            *reinterpret_cast<f32x4*>(&tLH1[r * R1X + 2 * t]) = o;
        }
    }
    PDWT_SYNC();

    // ---- phase 3: level-l column analysis on the whole region: A(l) -> LDS, owned H,V,D -> HBM
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int HT = R1X / 2;
        const long long b1 = (long long)bz * a.l1_bstride;
        for (int idx = tid; idx < (R1Y / 2) * HT; idx += NT) {
            const int i2 = idx / HT, t = idx - i2 * HT;
            v2f accAV[2][2], accHD[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) accAV[i][0] = accAV[i][1] = accHD[i][0] = accHD[i][1] = mk2(0.f, 0.f);
            v4f w[H + 2];
#pragma unroll
            for (int r = 0; r < H + 2; ++r) w[r] = lds_load16(&tLH1[(4 * i2 + r) * R1X + 2 * t]);
#pragma unroll
            for (int r = 0; r < H + 2; ++r) {
                lds_pin(w[r]);
                const v2f lh0 = mk2(w[r].x, w[r].y), lh1 = mk2(w[r].z, w[r].w);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int j = r - 2 * i;
                    if (j >= 0 && j < H) {
                        const v2f tap = a.fb.t[H - 1 - j];
                        accAV[i][0] = fma2(lh0, bc(tap.x), accAV[i][0]);
                        accHD[i][0] = fma2(lh0, bc(tap.y), accHD[i][0]);
                        accAV[i][1] = fma2(lh1, bc(tap.x), accAV[i][1]);
                        accHD[i][1] = fma2(lh1, bc(tap.y), accHD[i][1]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int li = 2 * i2 + i;  // row inside the level-l region
                f32x2 v;
                v.x = accAV[i][0].x; v.y = accAV[i][1].x;
                *reinterpret_cast<f32x2*>(sA1 + li * S1 + 2 * t) = v;
                const int gy = r1y0 + li, gx = r1x0 + 2 * t;
                const bool owned = li >= C && li < C + 2 * TY2 && 2 * t >= C + E && 2 * t < C + E + 2 * TX2 &&
                                   gy < N1r && gx < N1c;
                if (owned) {
                    const long long o = b1 + (long long)gy * N1c + gx;
                    v.x = accAV[i][0].y; v.y = accAV[i][1].y; *reinterpret_cast<f32x2*>(a.V1 + o) = v;
                    v.x = accHD[i][0].x; v.y = accHD[i][1].x; *reinterpret_cast<f32x2*>(a.H1 + o) = v;
                    v.x = accHD[i][0].y; v.y = accHD[i][1].y; *reinterpret_cast<f32x2*>(a.D1 + o) = v;
                }
            }
        }
    }
    PDWT_SYNC();  // note: sA1 aliases sIn, which nobody reads after phase 2

    // ---- phase 4: level-(l+1) row analysis on the A(l) tile
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int HT = TX2 / 2;
        for (int idx = tid; idx < R1Y * HT; idx += NT) {
            const int r = idx / HT, t = idx - r * HT;
            float v[NV2];
v4f w[NV2 / 4];
#pragma unroll
for (int q = 0; q < NV2 / 4; ++q) w[q] = lds_load16(sA1 + r * S1 + 4 * t + 4 * q);
#pragma unroll
for (int q = 0; q < NV2 / 4; ++q) {
    lds_pin(w[q]);
                v[4 * q + 0] = w[q].x; v[4 * q + 1] = w[q].y; v[4 * q + 2] = w[q].z; v[4 * q + 3] = w[q].w;
}
            v2f acc0 = mk2(0.f, 0.f), acc1 = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const v2f tap = a.fb.t[H - 1 - j];
                acc0 = fma2(bc(v[E + j]), tap, acc0);
                acc1 = fma2(bc(v[E + 2 + j]), tap, acc1);
            }
            f32x4 o;
            o.x = acc0.x; o.y = acc0.y; o.z = acc1.x; o.w = acc1.y;
            *reinterpret_cast<f32x4*>(&tLH2[r * TX2 + 2 * t]) = o;
        }
    }
    PDWT_SYNC();

    // ---- phase 5: level-(l+1) column analysis -> A, H, V, D of level l+1
    PDWT_FOR_THREADS(tid, NT) {
        constexpr int HT = TX2 / 2;
        const long long b2 = (long long)bz * a.l2_bstride;
        for (int idx = tid; idx < TY2 * HT; idx += NT) {
            const int i = idx / HT, t = idx - i * HT;
            v2f aAV0 = mk2(0.f, 0.f), aAV1 = aAV0, aHD0 = aAV0, aHD1 = aAV0;
            v4f w[H];
#pragma unroll
            for (int j = 0; j < H; ++j) w[j] = lds_load16(&tLH2[(2 * i + j) * TX2 + 2 * t]);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                lds_pin(w[j]);
                const v2f lh0 = mk2(w[j].x, w[j].y), lh1 = mk2(w[j].z, w[j].w);
                const v2f tap = a.fb.t[H - 1 - j];
                aAV0 = fma2(lh0, bc(tap.x), aAV0);
                aHD0 = fma2(lh0, bc(tap.y), aHD0);
                aAV1 = fma2(lh1, bc(tap.x), aAV1);
                aHD1 = fma2(lh1, bc(tap.y), aHD1);
            }
            const int gy = oy2 + i, gx = ox2 + 2 * t;
            if (gy < N2r && gx < N2c) {
                const long long o = b2 + (long long)gy * N2c + gx;
                f32x2 v;
                v.x = aAV0.x; v.y = aAV1.x; *reinterpret_cast<f32x2*>(a.A2 + o) = v;
                v.x = aAV0.y; v.y = aAV1.y; *reinterpret_cast<f32x2*>(a.V2 + o) = v;
                v.x = aHD0.x; v.y = aHD1.x; *reinterpret_cast<f32x2*>(a.H2 + o) = v;
                v.x = aHD0.y; v.y = aHD1.y; *reinterpret_cast<f32x2*>(a.D2 + o) = v;
            }
        }
    }
}

#ifndef PDWT_CPU_EMU
PDWT_DEVICE bool pyr_xcd_tile(int block, int tiles_x, int tiles_y, int& bx, int& by) {
    const int total = tiles_x * tiles_y;
    const int chunk = (total + 7) >> 3;
    const int tile = (block & 7) * chunk + (block >> 3);
    if ((block >> 3) >= chunk || tile >= total) return false;
    by = tile / tiles_x;
    bx = tile - by * tiles_x;
    return true;
}

template <int HLEN, int TX2, int TY2, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_pyr2_kernel(const FwdPyr2Args a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int bx, by;
    if (!pyr_xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_fwd_pyr2_tile<HLEN, TX2, TY2, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt

// ===========================================================================================
// Inverse: levels l+1 and l undone in one launch.  The workgroup is an ordinary level-l inverse tile
// (dwt2_fast_kernels.hpp: coefficient region CR x CXA -> 2TY x 2TX samples) whose A(l) inputs are not
// read from HBM but synthesised, in LDS, from the level-(l+1) bands.  The synthesis halo does not
// compound: the extra level costs ~ (CR/2 + H2) x (CXA/2 + H2) coefficient quadruples per tile.
// ===========================================================================================
#include "dwt2_fast_kernels.hpp"

namespace pdwt {

struct InvPyr2Args {
    const float *A2, *H2, *V2, *D2;  // level l+1: (N0r/4, N0c/4)
    const float *H1, *V1, *D1;       // level l  : (N0r/2, N0c/2)
    float* out;                      // level l-1: (N0r, N0c)
    int N0r, N0c;
    long long l2_bstride, l1_bstride, out_bstride;
    int tiles_x, tiles_y;            // level-l coefficient tile grid (TX x TY)
    FilterBankI fb;                  // (rec_lo, rec_hi)
};

template <int HLEN, int TX, int TY>
struct InvPyr2Geom {
    using G1 = InvFastGeom<HLEN, TX>;
    static constexpr int H2 = G1::H2;
    static constexpr int CR = TY + H2 + 1;          // level-l rows staged
    static constexpr int CXA = G1::CXA;             // level-l cols staged
    static constexpr int CR2 = CR / 2 + H2 + 2;     // level-(l+1) rows (upper bound)
    static constexpr int W2 = ((CXA / 2 + H2 + 2 + 3 + 3) & ~3);  // level-(l+1) cols incl. alignment slack
    static constexpr int BASE = 4 * CR * CXA + 2 * (2 * TY) * CXA;  // floats of the level-l buffers
    static constexpr int LDS_FLOATS = BASE + 4 * CR2 * W2 + 2 * CR * W2 + 16;
};

template <int HLEN, int TX, int TY, int NT>
PDWT_DEVICE void dwt2_inv_pyr2_tile(const InvPyr2Args& a, int bx, int by, int bz, float* smem) {
    using G = InvFastGeom<HLEN, TX>;
    using P = InvPyr2Geom<HLEN, TX, TY>;
    constexpr int H = HLEN, H2 = G::H2, C2 = G::C, S = G::S, PADL = G::PADL, CXA = G::CXA;
    constexpr int CR = P::CR, CR2 = P::CR2, W2 = P::W2;
    constexpr int V4 = CXA / 4;

    v2f* sAV = reinterpret_cast<v2f*>(smem);  // level l: (A,V), (H,D), then (t1,t2) -- as in dwt2_inv_fast_tile
    v2f* sHD = sAV + CR * CXA;
    v2f* tt = sHD + CR * CXA;
    v2f* sAV2 = reinterpret_cast<v2f*>(smem + P::BASE);  // level l+1 pairs, CR2 x W2
    v2f* sHD2 = sAV2 + CR2 * W2;
    v2f* tt2 = sHD2 + CR2 * W2;                           // CR x W2

    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2;
    const int cy0 = by * TY - C2;            // level-l region origin
    const int cxa = bx * TX - C2 - PADL;     // multiple of 4
    // level-(l+1) region needed to synthesise A(l) over [cy0, cy0+CR) x [cxa, cxa+CXA)
    const int ky_lo = (cy0 + S) >> 1, ky_hi = (cy0 + CR - 1 + S) >> 1;
    const int kx_lo = (cxa + S) >> 1, kx_hi = (cxa + CXA - 1 + S) >> 1;
    const int c2y0 = ky_lo - C2, nr2 = ky_hi - C2 + H2 - c2y0;            // rows [c2y0, c2y0+nr2)
    const int c2x0 = kx_lo - C2;
    const int c2xa = c2x0 & ~3;                                            // 4-aligned (two's complement floor)
    const int nc2 = (kx_hi - C2 + H2 - c2xa + 3) & ~3;                     // cols [c2xa, c2xa+nc2), <= W2

    // ---- phase 1: stage level-(l+1) quadruples and the level-l details.  The first trip of both loops is taken with
    // clamped indices and predicated LDS writes, so its seven 16-B loads are in flight together (as two loops with an
    // exit test each, the second loop's loads waited for the first loop's); later trips (larger tiles) follow.
    PDWT_FOR_THREADS(tid, NT) {
        const long long b2 = (long long)bz * a.l2_bstride, b1 = (long long)bz * a.l1_bstride;
        const int g4 = nc2 >> 2;
        // rows of level l+1 that are not whole quads (N2c % 4 == 2: a 1000-column image, round 5): its bands are staged in PAIRS (the
        // window start and N2c are even: a pair never straddles the periodic wrap) by plain loops, the quad path below takes none
        const bool pairs2 = (N2c & 2) != 0;
        const int n2 = pairs2 ? 0 : nr2 * g4;
        constexpr int n1 = CR * V4;
        const int first = pairs2 ? 0 : NT;  // the quad path's first trip is taken together with level l's (below)
        if (pairs2) {
            const int g2 = nc2 >> 1;
            for (int idx = tid; idx < nr2 * g2; idx += NT) {
                const int r = idx / g2, g = idx - r * g2;
                const long long o = b2 + (long long)wrap_periodic(c2y0 + r, N2r) * N2c + wrap_periodic(c2xa + 2 * g, N2c);
                const f32x2 qA = *reinterpret_cast<const f32x2*>(a.A2 + o), qV = *reinterpret_cast<const f32x2*>(a.V2 + o);
                const f32x2 qH = *reinterpret_cast<const f32x2*>(a.H2 + o), qD = *reinterpret_cast<const f32x2*>(a.D2 + o);
                f32x4 w;
                w.x = qA.x; w.y = qV.x; w.z = qA.y; w.w = qV.y;
                *reinterpret_cast<f32x4*>(sAV2 + r * W2 + 2 * g) = w;
                w.x = qH.x; w.y = qD.x; w.z = qH.y; w.w = qD.y;
                *reinterpret_cast<f32x4*>(sHD2 + r * W2 + 2 * g) = w;
            }
        } else {
            const int i2 = tid < n2 ? tid : n2 - 1, i1 = tid < n1 ? tid : n1 - 1;
            const int r2 = i2 / g4, gg2 = i2 - r2 * g4;
            const int r1 = i1 / V4, gg1 = i1 - r1 * V4;
            const long long o2 = b2 + (long long)wrap_periodic(c2y0 + r2, N2r) * N2c + wrap_periodic(c2xa + 4 * gg2, N2c);
            const long long o1 = b1 + (long long)wrap_periodic(cy0 + r1, N1r) * N1c + wrap_periodic(cxa + 4 * gg1, N1c);
            const v4f qA = *reinterpret_cast<const v4f*>(a.A2 + o2), qV = *reinterpret_cast<const v4f*>(a.V2 + o2);
            const v4f qH = *reinterpret_cast<const v4f*>(a.H2 + o2), qD = *reinterpret_cast<const v4f*>(a.D2 + o2);
            const v4f vV = *reinterpret_cast<const v4f*>(a.V1 + o1);
            const v4f vH = *reinterpret_cast<const v4f*>(a.H1 + o1);
            const v4f vD = *reinterpret_cast<const v4f*>(a.D1 + o1);
            if (tid < n2) inv_fast_interleave(sAV2, sHD2, r2 * W2 + 4 * gg2, qA, qV, qH, qD);
            if (tid < n1) {
                v2f* dAV = sAV + r1 * CXA + 4 * gg1;
                dAV[0].y = vV.x; dAV[1].y = vV.y; dAV[2].y = vV.z; dAV[3].y = vV.w;
                f32x4 w;
                f32x4* dHD = reinterpret_cast<f32x4*>(sHD + r1 * CXA + 4 * gg1);
                w.x = vH.x; w.y = vD.x; w.z = vH.y; w.w = vD.y; dHD[0] = w;
                w.x = vH.z; w.y = vD.z; w.z = vH.w; w.w = vD.w; dHD[1] = w;
            }
        }
        for (int idx = tid + NT; idx < n2; idx += NT) {  // (n2 == 0 with pairs)
            const int r = idx / g4, g = idx - r * g4;
            const long long o = b2 + (long long)wrap_periodic(c2y0 + r, N2r) * N2c + wrap_periodic(c2xa + 4 * g, N2c);
            inv_fast_interleave(sAV2, sHD2, r * W2 + 4 * g, *reinterpret_cast<const v4f*>(a.A2 + o),
                                *reinterpret_cast<const v4f*>(a.V2 + o), *reinterpret_cast<const v4f*>(a.H2 + o),
                                *reinterpret_cast<const v4f*>(a.D2 + o));
        }
        for (int idx = tid + first; idx < n1; idx += NT) {
            const int r = idx / V4, g = idx - r * V4;
            const long long o = b1 + (long long)wrap_periodic(cy0 + r, N1r) * N1c + wrap_periodic(cxa + 4 * g, N1c);
            const v4f vV = *reinterpret_cast<const v4f*>(a.V1 + o);
            const v4f vH = *reinterpret_cast<const v4f*>(a.H1 + o);
            const v4f vD = *reinterpret_cast<const v4f*>(a.D1 + o);
            v2f* dAV = sAV + r * CXA + 4 * g;
            dAV[0].y = vV.x; dAV[1].y = vV.y; dAV[2].y = vV.z; dAV[3].y = vV.w;
            f32x4 w;
            f32x4* dHD = reinterpret_cast<f32x4*>(sHD + r * CXA + 4 * g);
            w.x = vH.x; w.y = vD.x; w.z = vH.y; w.w = vD.y; dHD[0] = w;
            w.x = vH.z; w.y = vD.z; w.z = vH.w; w.w = vD.w; dHD[1] = w;
        }
    }
    PDWT_SYNC();

    // ---- phase 2: level-(l+1) column synthesis -> (t1,t2) for the level-l rows [cy0, cy0+CR)
    PDWT_FOR_THREADS(tid, NT) {
        const int nk = ky_hi - ky_lo + 1, q2n = nc2 >> 1;
        for (int idx = tid; idx < nk * q2n; idx += NT) {
            const int ki = idx / q2n, q = 2 * (idx - ki * q2n);
            const int kk = ky_lo + ki;
            const int r0 = kk - C2 - c2y0;
            v2f e0 = mk2(0.f, 0.f), o0 = e0, e1 = e0, o1 = e0;
            v4f wavs[H2], whds[H2];
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                wavs[j] = lds_load16(&sAV2[(r0 + j) * W2 + q]);
                whds[j] = lds_load16(&sHD2[(r0 + j) * W2 + q]);
            }
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                lds_pin(wavs[j]);
                lds_pin(whds[j]);
                const v4f wav = wavs[j], whd = whds[j];
                const v2f te = a.fb.t[H - 2 - 2 * j], to = a.fb.t[H - 1 - 2 * j];
                const v2f av0 = mk2(wav.x, wav.y), av1 = mk2(wav.z, wav.w);
                const v2f hd0 = mk2(whd.x, whd.y), hd1 = mk2(whd.z, whd.w);
                e0 = fma2(av0, bc(te.x), e0); e0 = fma2(hd0, bc(te.y), e0);
                o0 = fma2(av0, bc(to.x), o0); o0 = fma2(hd0, bc(to.y), o0);
                e1 = fma2(av1, bc(te.x), e1); e1 = fma2(hd1, bc(te.y), e1);
                o1 = fma2(av1, bc(to.x), o1); o1 = fma2(hd1, bc(to.y), o1);
            }
            const int ge = 2 * kk - S - cy0, go = ge + 1;  // local level-l rows
            f32x4 w;
            if (ge >= 0 && ge < CR) {
                w.x = e0.x; w.y = e0.y; w.z = e1.x; w.w = e1.y;
                *reinterpret_cast<f32x4*>(&tt2[ge * W2 + q]) = w;
            }
            if (go >= 0 && go < CR) {
                w.x = o0.x; w.y = o0.y; w.z = o1.x; w.w = o1.y;
                *reinterpret_cast<f32x4*>(&tt2[go * W2 + q]) = w;
            }
        }
    }
    PDWT_SYNC();

    // ---- phase 3: level-(l+1) row synthesis -> A(l) into the .x lanes of the (A,V) plane
    PDWT_FOR_THREADS(tid, NT) {
        const int nk = kx_hi - kx_lo + 1;
        for (int idx = tid; idx < CR * nk; idx += NT) {
            const int r = idx / nk, kk = kx_lo + (idx - r * nk);
            const v2f* u = tt2 + r * W2 + (kk - C2 - c2xa);
            v2f re = mk2(0.f, 0.f), ro = mk2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const v2f w = u[j];
                re = fma2(w, a.fb.t[H - 2 - 2 * j], re);
                ro = fma2(w, a.fb.t[H - 1 - 2 * j], ro);
            }
            const int ge = 2 * kk - S - cxa, go = ge + 1;  // local level-l cols
            if (ge >= 0 && ge < CXA) sAV[r * CXA + ge].x = re.x + re.y;
            if (go >= 0 && go < CXA) sAV[r * CXA + go].x = ro.x + ro.y;
        }
    }
    PDWT_SYNC();

    // ---- phases 4, 5: the ordinary level-l inverse tile
    Inv2DFastArgs f;
    f.A = nullptr; f.H = nullptr; f.V = nullptr; f.D = nullptr;
    f.out = a.out;
    f.Nrc = N1r; f.Ncc = N1c; f.Nr = a.N0r; f.Nc = a.N0c;
    f.in_bstride = a.l1_bstride; f.out_bstride = a.out_bstride;
    f.tiles_x = a.tiles_x; f.tiles_y = a.tiles_y;
    PDWT_FOR_THREADS(tid, NT) { inv_fast_col_pass<HLEN, TX, TY, NT>(tid, sAV, sHD, tt, a.fb); }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        f.fb = a.fb;
        inv_fast_row_pass<HLEN, TX, TY, NT>(tid, tt, f, bx, by, bz);
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_pyr2_kernel(const InvPyr2Args a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    int bx, by;
    if (!pyr_xcd_tile(blockIdx.x, a.tiles_x, a.tiles_y, bx, by)) return;
    dwt2_inv_pyr2_tile<HLEN, TX, TY, NT>(a, bx, by, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
