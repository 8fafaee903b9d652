// dwt2_strip_kernels.hpp -- levels l and l+1 of a 2D DWT in one launch, STREAMING down column strips.
//
// The tile-pyramid kernel (dwt2_pyramid_kernels.hpp) recomputes a 1.9x halo of level l and needs 40-60 KB
// of LDS, so it only pays for small levels.  This kernel fuses the two levels at the cost of the
// ordinary single-level kernel: a workgroup owns a strip of 4*TX2 input columns and walks down it in
// chunks of 16 input rows.  Per chunk it produces 8 rows of level l (details -> HBM) and 4 rows of
// level l+1 (all four bands -> HBM); what the column filters need from earlier chunks -- the last
// hlen-2 rows of row-filtered (L,H) pairs of each level -- stays in LDS in small carry buffers, so the
// level-l approximation never goes to HBM and the y-halo is paid once per segment (3(hlen-2) warm-up
// rows per `seg2` level-(l+1) rows) instead of once per 16-row tile.  The next chunk's 16-B loads are
// in flight while the current chunk is computed.
//
// Per level pair this moves 8 B per input sample instead of 8 + 2 (no A_l write + read), and replaces
// two dependent launches by one.
//
// Exactness conditions are those of the tile pyramid (rows % 4 == 0, cols % 16 == 0: every level even).
#pragma once

#include "dwt2_pyramid_kernels.hpp"

namespace pdwt {

struct FwdStrip2Args {
    const float* in;
    float *H1, *V1, *D1;
    float *A2, *H2, *V2, *D2;
    int N0r, N0c;
    long long in_bstride, l1_bstride, l2_bstride;
    int strips;  // column strips of TX2 level-(l+1) columns
    int segs;    // row segments of seg2 level-(l+1) rows
    int seg2;
    FilterBankI fb;
};

template <int HLEN, int TX2, int CHUNK2 = 4>
struct Strip2Geom {
    using P = Pyr2Geom<HLEN, TX2, 4>;  // x geometry is the tile pyramid's
    static constexpr int H = HLEN, C = P::C, E = P::E;
    static constexpr int R1X = P::R1X, S1 = P::S1, PADL = P::PADL, RXA = P::RXA, NV1 = P::NV1, NV2 = P::NV2;
    static constexpr int CH2 = CHUNK2, CH1 = 2 * CH2, CH0 = 4 * CH2;  // rows per chunk at the three levels
    static constexpr int D = (3 * C + 1) / 2;             // level-(l+1) row lag, see kernel
    static constexpr int KEEP0 = H - 2;                   // carried (L,H) rows of level l
    static constexpr int KEEP1 = 2 * D - C;               // carried (L,H) rows of level l+1
    static constexpr int T1R = KEEP0 + CH0;               // rows of the level-l (L,H) buffer
    static constexpr int T2R = KEEP1 + CH1;               // rows of the level-(l+1) (L,H) buffer
    static constexpr int SIN = CH0 * RXA + 8;
    static constexpr int LDS_FLOATS = SIN + 2 * T1R * R1X + CH1 * S1 + 2 * T2R * TX2 + 16;
    static_assert(2 * (CH2 - 1) + H <= T2R, "level-(l+1) column window fits its buffer");
    static_assert(2 * (CH1 - 1) + H <= T1R, "level-l column window fits its buffer");
    static_assert(KEEP1 <= CH1 && KEEP0 <= CH0, "carry copies do not overlap their source");
};

template <int HLEN, int TX2, int NT, int PF = 2, int CHUNK2 = 4>
PDWT_DEVICE void dwt2_fwd_strip2_wg(const FwdStrip2Args& a, int strip, int seg, int bz, float* smem) {
    using G = Strip2Geom<HLEN, TX2, CHUNK2>;
    constexpr int H = G::H, C = G::C, E = G::E, R1X = G::R1X, S1 = G::S1, PADL = G::PADL, RXA = G::RXA, NV1 = G::NV1,
                  NV2 = G::NV2, CH0 = G::CH0, CH1 = G::CH1, CH2 = G::CH2, D = G::D, KEEP0 = G::KEEP0, KEEP1 = G::KEEP1,
                  T1R = G::T1R;
    constexpr int V4 = RXA / 4;
    constexpr int NLD = (CH0 * V4 + NT - 1) / NT;

    float* sIn = smem;                                          // CH0 x RXA
    v2f* tLH1 = reinterpret_cast<v2f*>(smem + G::SIN);          // T1R x R1X pairs: [0,KEEP0) carried, then the chunk
    float* sA1 = smem + G::SIN + 2 * T1R * R1X;                 // CH1 x S1
    v2f* tLH2 = reinterpret_cast<v2f*>(sA1 + CH1 * S1);         // T2R x TX2 pairs: [0,KEEP1) carried, then the chunk

    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2;
    const int ox2 = strip * TX2, oy2 = seg * a.seg2;
    const int n2 = (oy2 + a.seg2 <= N2r) ? a.seg2 : N2r - oy2;   // level-(l+1) rows owned by this segment
    const int r1x0 = 2 * ox2 - C - E, r1y0 = 2 * oy2 - C;
    const int xa = 2 * r1x0 - C - PADL;
    const int y0 = 2 * r1y0 - C;
    const int T = (n2 + D + CH2 - 1) / CH2;                      // chunks
    const float* PDWT_RESTRICT in = a.in + (long long)bz * a.in_bstride;
    const bool x_interior = xa >= 0 && xa + RXA <= a.N0c;

    PDWT_PER_THREAD(v4f, stage, PF * NLD, NT);  // PF chunks in flight, slot = chunk % PF (compile-time below)
    auto issue = [&](int tid, int t, int slot) {  // 16-B loads of input rows [y0 + 16 t, +16)
        const int yb = y0 + CH0 * t;
        const bool interior = x_interior && yb >= 0 && yb + CH0 <= a.N0r;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = (tid + i * NT < CH0 * V4) ? tid + i * NT : CH0 * V4 - 1;
            const int r = idx / V4, g = idx - r * V4;
            const int sy = interior ? yb + r : wrap_periodic(yb + r, a.N0r);
            const int sx = interior ? xa + 4 * g : wrap_periodic(xa + 4 * g, a.N0c);
            PDWT_MINE(stage, tid)[slot * NLD + i] = *reinterpret_cast<const v4f*>(in + (long long)sy * a.N0c + sx);
        }
    };

    PDWT_FOR_THREADS(tid, NT) {
#pragma unroll
        for (int s = 0; s < PF; ++s)
            if (s < T) issue(tid, s, s);
    }
    auto chunk = [&](const int t, const int slot) {
        // ---- P1: staged chunk -> LDS ; carry the level-(l+1) (L,H) rows ; prefetch the next chunk
        PDWT_FOR_THREADS(tid, NT) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + i * NT;
                if (idx < CH0 * V4) *reinterpret_cast<v4f*>(sIn + 4 * idx) = PDWT_MINE(stage, tid)[slot * NLD + i];
            }
            if (tid < 8) sIn[CH0 * RXA + tid] = 0.f;
            for (int idx = tid; idx < KEEP1 * TX2; idx += NT) tLH2[idx] = tLH2[CH1 * TX2 + idx];
        }
        PDWT_SYNC();
        if (t + PF < T) {
            PDWT_FOR_THREADS(tid, NT) { issue(tid, t + PF, slot); }
        }
        // ---- P2: level-l row pass of the 16 new rows -> tLH1 rows [KEEP0, KEEP0+16)
        PDWT_FOR_THREADS(tid, NT) {
            constexpr int HT = R1X / 2;
            for (int idx = tid; idx < CH0 * HT; idx += NT) {
                const int r = idx / HT, u = idx - r * HT;
                float v[NV1];
                v4f w[NV1 / 4];
#pragma unroll
                for (int q = 0; q < NV1 / 4; ++q) w[q] = lds_load16(sIn + r * RXA + 4 * u + 4 * q);
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
                *reinterpret_cast<f32x4*>(&tLH1[(KEEP0 + r) * R1X + 2 * u]) = o;
            }
        }
        PDWT_SYNC();
        // ---- P3: level-l column pass: region rows m1 = 8t - C + i (i < 8): A -> sA1, owned H,V,D -> HBM
        PDWT_FOR_THREADS(tid, NT) {
            constexpr int HT = R1X / 2;
            const long long b1 = (long long)bz * a.l1_bstride;
            for (int idx = tid; idx < (CH1 / 2) * HT; idx += NT) {
                const int i2 = idx / HT, u = idx - i2 * HT;
                v2f accAV[2][2], accHD[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) accAV[i][0] = accAV[i][1] = accHD[i][0] = accHD[i][1] = mk2(0.f, 0.f);
                v4f w[H + 2];
#pragma unroll
                for (int r = 0; r < H + 2; ++r) w[r] = lds_load16(&tLH1[(4 * i2 + r) * R1X + 2 * u]);
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
                    const int li = 2 * i2 + i;          // row inside the chunk
                    const int m1 = CH1 * t - C + li;    // row inside the level-l region of the segment
                    f32x2 v;
                    v.x = accAV[i][0].x; v.y = accAV[i][1].x;
                    *reinterpret_cast<f32x2*>(sA1 + li * S1 + 2 * u) = v;
                    const int gy = r1y0 + m1, gx = r1x0 + 2 * u;
                    const bool owned = m1 >= C && m1 < C + 2 * n2 && 2 * u >= C + E && 2 * u < C + E + 2 * TX2 &&
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
        PDWT_SYNC();
        // ---- P4: carry the last hlen-2 level-l (L,H) rows ; level-(l+1) row pass of the 8 new A rows
        PDWT_FOR_THREADS(tid, NT) {
            for (int idx = tid; idx < KEEP0 * R1X; idx += NT) tLH1[idx] = tLH1[CH0 * R1X + idx];
            constexpr int HT = TX2 / 2;
            for (int idx = tid; idx < CH1 * HT; idx += NT) {
                const int r = idx / HT, u = idx - r * HT;
                float v[NV2];
                v4f w[NV2 / 4];
#pragma unroll
                for (int q = 0; q < NV2 / 4; ++q) w[q] = lds_load16(sA1 + r * S1 + 4 * u + 4 * q);
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
                *reinterpret_cast<f32x4*>(&tLH2[(KEEP1 + r) * TX2 + 2 * u]) = o;
            }
        }
        PDWT_SYNC();
        // ---- P5: level-(l+1) column pass: rows m2 = 4t - D + i (i < 4); buffer row 0 is level-l row 2(4t - D)
        PDWT_FOR_THREADS(tid, NT) {
            constexpr int HT = TX2 / 2;
            const long long b2 = (long long)bz * a.l2_bstride;
            for (int idx = tid; idx < CH2 * HT; idx += NT) {
                const int i = idx / HT, u = idx - i * HT;
                const int m2 = CH2 * t - D + i;
                v2f aAV0 = mk2(0.f, 0.f), aAV1 = aAV0, aHD0 = aAV0, aHD1 = aAV0;
                v4f w[H];
#pragma unroll
                for (int j = 0; j < H; ++j) w[j] = lds_load16(&tLH2[(2 * i + j) * TX2 + 2 * u]);
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
                const int gy = oy2 + m2, gx = ox2 + 2 * u;
                if (m2 >= 0 && m2 < n2 && gx < N2c) {
                    const long long o = b2 + (long long)gy * N2c + gx;
                    f32x2 v;
                    v.x = aAV0.x; v.y = aAV1.x; *reinterpret_cast<f32x2*>(a.A2 + o) = v;
                    v.x = aAV0.y; v.y = aAV1.y; *reinterpret_cast<f32x2*>(a.V2 + o) = v;
                    v.x = aHD0.x; v.y = aHD1.x; *reinterpret_cast<f32x2*>(a.H2 + o) = v;
                    v.x = aHD0.y; v.y = aHD1.y; *reinterpret_cast<f32x2*>(a.D2 + o) = v;
                }
            }
        }
        PDWT_SYNC();
    };
    for (int t0 = 0; t0 < T; t0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s)  // unrolled: `slot` is a constant in each copy, the staging stays in registers
            if (t0 + s < T) chunk(t0 + s, s);
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX2, int NT, int PF = 2, int CHUNK2 = 4>
__global__ void __launch_bounds__(NT) dwt2_fwd_strip2_kernel(const FwdStrip2Args a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    // Consecutive workgroup ids land on DIFFERENT XCDs (round-robin), ids b and b + 8 on the same one: XCD x takes the
    // contiguous range [x total/8, (x+1) total/8) of (segment, strip) pairs, i.e. neighbouring column strips of one row
    // segment, which walk down their strips together and share the lines that hold their x-halo in that XCD's L2.
    // With consecutive ids every strip's two border lines per row were fetched by two XCDs: rocprofv3 FETCH_SIZE
    // 1711 MB per 16 images of 4096^2 against 1074 MB read (profiles/r03a_rocprofv3_summary_cfg2_b16.txt).
    const int total = a.strips * a.segs;
    const int lin = (total & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (total >> 3) + (int)(blockIdx.x >> 3);
    const int seg = lin / a.strips;
    dwt2_fwd_strip2_wg<HLEN, TX2, NT, PF, CHUNK2>(a, lin - seg * a.strips, seg, blockIdx.y, pdwt_smem);
}
#endif

// ===========================================================================================
// Inverse strips: levels l+1 and l undone in one launch, streaming down strips of 2*TX output columns.
// Per chunk a workgroup reads 4 coefficient rows of level l+1 (four bands) and 8 detail rows of
// level l (three bands), synthesises 8 rows of A_l in LDS and from them 16 output rows.  The H2-1 =
// hlen/2-1 coefficient rows a synthesis window shares with the previous chunk stay in LDS (carry
// rows of the two ring buffers); A_l is never in HBM.  Loads of the next chunk are in flight (one
// work item per thread: a 4-column group of one row, three or four bands) while this one is computed.
//
// Row bookkeeping (C = H2/2, S = 1 if H2 even): coefficient index kk yields outputs p = 2kk, 2kk+1 at
// position g = p - S from coefficient rows kk-C .. kk-C+H2-1.  Chunk t stages level-(l+1) rows
// [c2b+4t, +4), synthesises the 4 items whose window ends in them = level-l rows [a1+8t, +8),
// a1 = 2(c2b+C-H2+1)-S, stages the level-l details of those rows, and synthesises the 8 items whose
// window ends in them = output rows [o0+16t, +16), o0 = 2(a1+C-H2+1)-S.  c2b is chosen so that
// o0 + 16 WARM <= first owned row < o0 + 16 WARM + 4; chunk 0 is a warm-up (fills the level-l carry
// rows) when the filter has carry rows at all.
// ===========================================================================================
struct InvStrip2Args {
    const float *A2, *H2, *V2, *D2;  // level l+1: (N0r/4, N0c/4)
    const float *H1, *V1, *D1;       // level l  : (N0r/2, N0c/2)
    float* out;                      // level l-1: (N0r, N0c)
    int N0r, N0c;
    long long l2_bstride, l1_bstride, out_bstride;
    int strips;    // strips of TX level-l coefficient columns (2 TX output columns)
    int segs;      // row segments
    int seg_rows;  // output rows per segment
    FilterBankI fb;  // (rec_lo, rec_hi)
};

template <int HLEN, int TX>
struct InvStrip2Geom {
    using G1 = InvFastGeom<HLEN, TX>;
    static constexpr int H2 = G1::H2, C = G1::C, S = G1::S, PADL = G1::PADL, CXA = G1::CXA;
    static constexpr int K = H2 - 1;                 // carry rows of both rings
    static constexpr int WARM = K > 0 ? 1 : 0;
    static constexpr int R2 = K + 4, R1 = K + 8;     // ring rows
    static constexpr int W2 = ((CXA / 2 + H2 + 2 + 3 + 3) & ~3);   // level-(l+1) pair columns incl. alignment slack
    static constexpr int NK = (CXA - 1 + S) / 2 + 1;                // level-(l+1) row-synthesis items per row
    static constexpr int LDS_FLOATS = 2 * (2 * R2 * W2 + 8 * W2 + 2 * R1 * CXA + 16 * CXA) + 16;
    static_assert(K <= 4, "carry copy of the level-(l+1) ring must not overlap its source");
};

template <int HLEN, int TX, int NT>
PDWT_DEVICE void dwt2_inv_strip2_wg(const InvStrip2Args& a, int bx, int sg, int bz, float* smem) {
    using G = InvStrip2Geom<HLEN, TX>;
    constexpr int H2 = G::H2, C = G::C, S = G::S, PADL = G::PADL, CXA = G::CXA, K = G::K, WARM = G::WARM, R2 = G::R2,
                  R1 = G::R1, W2 = G::W2, NK = G::NK;
    constexpr int V4 = CXA / 4;
    constexpr int N1I = 8 * V4;           // level-l load items per chunk (row, 4-column group)
    static_assert(N1I + 4 * (W2 / 4) <= NT, "one load item per thread");

    v2f* sAV2 = reinterpret_cast<v2f*>(smem);  // R2 x W2
    v2f* sHD2 = sAV2 + R2 * W2;
    v2f* tt2 = sHD2 + R2 * W2;                 // 8 x W2
    v2f* sAV1 = tt2 + 8 * W2;                  // R1 x CXA
    v2f* sHD1 = sAV1 + R1 * CXA;
    v2f* tt1 = sHD1 + R1 * CXA;                // 16 x CXA

    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2;
    const int oy = sg * a.seg_rows;
    const int oy_end = (oy + a.seg_rows < a.N0r) ? oy + a.seg_rows : a.N0r;
    const int X = oy - 6 * (C - H2 + 1) + 3 * S;
    const int c2b = (X >> 2) - 4 * WARM;             // arithmetic shift = floor
    const int a1 = 2 * (c2b + C - H2 + 1) - S;
    const int o0 = 2 * (a1 + C - H2 + 1) - S;
    const int T = (oy_end - o0 + 15) >> 4;           // chunks 0 .. T-1

    const int cxa = bx * TX - C - PADL;              // level-l column origin, multiple of 4
    const int kx_lo = (cxa + S) >> 1;
    const int c2xa = (kx_lo - C) & ~3;               // level-(l+1) column origin, multiple of 4
    const int nc2 = ((((cxa + CXA - 1 + S) >> 1) - C + H2 - c2xa) + 3) & ~3;  // <= W2
    const int g42 = nc2 >> 2;
    const int N2I = 4 * g42;

    const long long b2 = (long long)bz * a.l2_bstride, b1 = (long long)bz * a.l1_bstride;
    float* PDWT_RESTRICT out = a.out + (long long)bz * a.out_bstride;

    // ---- prologue: carry rows of the level-(l+1) ring = rows [c2b-K, c2b)
    PDWT_FOR_THREADS(tid, NT) {
        for (int idx = tid; idx < K * g42; idx += NT) {
            const int r = idx / g42, g = idx - r * g42;
            const long long o = b2 + (long long)wrap_periodic(c2b - K + r, N2r) * N2c + wrap_periodic(c2xa + 4 * g, N2c);
            inv_fast_interleave(sAV2, sHD2, r * W2 + 4 * g, *reinterpret_cast<const v4f*>(a.A2 + o),
                                *reinterpret_cast<const v4f*>(a.V2 + o), *reinterpret_cast<const v4f*>(a.H2 + o),
                                *reinterpret_cast<const v4f*>(a.D2 + o));
        }
    }

    PDWT_PER_THREAD(v4f, stage, 4, NT);
    // one load item per thread: tid < N1I: level-l (row r, group g): V,H,D ; next N2I threads: level-(l+1)
    // (row r, group g): A,V,H,D ; the rest reload a valid address (branch-free: staging stays in VGPRs)
    auto issue = [&](int tid, int t) {
        const bool l2 = tid >= N1I;
        int i2 = tid - N1I;
        if (i2 >= N2I) i2 = N2I - 1;
        const int r = l2 ? i2 / g42 : tid / V4;
        const int g = l2 ? i2 - r * g42 : tid - r * V4;
        const int row = l2 ? wrap_periodic(c2b + 4 * t + r, N2r) : wrap_periodic(a1 + 8 * t + r, N1r);
        const int col = l2 ? wrap_periodic(c2xa + 4 * g, N2c) : wrap_periodic(cxa + 4 * g, N1c);
        const long long o = l2 ? b2 + (long long)row * N2c + col : b1 + (long long)row * N1c + col;
        const float* p0 = l2 ? a.A2 : a.V1;
        const float* p1 = l2 ? a.V2 : a.H1;
        const float* p2 = l2 ? a.H2 : a.D1;
        const float* p3 = l2 ? a.D2 : a.D1;
        PDWT_MINE(stage, tid)[0] = *reinterpret_cast<const v4f*>(p0 + o);
        PDWT_MINE(stage, tid)[1] = *reinterpret_cast<const v4f*>(p1 + o);
        PDWT_MINE(stage, tid)[2] = *reinterpret_cast<const v4f*>(p2 + o);
        PDWT_MINE(stage, tid)[3] = *reinterpret_cast<const v4f*>(p3 + o);
    };
    PDWT_FOR_THREADS(tid, NT) { issue(tid, 0); }

    for (int t = 0; t < T; ++t) {
        // ---- P1: staged loads -> the new rows of the two rings
        PDWT_FOR_THREADS(tid, NT) {
            const v4f s0 = PDWT_MINE(stage, tid)[0], s1 = PDWT_MINE(stage, tid)[1], s2 = PDWT_MINE(stage, tid)[2],
                      s3 = PDWT_MINE(stage, tid)[3];
            if (tid < N1I) {
                const int r = tid / V4, g = tid - r * V4;
                v2f* dAV = sAV1 + (K + r) * CXA + 4 * g;
                dAV[0].y = s0.x; dAV[1].y = s0.y; dAV[2].y = s0.z; dAV[3].y = s0.w;
                f32x4 w;
                f32x4* dHD = reinterpret_cast<f32x4*>(sHD1 + (K + r) * CXA + 4 * g);
                w.x = s1.x; w.y = s2.x; w.z = s1.y; w.w = s2.y; dHD[0] = w;
                w.x = s1.z; w.y = s2.z; w.z = s1.w; w.w = s2.w; dHD[1] = w;
            } else if (tid - N1I < N2I) {
                const int i2 = tid - N1I;
                const int r = i2 / g42, g = i2 - r * g42;
                inv_fast_interleave(sAV2, sHD2, (K + r) * W2 + 4 * g, s0, s1, s2, s3);
            }
        }
        PDWT_SYNC();
        if (t + 1 < T) {
            PDWT_FOR_THREADS(tid, NT) { issue(tid, t + 1); }
        }
        // ---- P2: level-(l+1) column synthesis: item i uses ring rows i .. i+H2-1 -> tt2 rows 2i, 2i+1
        PDWT_FOR_THREADS(tid, NT) {
            const int q2n = nc2 >> 1;
            for (int idx = tid; idx < 4 * q2n; idx += NT) {
                const int i = idx / q2n, q = 2 * (idx - i * q2n);
                v2f e0, o0v, e1, o1v;
                inv_col_synth2<HLEN>(&sAV2[i * W2 + q], &sHD2[i * W2 + q], W2, a.fb, e0, o0v, e1, o1v);
                f32x4 w;
                w.x = e0.x; w.y = e0.y; w.z = e1.x; w.w = e1.y;
                *reinterpret_cast<f32x4*>(&tt2[(2 * i) * W2 + q]) = w;
                w.x = o0v.x; w.y = o0v.y; w.z = o1v.x; w.w = o1v.y;
                *reinterpret_cast<f32x4*>(&tt2[(2 * i + 1) * W2 + q]) = w;
            }
        }
        PDWT_SYNC();
        // ---- P3: level-(l+1) row synthesis -> A_l into the .x lanes of the new level-l ring rows ;
        //          carry the level-(l+1) ring
        PDWT_FOR_THREADS(tid, NT) {
            for (int idx = tid; idx < 8 * NK; idx += NT) {
                const int r = idx / NK, kk = kx_lo + (idx - r * NK);
                const v2f* u = tt2 + r * W2 + (kk - C - c2xa);
                v2f re = mk2(0.f, 0.f), ro = mk2(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < H2; ++j) {
                    const v2f w = u[j];
                    re = fma2(w, a.fb.t[HLEN - 2 - 2 * j], re);
                    ro = fma2(w, a.fb.t[HLEN - 1 - 2 * j], ro);
                }
                const int ge = 2 * kk - S - cxa, go = ge + 1;  // local level-l columns
                if (ge >= 0 && ge < CXA) sAV1[(K + r) * CXA + ge].x = re.x + re.y;
                if (go >= 0 && go < CXA) sAV1[(K + r) * CXA + go].x = ro.x + ro.y;
            }
            for (int idx = tid; idx < K * W2; idx += NT) {
                sAV2[idx] = sAV2[4 * W2 + idx];
                sHD2[idx] = sHD2[4 * W2 + idx];
            }
        }
        PDWT_SYNC();
        if (t >= WARM) {
            // ---- P4: level-l column synthesis: item i uses ring rows i .. i+H2-1 -> tt1 rows 2i, 2i+1
            PDWT_FOR_THREADS(tid, NT) {
                constexpr int Q2 = CXA / 2;
                for (int idx = tid; idx < 8 * Q2; idx += NT) {
                    const int i = idx / Q2, q = 2 * (idx - i * Q2);
                    v2f e0, o0v, e1, o1v;
                    inv_col_synth2<HLEN>(&sAV1[i * CXA + q], &sHD1[i * CXA + q], CXA, a.fb, e0, o0v, e1, o1v);
                    f32x4 w;
                    w.x = e0.x; w.y = e0.y; w.z = e1.x; w.w = e1.y;
                    *reinterpret_cast<f32x4*>(&tt1[(2 * i) * CXA + q]) = w;
                    w.x = o0v.x; w.y = o0v.y; w.z = o1v.x; w.w = o1v.y;
                    *reinterpret_cast<f32x4*>(&tt1[(2 * i + 1) * CXA + q]) = w;
                }
            }
            PDWT_SYNC();
        }
        // ---- P5: level-l row synthesis, 16-B stores of the owned rows ; carry the level-l ring
        PDWT_FOR_THREADS(tid, NT) {
            if (t >= WARM) {
                constexpr int HT = TX / 2;
                constexpr int PE = PADL & 1;
                for (int idx = tid; idx < 16 * HT; idx += NT) {
                    const int gy = idx / HT, k = 2 * (idx - gy * HT);
                    float res[4];
                    inv_row_synth4<HLEN, PADL>(tt1 + gy * CXA + (PADL - PE) + k, a.fb, res);
                    const int y = o0 + 16 * t + gy;
                    const int x = 2 * (bx * TX + k);
                    if (y >= oy && y < oy_end && x < a.N0c) {
                        f32x4 v;
                        v.x = res[0]; v.y = res[1]; v.z = res[2]; v.w = res[3];
                        *reinterpret_cast<f32x4*>(out + (long long)y * a.N0c + x) = v;
                    }
                }
            }
            for (int idx = tid; idx < K * CXA; idx += NT) {
                sAV1[idx] = sAV1[8 * CXA + idx];
                sHD1[idx] = sHD1[8 * CXA + idx];
            }
        }
        PDWT_SYNC();
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int TX, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_strip2_kernel(const InvStrip2Args a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const int total = a.strips * a.segs;  // XCD-contiguous (segment, strip) ranges, see the forward kernel
    const int lin = (total & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (total >> 3) + (int)(blockIdx.x >> 3);
    const int sg = lin / a.strips;
    dwt2_inv_strip2_wg<HLEN, TX, NT>(a, lin - sg * a.strips, sg, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
