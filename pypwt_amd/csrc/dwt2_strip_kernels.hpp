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

template <int HLEN, int TX2>
struct Strip2Geom {
    using P = Pyr2Geom<HLEN, TX2, 4>;  // x geometry is the tile pyramid's
    static constexpr int H = HLEN, C = P::C, E = P::E;
    static constexpr int R1X = P::R1X, S1 = P::S1, PADL = P::PADL, RXA = P::RXA, NV1 = P::NV1, NV2 = P::NV2;
    static constexpr int CH0 = 16, CH1 = 8, CH2 = 4;      // rows per chunk at the three levels
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

template <int HLEN, int TX2, int NT, int PF = 2>
PDWT_DEVICE void dwt2_fwd_strip2_wg(const FwdStrip2Args& a, int strip, int seg, int bz, float* smem) {
    using G = Strip2Geom<HLEN, TX2>;
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
template <int HLEN, int TX2, int NT, int PF = 2>
__global__ void __launch_bounds__(NT) dwt2_fwd_strip2_kernel(const FwdStrip2Args a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    // consecutive workgroup ids = consecutive column strips of one row segment (neighbours share x-halo in L2)
    const int seg = blockIdx.x / a.strips;
    dwt2_fwd_strip2_wg<HLEN, TX2, NT, PF>(a, blockIdx.x - seg * a.strips, seg, blockIdx.y, pdwt_smem);
}
#endif

}  // namespace pdwt
