// dwt2_pyr3_kernels.hpp -- THREE consecutive 2D DWT levels in one launch, for SMALL images (gfx950).
//
// Why: a level of a small image (<= 2^20 samples) costs what a dependent launch costs (an empty one takes 2.4 us,
// the 1 MB two-level pyramid 3.9 us), so the step of a 512^2 three-level transform is its number of launches:
// pyramid(1,2) + level 3 forward, level 3 + pyramid(2,1) inverse = four launches, 15.5 us.  Here one workgroup
// carries a tile through all three levels out of LDS: two launches per forward+inverse.
// Reference: w_kern_forward_pass1/2 (pdwt/src/separable.cu:91-176) and w_kern_inverse_pass1/2 (:246-328), three
// times over; the index conventions are those of oracle/pdwt_oracle.c (analysis centre hlen/2 - 1; polyphase
// synthesis with h2 = hlen/2, c = h2/2, s = 1 - (h2 & 1)).
//
// Forward, per workgroup and axis (T = tile edge at level l+3, C = hlen/2 - 1):
//   level l+3 outputs  [o3, o3 + T)
//   A_{l+2} positions  [s2, s2 + R2),  s2 = 2 o3 - C,  R2 = 2 T  + hlen - 2
//   A_{l+1} positions  [s1, s1 + R1),  s1 = 2 s2 - C,  R1 = 2 R2 + hlen - 2
//   input   positions  [s0, s0 + R0),  s0 = 2 s1 - C,  R0 = 2 R1 + hlen - 2      (periodic wrap on the INPUT only)
// The workgroup writes the details it owns: level l+1 rows/columns [4 o3, 4 o3 + 4 T), level l+2 [2 o3, 2 o3 + 2 T),
// all four bands of level l+3 on [o3, o3 + T).  The halo of the intermediate levels is recomputed by the neighbours
// ((R0 / 8T)^2 = 1.5x for db2, 2.7x for db4): irrelevant where the launch is the cost.
// Exactness: approximation values at positions outside [0, N) computed from periodically wrapped input equal the
// reference's per-level periodization when every level's input has even sizes: the host requires rows % 8 == 0 and
// cols % 8 == 0.
//
// Inverse, per workgroup and axis: image samples [o0, o0 + T0), T0 = 64, need the coefficients
//   [b/2 + D1, b/2 + D1 + N1) of level l+1,  D1 = floor((0 + s)/2) - c,  N1 = floor((T0 - 1 + s)/2) - c + h2 - D1
// and so on down to level l+3 (Pyr3InvGeom); all regions are loaded up front (periodic wrap), then three
// column+row synthesis phases run out of LDS, the last one storing the image tile.
//
// Written over real_t with plain loops (no packed math): the kernels are latency-bound by construction.  CPU
// emulation (tests/cpu_emu): PDWT_FOR_THREADS / PDWT_SYNC as in the other LDS kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

struct Pyr3Args {
    const real_t* in;        // forward: A_l (or the image), (N0r, N0c); inverse: A_{l+3}
    real_t* out;             // forward: A_{l+3}; inverse: A_l (or the image)
    real_t* det[3][3];       // det[k] = (H, V, D) of level l+1+k: forward written, inverse read
    int N0r, N0c;            // multiples of 8
    int tiles_x, tiles_y;
    FilterBank fb;           // forward: analysis (dec_lo, dec_hi); inverse: synthesis (rec_lo, rec_hi)
};

constexpr int pyr3_floor_half(int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); }
// periodic wrap of an index in [-n, 2n)
PDWT_DEVICE int pyr3_wrap1(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }

template <int HLEN, int T>
struct Pyr3FwdGeom {
    static constexpr int C = HLEN / 2 - 1;
    static constexpr int R2 = 2 * T + HLEN - 2, R1 = 2 * R2 + HLEN - 2, R0 = 2 * R1 + HLEN - 2;
    static constexpr int S_IN = R0 * R0;       // staged input; later the level-(l+2) row pass (2 x R1 x R2) and A_{l+2} (R2 x R2)
    static constexpr int S_T1 = 2 * R0 * R1;   // level-(l+1) row pass (L | H); later the level-(l+3) row pass (2 x R2 x T)
    static constexpr int S_A1 = R1 * R1;
    static constexpr int LDS = S_IN + S_T1 + S_A1;
    static_assert(2 * R1 * R2 + R2 * R2 <= S_IN && 2 * R2 * T <= S_T1, "later buffers alias dead ones");
    static_assert(HLEN >= 2 && (HLEN & 1) == 0, "even filters");
};

// rows [y0, y0 + n) of one pass: out[y][k] = sum_j in[y][2k + j] f[HLEN-1-j], for both filters
template <int HLEN, int NT>
PDWT_DEVICE void pyr3_rows(int tid, const real_t* src, int src_stride, int rows, int cols_out, const FilterBank& fb,
                           real_t* outL, real_t* outH) {
    for (int idx = tid; idx < rows * cols_out; idx += NT) {
        const int y = idx / cols_out, k = idx - y * cols_out;
        const real_t* r = src + y * src_stride + 2 * k;
        real_t l = 0, h = 0;
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            l = pdwt_fma(r[j], fb.lo[HLEN - 1 - j], l);
            h = pdwt_fma(r[j], fb.hi[HLEN - 1 - j], h);
        }
        outL[idx] = l;
        outH[idx] = h;
    }
}

// column pass over the (L | H) row-pass buffers (2 n_out + HLEN - 2 rows of `cols`): A to `A` (n_out x cols, LDS or
// global), the details of the positions the workgroup owns to global memory
template <int HLEN, int NT, bool LAST>
PDWT_DEVICE void pyr3_cols(int tid, const real_t* tL, const real_t* tH, int n_out, int cols, const FilterBank& fb, real_t* A,
                           int gy0, int gx0, int own_y0, int own_x0, int own_n, int Nr, int Nc, real_t* gA, real_t* gH,
                           real_t* gV, real_t* gD) {
    for (int idx = tid; idx < n_out * cols; idx += NT) {
        const int i = idx / cols, k = idx - i * cols;
        real_t ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
        for (int j = 0; j < HLEN; ++j) {
            const real_t vL = tL[(2 * i + j) * cols + k], vH = tH[(2 * i + j) * cols + k];
            const real_t flo = fb.lo[HLEN - 1 - j], fhi = fb.hi[HLEN - 1 - j];
            ll = pdwt_fma(vL, flo, ll);
            lh = pdwt_fma(vL, fhi, lh);
            hl = pdwt_fma(vH, flo, hl);
            hh = pdwt_fma(vH, fhi, hh);
        }
        if (!LAST) A[idx] = ll;
        const int gy = gy0 + i, gx = gx0 + k;
        if (gy >= own_y0 && gy < own_y0 + own_n && gx >= own_x0 && gx < own_x0 + own_n && gy < Nr && gx < Nc) {
            const long long o = (long long)gy * Nc + gx;
            if (LAST) gA[o] = ll;
            gH[o] = lh;  // (row low, column high), (row high, column low), (row high, column high): separable.cu:135-176
            gV[o] = hl;
            gD[o] = hh;
        }
    }
}

template <int HLEN, int T, int NT>
PDWT_DEVICE void dwt2_fwd_pyr3_tile(const Pyr3Args& a, int bx, int by, int bz, real_t* smem) {
    using G = Pyr3FwdGeom<HLEN, T>;
    constexpr int C = G::C, R0 = G::R0, R1 = G::R1, R2 = G::R2;
    real_t* sIn = smem;
    real_t* t1L = smem + G::S_IN;
    real_t* t1H = t1L + R0 * R1;
    real_t* A1 = smem + G::S_IN + G::S_T1;
    real_t* t2L = smem;  // aliases sIn (dead after the first row pass)
    real_t* t2H = t2L + R1 * R2;
    real_t* A2 = t2H + R1 * R2;
    real_t* t3L = t1L;   // aliases the first row pass (dead after the first column pass)
    real_t* t3H = t3L + R2 * T;

    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2, N3r = a.N0r >> 3, N3c = a.N0c >> 3;
    const int o3x = bx * T, o3y = by * T;
    const int s2x = 2 * o3x - C, s2y = 2 * o3y - C;
    const int s1x = 2 * s2x - C, s1y = 2 * s2y - C;
    const int s0x = 2 * s1x - C, s0y = 2 * s1y - C;
    const long long b0 = (long long)bz * a.N0r * a.N0c, b1 = (long long)bz * N1r * N1c, b2 = (long long)bz * N2r * N2c,
                    b3 = (long long)bz * N3r * N3c;

    // ---- stage the input region.  Branch-free: a constant number of trips per thread, indices past the end clamped
    // (those threads re-write the last element with the same value), the periodic wrap as one conditional add/sub --
    // so all of a thread's loads are issued back to back.  Images smaller than the region wrap more than once: modulo.
    PDWT_FOR_THREADS(tid, NT) {
        const real_t* PDWT_RESTRICT in = a.in + b0;
        constexpr int TRIPS = (R0 * R0 + NT - 1) / NT;
        if (a.N0r >= R0 && a.N0c >= R0) {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int idx = tid + t * NT;
                idx = idx < R0 * R0 ? idx : R0 * R0 - 1;
                const int y = idx / R0, x = idx - y * R0;
                sIn[idx] = in[(long long)pyr3_wrap1(s0y + y, a.N0r) * a.N0c + pyr3_wrap1(s0x + x, a.N0c)];
            }
        } else {
            for (int idx = tid; idx < R0 * R0; idx += NT) {
                const int y = idx / R0, x = idx - y * R0;
                sIn[idx] = in[(long long)true_mod(s0y + y, a.N0r) * a.N0c + true_mod(s0x + x, a.N0c)];
            }
        }
    }
    PDWT_SYNC();
    // ---- level l+1
    PDWT_FOR_THREADS(tid, NT) { pyr3_rows<HLEN, NT>(tid, sIn, R0, R0, R1, a.fb, t1L, t1H); }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        pyr3_cols<HLEN, NT, false>(tid, t1L, t1H, R1, R1, a.fb, A1, s1y, s1x, 4 * o3y, 4 * o3x, 4 * T, N1r, N1c, nullptr,
                                   a.det[0][0] + b1, a.det[0][1] + b1, a.det[0][2] + b1);
    }
    PDWT_SYNC();
    // ---- level l+2
    PDWT_FOR_THREADS(tid, NT) { pyr3_rows<HLEN, NT>(tid, A1, R1, R1, R2, a.fb, t2L, t2H); }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        pyr3_cols<HLEN, NT, false>(tid, t2L, t2H, R2, R2, a.fb, A2, s2y, s2x, 2 * o3y, 2 * o3x, 2 * T, N2r, N2c, nullptr,
                                   a.det[1][0] + b2, a.det[1][1] + b2, a.det[1][2] + b2);
    }
    PDWT_SYNC();
    // ---- level l+3
    PDWT_FOR_THREADS(tid, NT) { pyr3_rows<HLEN, NT>(tid, A2, R2, R2, T, a.fb, t3L, t3H); }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        pyr3_cols<HLEN, NT, true>(tid, t3L, t3H, T, T, a.fb, nullptr, o3y, o3x, o3y, o3x, T, N3r, N3c, a.out + b3,
                                  a.det[2][0] + b3, a.det[2][1] + b3, a.det[2][2] + b3);
    }
}

// ------------------------------------------------------------------------------------------------ inverse
template <int HLEN, int T0>
struct Pyr3InvGeom {
    static constexpr int H2 = HLEN / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    // coefficient region of the level below, relative to (tile base) / 2: first index and count
    static constexpr int D1 = pyr3_floor_half(0 + S) - C;
    static constexpr int N1 = pyr3_floor_half(T0 - 1 + S) - C + H2 - D1;
    static constexpr int D2 = pyr3_floor_half(D1 + S) - C;
    static constexpr int N2 = pyr3_floor_half(D1 + N1 - 1 + S) - C + H2 - D2;
    static constexpr int D3 = pyr3_floor_half(D2 + S) - C;
    static constexpr int N3 = pyr3_floor_half(D2 + N2 - 1 + S) - C + H2 - D3;
    // LDS layout
    static constexpr int O_C3 = 0;                      // A, H, V, D of level l+3: 4 x N3 x N3
    static constexpr int O_D2 = O_C3 + 4 * N3 * N3;     // H, V, D of level l+2: 3 x N2 x N2
    static constexpr int O_D1 = O_D2 + 3 * N2 * N2;     // H, V, D of level l+1: 3 x N1 x N1
    static constexpr int O_A2 = O_D1 + 3 * N1 * N1;     // reconstructed A_{l+2}: N2 x N2
    static constexpr int O_A1 = O_A2 + N2 * N2;         // reconstructed A_{l+1}: N1 x N1
    static constexpr int O_U = O_A1 + N1 * N1;          // column-synthesis results (u1 | u2): 2 x T0 x N1 at most
    static constexpr int LDS = O_U + 2 * T0 * N1;
    static_assert(HLEN >= 2 && (HLEN & 1) == 0 && (T0 % 8) == 0, "even filters, tiles of whole level-3 samples");
};

// One synthesis level out of LDS: (a, d1) and (d2, d3) -- n_in x n_in regions starting at relative index d_in --
// give the n_out x n_out region starting at relative index d_out of the level above (d_* relative to the tile bases,
// base_out = 2 base_in).  Columns first, then rows (separable.cu:246-328).  `to_global`: the rows pass stores to
// (gout, row length gNc) at (gy0 + q, gx0 + p), guarded by the image size.
template <int HLEN, int NT, bool TO_GLOBAL>
PDWT_DEVICE void pyr3_synth(const real_t* a, const real_t* d1, const real_t* d2, const real_t* d3, int n_in, int d_in, int n_out,
                            int d_out, const FilterBank& fb, real_t* u, real_t* out, real_t* gout, int gNr, int gNc, int gy0,
                            int gx0) {
    constexpr int H2 = HLEN / 2, C = H2 / 2, S = (H2 & 1) ? 0 : 1;
    real_t* u1 = u;
    real_t* u2 = u + n_out * n_in;
    PDWT_FOR_THREADS(tid, NT) {
        for (int idx = tid; idx < n_out * n_in; idx += NT) {  // column synthesis: rows q of the output region, all n_in columns
            const int q = idx / n_in, x = idx - q * n_in;
            const int p = d_out + q + S;
            const int rel = pyr3_floor_half(p) - C - d_in, par = 1 - (p & 1);
            real_t r1 = 0, r2 = 0;
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const int t = HLEN - 1 - (2 * j + par);
                const int src = (rel + j) * n_in + x;
                r1 = pdwt_fma(a[src], fb.lo[t], r1);
                r1 = pdwt_fma(d1[src], fb.hi[t], r1);
                r2 = pdwt_fma(d2[src], fb.lo[t], r2);
                r2 = pdwt_fma(d3[src], fb.hi[t], r2);
            }
            u1[idx] = r1;
            u2[idx] = r2;
        }
    }
    PDWT_SYNC();
    PDWT_FOR_THREADS(tid, NT) {
        for (int idx = tid; idx < n_out * n_out; idx += NT) {  // row synthesis
            const int q = idx / n_out, g = idx - q * n_out;
            const int p = d_out + g + S;
            const int rel = pyr3_floor_half(p) - C - d_in, par = 1 - (p & 1);
            real_t r = 0;
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const int t = HLEN - 1 - (2 * j + par);
                r = pdwt_fma(u1[q * n_in + rel + j], fb.lo[t], r);
                r = pdwt_fma(u2[q * n_in + rel + j], fb.hi[t], r);
            }
            if (TO_GLOBAL) {
                const int gy = gy0 + q, gx = gx0 + g;
                if (gy < gNr && gx < gNc) gout[(long long)gy * gNc + gx] = r;
            } else {
                out[idx] = r;
            }
        }
    }
    PDWT_SYNC();
}

// the n x n regions (same geometry) of NP planes into dst[p n n + y n + x]; FAST: branch-free as in the forward staging
// (needs Nr, Nc >= n), otherwise the modulo form
template <int NT, int NP, int N, bool FAST>
PDWT_DEVICE void pyr3_stage(int tid, const real_t* p0, const real_t* p1, const real_t* p2, const real_t* p3, int Nr, int Nc, int y0,
                            int x0, real_t* dst) {
    constexpr int TOTAL = NP * N * N, TRIPS = (TOTAL + NT - 1) / NT;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int idx = tid + t * NT;
        if (!FAST && idx >= TOTAL) break;
        idx = idx < TOTAL ? idx : TOTAL - 1;
        const int p = idx / (N * N), r = idx - p * (N * N), y = r / N, x = r - y * N;
        const real_t* pl = p == 0 ? p0 : (p == 1 ? p1 : (p == 2 ? p2 : p3));
        const int gy = FAST ? pyr3_wrap1(y0 + y, Nr) : true_mod(y0 + y, Nr);
        const int gx = FAST ? pyr3_wrap1(x0 + x, Nc) : true_mod(x0 + x, Nc);
        dst[idx] = pl[(long long)gy * Nc + gx];
    }
}

template <int HLEN, int T0, int NT>
PDWT_DEVICE void dwt2_inv_pyr3_tile(const Pyr3Args& a, int bx, int by, int bz, real_t* smem) {
    using G = Pyr3InvGeom<HLEN, T0>;
    constexpr int N1 = G::N1, N2 = G::N2, N3 = G::N3;
    const int N1r = a.N0r >> 1, N1c = a.N0c >> 1, N2r = a.N0r >> 2, N2c = a.N0c >> 2, N3r = a.N0r >> 3, N3c = a.N0c >> 3;
    const long long b0 = (long long)bz * a.N0r * a.N0c, b1 = (long long)bz * N1r * N1c, b2 = (long long)bz * N2r * N2c,
                    b3 = (long long)bz * N3r * N3c;
    const int o0x = bx * T0, o0y = by * T0;
    real_t* c3 = smem + G::O_C3;
    real_t* d2 = smem + G::O_D2;
    real_t* d1 = smem + G::O_D1;
    real_t* A2 = smem + G::O_A2;
    real_t* A1 = smem + G::O_A1;
    real_t* u = smem + G::O_U;

    // ---- every coefficient region the tile needs, up front (one basic block: the loads of all levels overlap)
    PDWT_FOR_THREADS(tid, NT) {
        const int y3 = o0y / 8 + G::D3, x3 = o0x / 8 + G::D3, y2 = o0y / 4 + G::D2, x2 = o0x / 4 + G::D2;
        const int y1 = o0y / 2 + G::D1, x1 = o0x / 2 + G::D1;
        if (N3r >= N3 && N3c >= N3) {  // then the finer levels are large enough too
            pyr3_stage<NT, 4, N3, true>(tid, a.in + b3, a.det[2][0] + b3, a.det[2][1] + b3, a.det[2][2] + b3, N3r, N3c, y3, x3, c3);
            pyr3_stage<NT, 3, N2, true>(tid, a.det[1][0] + b2, a.det[1][1] + b2, a.det[1][2] + b2, nullptr, N2r, N2c, y2, x2, d2);
            pyr3_stage<NT, 3, N1, true>(tid, a.det[0][0] + b1, a.det[0][1] + b1, a.det[0][2] + b1, nullptr, N1r, N1c, y1, x1, d1);
        } else {
            pyr3_stage<NT, 4, N3, false>(tid, a.in + b3, a.det[2][0] + b3, a.det[2][1] + b3, a.det[2][2] + b3, N3r, N3c, y3, x3, c3);
            pyr3_stage<NT, 3, N2, false>(tid, a.det[1][0] + b2, a.det[1][1] + b2, a.det[1][2] + b2, nullptr, N2r, N2c, y2, x2, d2);
            pyr3_stage<NT, 3, N1, false>(tid, a.det[0][0] + b1, a.det[0][1] + b1, a.det[0][2] + b1, nullptr, N1r, N1c, y1, x1, d1);
        }
    }
    PDWT_SYNC();
    pyr3_synth<HLEN, NT, false>(c3, c3 + N3 * N3, c3 + 2 * N3 * N3, c3 + 3 * N3 * N3, N3, G::D3, N2, G::D2, a.fb, u, A2, nullptr, 0,
                                0, 0, 0);
    pyr3_synth<HLEN, NT, false>(A2, d2, d2 + N2 * N2, d2 + 2 * N2 * N2, N2, G::D2, N1, G::D1, a.fb, u, A1, nullptr, 0, 0, 0, 0);
    pyr3_synth<HLEN, NT, true>(A1, d1, d1 + N1 * N1, d1 + 2 * N1 * N1, N1, G::D1, T0, 0, a.fb, u, nullptr, a.out + b0, a.N0r, a.N0c,
                               o0y, o0x);
}

#ifndef PDWT_CPU_EMU
// tile id -> (bx, by): consecutive workgroups are horizontal neighbours (they share halo columns in L2)
template <int HLEN, int T, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_pyr3_kernel(const Pyr3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pyr3_smem[];
    const int t = blockIdx.x;
    if (t >= a.tiles_x * a.tiles_y) return;
    dwt2_fwd_pyr3_tile<HLEN, T, NT>(a, t % a.tiles_x, t / a.tiles_x, blockIdx.y, reinterpret_cast<real_t*>(pyr3_smem));
}
template <int HLEN, int T0, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_pyr3_kernel(const Pyr3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pyr3_smem[];
    const int t = blockIdx.x;
    if (t >= a.tiles_x * a.tiles_y) return;
    dwt2_inv_pyr3_tile<HLEN, T0, NT>(a, t % a.tiles_x, t / a.tiles_x, blockIdx.y, reinterpret_cast<real_t*>(pyr3_smem));
}
#endif

}  // namespace pdwt
