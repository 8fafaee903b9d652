// ops_kernels.hpp -- streaming coefficient operators (HBM-bound, 16 B per lane).
//
// All bands of a plan live in one arena with 256-B aligned, zero-padded bands, so
// an operator that treats every band alike sweeps ONE contiguous range with
// real4_t loads/stores (the reference launches one 16x16-thread kernel per level
// or one cuBLAS-v1 call per band, pdwt/src/common.cu:219-371, wt.cu:368-416).
// Padding is zero and every operator here maps 0 -> 0.
//
// Semantics (restated in oracle/pdwt_oracle.c):
//   soft   x <- copysign(max(|x|-b,0), x)      pdwt/src/common.cu:13-52
//   hard   x <- x * [|x| > b]                  pdwt/src/common.cu:57-97
//   linf   x <- copysign(min(|x|,b), x)        pdwt/src/common.cu:101-137
//   scale  x <- x * s   (shrink: s = 1/(1+b))  pdwt/src/common.cu:347-371
//   group soft threshold                        pdwt/src/common.cu:145-198
//   axpy   dst += alpha*src                     pdwt/src/common.cu:499-526
//   norms  sum|x|, sum x^2                      pdwt/src/wt.cu:368-416
//   circshift                                   pdwt/src/common.cu:202-211
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.hpp"

namespace pdwt {

template <int OP>
__device__ __forceinline__ real_t ew_apply(real_t x, real_t b) {
    if (OP == EW_SOFT) return copysign(fmax(fabs(x) - b, real_t(0)), x);
    if (OP == EW_HARD) return (fabs(x) - b > real_t(0)) ? x : real_t(0);
    if (OP == EW_LINF) return copysign(fmin(fabs(x), b), x);
    return x * b;
}

// n4 = number of real4_t groups; the host only passes 16-B aligned, padded ranges
template <int OP>
__global__ void __launch_bounds__(256) ew_kernel(real4_t* __restrict__ p, long long n4, real_t b) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        real4_t v = p[i];
        v.x = ew_apply<OP>(v.x, b);
        v.y = ew_apply<OP>(v.y, b);
        v.z = ew_apply<OP>(v.z, b);
        v.w = ew_apply<OP>(v.w, b);
        p[i] = v;
    }
}

// nb detail bands (1 or 3) of one level, optional approximation band
__global__ void __launch_bounds__(256) group_soft_kernel(real_t* __restrict__ d0, real_t* __restrict__ d1,
                                                         real_t* __restrict__ d2, real_t* __restrict__ ap,
                                                         long long n, real_t beta, int nb) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const real_t a = d0[i];
        const real_t b = nb > 1 ? d1[i] : real_t(0);
        const real_t c = nb > 1 ? d2[i] : real_t(0);
        const real_t e = ap ? ap[i] : real_t(0);
        const real_t nrm = sqrt(a * a + b * b + c * c + e * e);
        const real_t res = (nrm == real_t(0)) ? real_t(0) : fmax(real_t(1) - beta / nrm, real_t(0));
        d0[i] = a * res;
        if (nb > 1) {
            d1[i] = b * res;
            d2[i] = c * res;
        }
        if (ap) ap[i] = e * res;
    }
}

__global__ void __launch_bounds__(256) axpy_kernel(real4_t* __restrict__ dst, const real4_t* __restrict__ src,
                                                   long long n4, real_t alpha) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        real4_t d = dst[i];
        const real4_t s = src[i];
        d.x = pdwt_fma(alpha, s.x, d.x);
        d.y = pdwt_fma(alpha, s.y, d.y);
        d.z = pdwt_fma(alpha, s.z, d.z);
        d.w = pdwt_fma(alpha, s.w, d.w);
        dst[i] = d;
    }
}

// sum|x| and sum x^2 in two launches without atomics (fp64 accumulation: one wave-shuffle reduction per wavefront, one LDS
// step per block): every block writes its two partial sums to its own slot, one block adds the slots up.  (Rounds 1-3 had
// every block add to ONE pair of doubles with atomicAdd: 4096 same-address fp64 atomics serialise at the memory side --
// norm1 of 2^24 coefficients took 81 us, a twelfth of the streaming rate, found in round 4 by tools/opsbench.py.)
constexpr int kNormsMaxBlocks = 1024;
__global__ void __launch_bounds__(256) norms_partial_kernel(const real4_t* __restrict__ p, long long n4,
                                                            double* __restrict__ partial) {
    double s1 = 0.0, s2 = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const real4_t v = p[i];
        s1 += (double)fabs(v.x) + (double)fabs(v.y) + (double)fabs(v.z) + (double)fabs(v.w);
        s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {  // 64-wide wavefront
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    __shared__ double part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        part[0][wave] = s1;
        part[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        partial[2 * blockIdx.x + 1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}
// The inner loop of iterative shrinkage -- soft_threshold, then norm1 of what is left (pdwt/README.md:6-7; wt.cu:308-315,
// 396-416) -- as ONE sweep: x' = soft(x, beta) for every coefficient (beta = b_lo in the first split4 groups -- the
// approximation band --, b_hi behind; keep_lo: the first part as it is), sum |x'| and sum x'^2 per block.  STORE = false is the
// read-only form for plans whose inverse applies the threshold as it loads the details (2D SWT): the norms of the
// thresholded coefficients without touching them.
template <bool STORE>
__global__ void __launch_bounds__(256) soft_norms_partial_kernel(real4_t* __restrict__ p, long long n4, long long split4, real_t b_lo,
                                                                 real_t b_hi, int keep_lo, double* __restrict__ partial) {
    double s1 = 0.0, s2 = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        real4_t v = p[i];
        const bool lo = i < split4;
        if (!(lo && keep_lo)) {
            const real_t b = lo ? b_lo : b_hi;
            v.x = ew_apply<EW_SOFT>(v.x, b);
            v.y = ew_apply<EW_SOFT>(v.y, b);
            v.z = ew_apply<EW_SOFT>(v.z, b);
            v.w = ew_apply<EW_SOFT>(v.w, b);
            if (STORE) p[i] = v;
        }
        s1 += (double)fabs(v.x) + (double)fabs(v.y) + (double)fabs(v.z) + (double)fabs(v.w);
        s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    __shared__ double part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        part[0][wave] = s1;
        part[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        partial[2 * blockIdx.x + 1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}
// one block: out[0] = sum of partial[2 b], out[1] = sum of partial[2 b + 1], b < nblocks (fixed order: deterministic)
__global__ void __launch_bounds__(256) norms_final_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        s1 += partial[2 * b];
        s2 += partial[2 * b + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    __shared__ double part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        part[0][wave] = s1;
        part[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        out[1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}

// out[b][y][x] = in[b][(y - sr) mod Nr][(x - sc) mod Nc],  0 <= sr < Nr, 0 <= sc < Nc
__global__ void __launch_bounds__(256) circshift_kernel(const real_t* __restrict__ in, real_t* __restrict__ out,
                                                        int Nr, int Nc, int sr, int sc) {
    const long long plane = (long long)Nr * Nc;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < plane) {
        const int y = (int)(idx / Nc), x = (int)(idx - (long long)y * Nc);
        int r = y - sr, c = x - sc;
        if (r < 0) r += Nr;
        if (c < 0) c += Nc;
        const long long b = (long long)blockIdx.y * plane;
        out[b + idx] = in[b + (long long)r * Nc + c];
    }
}

// deterministic test/bench input, identical to oracle_fill_hash and
// tests/golden/make_golden.py:hash_input
__global__ void __launch_bounds__(256) fill_hash_kernel(real_t* __restrict__ x, long long n, uint32_t seed,
                                                        real_t scale, long long index_offset) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = (uint32_t)(i + index_offset) ^ seed;
        h ^= h >> 16;
        h *= 0x7FEB352Du;
        h ^= h >> 15;
        h *= 0x846CA68Bu;
        h ^= h >> 16;
        x[i] = (real_t)((float)(h >> 8) * (1.0f / 16777216.0f)) * scale;  // 24-bit value, exact in fp32
    }
}

// plain 16-B grid-stride copy: the measured ceiling a streaming kernel of the same footprint is compared with
// (pdwt_time_copy; MI355X_MICROARCH.md quotes 6.29 TB/s for this kernel shape on buffers far beyond the Infinity Cache)
__global__ void __launch_bounds__(256) copy_kernel(const real4_t* __restrict__ a, real4_t* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

}  // namespace pdwt
