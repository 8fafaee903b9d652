// ops_kernels.hpp -- streaming coefficient operators (HBM-bound, 16 B per lane).
//
// All bands of a plan live in one arena with 256-B aligned, zero-padded bands, so
// an operator that treats every band alike sweeps ONE contiguous range with
// float4 loads/stores (the reference launches one 16x16-thread kernel per level
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
__device__ __forceinline__ float ew_apply(float x, float b) {
    if (OP == EW_SOFT) return copysignf(fmaxf(fabsf(x) - b, 0.0f), x);
    if (OP == EW_HARD) return (fabsf(x) - b > 0.0f) ? x : 0.0f;
    if (OP == EW_LINF) return copysignf(fminf(fabsf(x), b), x);
    return x * b;
}

// n4 = number of float4 groups; the host only passes 16-B aligned, padded ranges
template <int OP>
__global__ void __launch_bounds__(256) ew_kernel(float4* __restrict__ p, long long n4, float b) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = p[i];
        v.x = ew_apply<OP>(v.x, b);
        v.y = ew_apply<OP>(v.y, b);
        v.z = ew_apply<OP>(v.z, b);
        v.w = ew_apply<OP>(v.w, b);
        p[i] = v;
    }
}

// nb detail bands (1 or 3) of one level, optional approximation band
__global__ void __launch_bounds__(256) group_soft_kernel(float* __restrict__ d0, float* __restrict__ d1,
                                                         float* __restrict__ d2, float* __restrict__ ap,
                                                         long long n, float beta, int nb) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float a = d0[i];
        const float b = nb > 1 ? d1[i] : 0.f;
        const float c = nb > 1 ? d2[i] : 0.f;
        const float e = ap ? ap[i] : 0.f;
        const float nrm = sqrtf(a * a + b * b + c * c + e * e);
        const float res = (nrm == 0.f) ? 0.f : fmaxf(1.0f - beta / nrm, 0.0f);
        d0[i] = a * res;
        if (nb > 1) {
            d1[i] = b * res;
            d2[i] = c * res;
        }
        if (ap) ap[i] = e * res;
    }
}

__global__ void __launch_bounds__(256) axpy_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                   long long n4, float alpha) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 d = dst[i];
        const float4 s = src[i];
        d.x = fmaf(alpha, s.x, d.x);
        d.y = fmaf(alpha, s.y, d.y);
        d.z = fmaf(alpha, s.z, d.z);
        d.w = fmaf(alpha, s.w, d.w);
        dst[i] = d;
    }
}

// out[0] += sum|x| ; out[1] += sum x^2   (fp64 accumulation: one wave-shuffle
// reduction per wavefront, one LDS step per block, two atomics per block)
__global__ void __launch_bounds__(256) norms_kernel(const float4* __restrict__ p, long long n4,
                                                    double* __restrict__ out) {
    double s1 = 0.0, s2 = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = p[i];
        s1 += (double)fabsf(v.x) + (double)fabsf(v.y) + (double)fabsf(v.z) + (double)fabsf(v.w);
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
        atomicAdd(&out[0], part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(&out[1], part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

// out[b][y][x] = in[b][(y - sr) mod Nr][(x - sc) mod Nc],  0 <= sr < Nr, 0 <= sc < Nc
__global__ void __launch_bounds__(256) circshift_kernel(const float* __restrict__ in, float* __restrict__ out,
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
__global__ void __launch_bounds__(256) fill_hash_kernel(float* __restrict__ x, long long n, uint32_t seed,
                                                        float scale, long long index_offset) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = (uint32_t)(i + index_offset) ^ seed;
        h ^= h >> 16;
        h *= 0x7FEB352Du;
        h ^= h >> 15;
        h *= 0x846CA68Bu;
        h ^= h >> 16;
        x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) * scale;
    }
}

}  // namespace pdwt
