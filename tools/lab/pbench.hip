// pbench.hip -- "pattern copies": kernels that move exactly the bytes of one 2D DWT level with the access
// pattern of the wave-per-tile kernels (dwt2_wave_kernels.hpp) but no arithmetic, to tell what the pattern itself
// costs next to a flat float4 copy (developer tool; numbers in profiles/r02t_pbench_*.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pbench.hip -o tools/bin/pbench
//   tools/bin/pbench [batch] [N] [zeros: 0|1]
// Forward pattern: a wavefront walks down a strip of 256 columns; per pair of input rows it issues two 16-B loads
// per lane and writes one row of each of the four bands: ST8 = four 8-B stores per lane (what dwt2_fwd_wave does),
// ST16 = lane pairs swap halves (DPP quad_perm) and write two 16-B stores per lane.  NT = nontemporal loads/stores.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ f4 ld16(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    else return *reinterpret_cast<const f4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st16(float* p, f4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st8(float* p, f2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f2*>(p));
    else *reinterpret_cast<f2*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ f2 ld8(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
    else return *reinterpret_cast<const f2*>(p);
}

__device__ __forceinline__ float swap1(float v) {  // value of the other lane of the pair (2k <-> 2k+1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

__global__ void copy4(const f4* __restrict__ a, f4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
__global__ void copy4nt(const f4* __restrict__ a, f4* __restrict__ b, long long n4, int ntl, int nts) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 v = ntl ? __builtin_nontemporal_load(a + i) : a[i];
        if (nts) __builtin_nontemporal_store(v, b + i);
        else b[i] = v;
    }
}

// forward pattern; grid = strips * segs waves per image (blockDim 256 = 4 waves), seg_out output rows per wave
template <bool ST16, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) pat_fwd(const float* __restrict__ in, float* __restrict__ out, int N, int seg_out, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int segs = (N / 2) / seg_out;
    if (wave >= strips * segs) return;
    const int strip = wave % strips, seg = wave / strips;
    const long long img = (long long)blockIdx.y * N * N;
    const long long q = (long long)(N / 2) * (N / 2);
    const float* src = in + img + (long long)(2 * seg * seg_out) * N + strip * 256 + 4 * lane;
    float* o = out + img + (long long)(seg * seg_out) * (N / 2) + strip * 128;
    const bool odd = lane & 1;
#pragma unroll 2
    for (int r = 0; r < seg_out; ++r) {
        f4 x0 = ld16<NTL>(src), x1 = ld16<NTL>(src + N);
        src += 2 * N;
        f2 A = {x0.x + x1.x, x0.z + x1.z}, H = {x0.y + x1.y, x0.w + x1.w}, V = {x0.x - x1.x, x0.z - x1.z}, D = {x0.y - x1.y, x0.w - x1.w};
        if constexpr (!ST16) {
            st8<NTS>(o + 2 * lane, A);
            st8<NTS>(o + q + 2 * lane, H);
            st8<NTS>(o + 2 * q + 2 * lane, V);
            st8<NTS>(o + 3 * q + 2 * lane, D);
        } else {
            // even lane keeps (A, H) of both lanes of the pair, odd lane (V, D)
            f2 give0 = odd ? A : V, give1 = odd ? H : D;
            f2 got0 = {swap1(give0.x), swap1(give0.y)}, got1 = {swap1(give1.x), swap1(give1.y)};
            f2 own0 = odd ? V : A, own1 = odd ? D : H;
            f4 s0 = odd ? f4{got0.x, got0.y, own0.x, own0.y} : f4{own0.x, own0.y, got0.x, got0.y};
            f4 s1 = odd ? f4{got1.x, got1.y, own1.x, own1.y} : f4{own1.x, own1.y, got1.x, got1.y};
            float* p = o + (odd ? 2 * q : 0) + 2 * (lane & ~1);
            st16<NTS>(p, s0);
            st16<NTS>(p + q, s1);
        }
        o += N / 2;
    }
}

// inverse pattern: per coefficient row four band loads, two 16-B stores (output rows 2r, 2r+1)
template <bool LD16, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) pat_inv(const float* __restrict__ co, float* __restrict__ img_out, int N, int seg_out, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int segs = (N / 2) / seg_out;
    if (wave >= strips * segs) return;
    const int strip = wave % strips, seg = wave / strips;
    const long long img = (long long)blockIdx.y * N * N;
    const long long q = (long long)(N / 2) * (N / 2);
    float* dst = img_out + img + (long long)(2 * seg * seg_out) * N + strip * 256 + 4 * lane;
    const float* c = co + img + (long long)(seg * seg_out) * (N / 2) + strip * 128;
    const bool odd = lane & 1;
#pragma unroll 2
    for (int r = 0; r < seg_out; ++r) {
        f2 A, H, V, D;
        if constexpr (!LD16) {
            A = ld8<NTL>(c + 2 * lane);
            H = ld8<NTL>(c + q + 2 * lane);
            V = ld8<NTL>(c + 2 * q + 2 * lane);
            D = ld8<NTL>(c + 3 * q + 2 * lane);
        } else {
            const float* p = c + (odd ? 2 * q : 0) + 2 * (lane & ~1);
            f4 s0 = ld16<NTL>(p), s1 = ld16<NTL>(p + q);  // even: A, H of the pair; odd: V, D of the pair
            f2 keep0 = odd ? f2{s0.z, s0.w} : f2{s0.x, s0.y}, keep1 = odd ? f2{s1.z, s1.w} : f2{s1.x, s1.y};
            f2 give0 = odd ? f2{s0.x, s0.y} : f2{s0.z, s0.w}, give1 = odd ? f2{s1.x, s1.y} : f2{s1.z, s1.w};
            f2 got0 = {swap1(give0.x), swap1(give0.y)}, got1 = {swap1(give1.x), swap1(give1.y)};
            A = odd ? got0 : keep0; H = odd ? got1 : keep1; V = odd ? keep0 : got0; D = odd ? keep1 : got1;
        }
        f4 y0 = {A.x + V.x, H.x + D.x, A.y + V.y, H.y + D.y}, y1 = {A.x - V.x, H.x - D.x, A.y - V.y, H.y - D.y};
        st16<NTS>(dst, y0);
        st16<NTS>(dst + N, y1);
        dst += 2 * N;
        c += N / 2;
    }
}

// wide strips: a lane owns 8 adjacent columns (two 16-B loads per row), a wavefront 512 columns: every band row leaves as
// one 16-B store per lane = 1 KiB contiguous per wavefront (the 256-column strips write 512 B per band and row)
__global__ void __launch_bounds__(256) pat_fwd_wide(const float* __restrict__ in, float* __restrict__ out, int N, int seg_out, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int segs = (N / 2) / seg_out;
    if (wave >= strips * segs) return;
    const int strip = wave % strips, seg = wave / strips;
    const long long img = (long long)blockIdx.y * N * N;
    const long long q = (long long)(N / 2) * (N / 2);
    const float* src = in + img + (long long)(2 * seg * seg_out) * N + strip * 512 + 8 * lane;
    float* o = out + img + (long long)(seg * seg_out) * (N / 2) + strip * 256 + 4 * lane;
#pragma unroll 2
    for (int r = 0; r < seg_out; ++r) {
        f4 x0 = ld16<false>(src), x1 = ld16<false>(src + 4), y0 = ld16<false>(src + N), y1 = ld16<false>(src + N + 4);
        src += 2 * N;
        f4 A = {x0.x + y0.x, x0.z + y0.z, x1.x + y1.x, x1.z + y1.z}, H = {x0.y + y0.y, x0.w + y0.w, x1.y + y1.y, x1.w + y1.w};
        f4 V = {x0.x - y0.x, x0.z - y0.z, x1.x - y1.x, x1.z - y1.z}, D = {x0.y - y0.y, x0.w - y0.w, x1.y - y1.y, x1.w - y1.w};
        st16<false>(o, A);
        st16<false>(o + q, H);
        st16<false>(o + 2 * q, V);
        st16<false>(o + 3 * q, D);
        o += N / 2;
    }
}
__global__ void __launch_bounds__(256) pat_inv_wide(const float* __restrict__ co, float* __restrict__ img_out, int N, int seg_out, int strips) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int segs = (N / 2) / seg_out;
    if (wave >= strips * segs) return;
    const int strip = wave % strips, seg = wave / strips;
    const long long img = (long long)blockIdx.y * N * N;
    const long long q = (long long)(N / 2) * (N / 2);
    float* dst = img_out + img + (long long)(2 * seg * seg_out) * N + strip * 512 + 8 * lane;
    const float* c = co + img + (long long)(seg * seg_out) * (N / 2) + strip * 256 + 4 * lane;
#pragma unroll 2
    for (int r = 0; r < seg_out; ++r) {
        f4 A = ld16<false>(c), H = ld16<false>(c + q), V = ld16<false>(c + 2 * q), D = ld16<false>(c + 3 * q);
        f4 a0 = {A.x + V.x, H.x + D.x, A.y + V.y, H.y + D.y}, a1 = {A.z + V.z, H.z + D.z, A.w + V.w, H.w + D.w};
        f4 b0 = {A.x - V.x, H.x - D.x, A.y - V.y, H.y - D.y}, b1 = {A.z - V.z, H.z - D.z, A.w - V.w, H.w - D.w};
        st16<false>(dst, a0);
        st16<false>(dst + 4, a1);
        st16<false>(dst + N, b0);
        st16<false>(dst + N + 4, b1);
        dst += 2 * N;
        c += N / 2;
    }
}

static float time_it(const std::function<void()>& fn, int reps = 40, int warm = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; i++) fn();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1;
    const int N = argc > 2 ? atoi(argv[2]) : 4096;
    const long long n = (long long)B * N * N;
    float *img, *co;
    CK(hipMalloc((void**)&img, n * sizeof(float)));
    CK(hipMalloc((void**)&co, n * sizeof(float)));
    {
        std::vector<float> h((size_t)n);
        const bool zeros = argc > 3 && atoi(argv[3]) != 0;  // third argument 1: all-zero data
        for (size_t i = 0; i < h.size(); i++) h[i] = zeros ? 0.f : (float)((i * 2654435761u) >> 8 & 0xffff) / 257.0f;
        CK(hipMemcpy(img, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(co, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    const double bytes = 8.0 * n;
    printf("pattern copies, batch %d, N = %d, %.1f MB moved per launch\n", B, N, bytes / 1e6);
    for (int grid : {1024, 2048}) {
        float us = time_it([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, (const f4*)img, (f4*)co, n / 4); });
        printf("copy4 grid=%4d plain                   %8.2f us  %7.1f GB/s\n", grid, us, bytes / us / 1e3);
        for (int m = 1; m < 4; ++m) {
            us = time_it([&] { hipLaunchKernelGGL(copy4nt, dim3(grid), dim3(256), 0, 0, (const f4*)img, (f4*)co, n / 4, m & 1, m >> 1); });
            printf("copy4 grid=%4d nt-load=%d nt-store=%d     %8.2f us  %7.1f GB/s\n", grid, m & 1, m >> 1, us, bytes / us / 1e3);
        }
    }
    const int strips = N / 256;
    for (int seg_out : {8, 16, 32, 64}) {
        if ((N / 2) % seg_out) continue;
        const int waves = strips * ((N / 2) / seg_out);
        dim3 g((waves + 3) / 4, B), b(256);
        printf("-- seg_out=%d: %d waves per image\n", seg_out, waves);
#define RUN(name, K) { float us = time_it([&] { hipLaunchKernelGGL(K, g, b, 0, 0, (const float*)img, co, N, seg_out, strips); }); \
                       printf("%-40s %8.2f us  %7.1f GB/s\n", name, us, bytes / us / 1e3); }
        RUN("fwd pattern, 8-B stores", (pat_fwd<false, false, false>));
        RUN("fwd pattern, 16-B stores (pair swap)", (pat_fwd<true, false, false>));
        RUN("fwd pattern, 8-B stores, nt ld+st", (pat_fwd<false, true, true>));
        RUN("fwd pattern, 8-B stores, nt st", (pat_fwd<false, false, true>));
        RUN("fwd pattern, 16-B stores, nt ld+st", (pat_fwd<true, true, true>));
        RUN("inv pattern, 8-B loads", (pat_inv<false, false, false>));
        RUN("inv pattern, 16-B loads (pair swap)", (pat_inv<true, false, false>));
        RUN("inv pattern, 8-B loads, nt ld+st", (pat_inv<false, true, true>));
        RUN("inv pattern, 16-B loads, nt ld+st", (pat_inv<true, true, true>));
        RUN("inv pattern, 8-B loads, nt ld", (pat_inv<false, true, false>));
        if (N % 512 == 0) {  // 512-column strips: the same segment length gives half the wavefronts
            const int wstrips = N / 512, wwaves = wstrips * ((N / 2) / seg_out);
            dim3 gw((wwaves + 3) / 4, B);
#define RUNW(name, K) { float us = time_it([&] { hipLaunchKernelGGL(K, gw, b, 0, 0, (const float*)img, co, N, seg_out, wstrips); }); \
                        printf("%-40s %8.2f us  %7.1f GB/s  (%d waves)\n", name, us, bytes / us / 1e3, wwaves); }
            RUNW("fwd pattern, 512-column strips", pat_fwd_wide);
            RUNW("inv pattern, 512-column strips", pat_inv_wide);
        }
    }
    return 0;
}
