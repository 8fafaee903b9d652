// pkbench.hip -- issue-rate micro-benchmark (developer tool): v_fma_f32 against v_pk_fma_f32 on gfx950, by wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/pkbench.hip -o tools/bin/pkbench
// Prints nanoseconds per wavefront-instruction per SIMD: what one packed multiply-add costs next to two plain ones decides
// whether the level kernels should be written for instruction COUNT (packed) or not.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) kern(float* out, int iters, float s0, float s1) {
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const v2f t = {s0, s1};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 16 independent v_fma_f32 (SGPR multiplier)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(s0));
        } else if (MODE == 1) {  // 16 independent v_pk_fma_f32 (SGPR pair)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "s"(t));
        } else if (MODE == 2) {  // 16 independent v_pk_fma_f32, half broadcast through op_sel
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(p[i]) : "s"(t));
        } else if (MODE == 3) {  // one dependent chain of v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[0]) : "s"(t));
        } else if (MODE == 4) {  // two interleaved chains
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 1]) : "s"(t));
        } else if (MODE == 5) {  // v_mov_b32
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
        } else if (MODE == 6) {  // v_pk_fma_f32 with VGPR operands only
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 1234.5f) out[threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d) {
    const int iters = 4096;
    for (int wps = 1; wps <= 8; wps *= 2) {  // wavefronts per SIMD: blocks of 256 threads (one wavefront per SIMD each), wps blocks per CU
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms * 1e6 / ((double)iters * 16 * wps);  // ns per wavefront-instruction per SIMD
        printf("%-44s %d wavefronts/SIMD: %6.3f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, per, per * 2.4);
    }
}

int main() {
    float* d;
    CK(hipMalloc(&d, 1 << 20));
    run<0>("v_fma_f32 x16 independent", d);
    run<1>("v_pk_fma_f32 x16 independent (sgpr pair)", d);
    run<2>("v_pk_fma_f32 x16 independent, op_sel bcast", d);
    run<6>("v_pk_fma_f32 x16 independent (vgpr only)", d);
    run<3>("v_pk_fma_f32 one dependent chain", d);
    run<4>("v_pk_fma_f32 two interleaved chains", d);
    run<5>("v_mov_b32 x16", d);
    return 0;
}
