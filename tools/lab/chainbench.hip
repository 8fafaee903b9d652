// chainbench.hip -- what does a dependent launch cost next to an in-kernel flag wait?  (developer tool)
// A four-stage pipeline with the byte volumes of the 4096^2 forward pyramid (stage s reads n_s bytes and writes n_s
// bytes, the first quarter of which is the next stage's input: 64, 16, 4, 1 MiB) is run (a) as four launches on one
// stream and (b) as ONE launch whose wavefronts of stage s >= 1 spin on per-producer flags (release/acquire at agent
// scope).  Every stage adds 1 to its data so stale reads show up in the check.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chainbench.hip -o tools/bin/chainbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kStages = 4;
struct ChainArgs {
    const float* in[kStages];
    float* out[kStages];
    int waves[kStages];      // wavefronts of the stage
    int first[kStages + 1];  // prefix sums
    int chunk[kStages];      // bytes a wavefront moves (multiple of 4096)
    unsigned* flag[kStages]; // per producing wavefront: epoch of its last completed run
    unsigned epoch;
    int nstages;
    int mode;                // 0 compiler fences; 1 no fences (wrong, cost reference)
};

__device__ __forceinline__ void copy_chunk(const float* __restrict__ in, float* __restrict__ out, int chunk, int lane) {
    const f4* a = reinterpret_cast<const f4*>(in) + lane;
    f4* b = reinterpret_cast<f4*>(out) + lane;
    const int n = chunk / 4096;  // 4 x 1 KiB per iteration
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        f4 v0 = a[0], v1 = a[64], v2 = a[128], v3 = a[192];
        b[0] = v0 + 1.f; b[64] = v1 + 1.f; b[128] = v2 + 1.f; b[192] = v3 + 1.f;
        a += 256; b += 256;
    }
}

__global__ void __launch_bounds__(256) stage_kernel(ChainArgs g, int s) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= g.waves[s]) return;
    copy_chunk(g.in[s] + (size_t)w * (g.chunk[s] / 4), g.out[s] + (size_t)w * (g.chunk[s] / 4), g.chunk[s], threadIdx.x & 63);
}

__global__ void __launch_bounds__(256) chain_kernel(ChainArgs g) {
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int s = 0;
    while (s + 1 < g.nstages && gw >= g.first[s + 1]) ++s;
    const int w = gw - g.first[s];
    if (w >= g.waves[s]) return;
    if (s > 0) {
        // producers of bytes [w c_s, (w+1) c_s) of stage s-1's output, plus the next one (a halo)
        const long long b0 = (long long)w * g.chunk[s];
        int p0 = (int)(b0 / g.chunk[s - 1]);
        int p1 = (int)((b0 + g.chunk[s] - 1) / g.chunk[s - 1]) + 1;
        for (int p = p0; p <= p1; ++p) {
            const int pp = p % g.waves[s - 1];
            int spins = 0;  // bounded: a lost flag shows up as mismatches, not as a hung GPU
            while (__hip_atomic_load(g.flag[s - 1] + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch &&
                   ++spins < (1 << 20))
                __builtin_amdgcn_s_sleep(1);
        }
        if (g.mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    copy_chunk(g.in[s] + (size_t)w * (g.chunk[s] / 4), g.out[s] + (size_t)w * (g.chunk[s] / 4), g.chunk[s], threadIdx.x & 63);
    if (s + 1 < g.nstages) {
        if (g.mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_s_waitcnt(0);
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(g.flag[s] + w, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void fill(float* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v + (float)(i % 251);
}
__global__ void check(const float* p, size_t n, float v, unsigned* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v + (float)(i % 251)) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
    const int nst = argc > 1 ? atoi(argv[1]) : 4;
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    size_t bytes[kStages] = {64u << 20, 16u << 20, 4u << 20, 1u << 20};
    int waves[kStages] = {1024, 512, 256, 64};
    if (argc > 3) waves[0] = atoi(argv[3]);
    if (argc > 4) waves[1] = atoi(argv[4]);
    float* buf[kStages + 1];
    for (int s = 0; s <= kStages; ++s) CK(hipMalloc(&buf[s], bytes[s ? s - 1 : 0]));
    ChainArgs g{};
    g.nstages = nst;
    g.first[0] = 0;
    for (int s = 0; s < kStages; ++s) {
        g.in[s] = buf[s];
        g.out[s] = buf[s + 1];
        g.waves[s] = waves[s];
        g.chunk[s] = (int)(bytes[s] / waves[s]);
        g.first[s + 1] = g.first[s] + waves[s];
        CK(hipMalloc(&g.flag[s], waves[s] * sizeof(unsigned)));
        CK(hipMemset(g.flag[s], 0, waves[s] * sizeof(unsigned)));
    }
    unsigned* bad;
    CK(hipMalloc(&bad, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int total = g.first[nst];
    unsigned epoch = 0;

    auto run_sep = [&]() {
        for (int s = 0; s < nst; ++s) hipLaunchKernelGGL(stage_kernel, dim3((waves[s] + 3) / 4), dim3(256), 0, st, g, s);
    };
    auto run_chain = [&](int mode) {
        g.epoch = ++epoch;
        g.mode = mode;
        hipLaunchKernelGGL(chain_kernel, dim3((total + 3) / 4), dim3(256), 0, st, g);
    };
    auto validate = [&](const char* name, auto&& run) {
        unsigned nbad_total = 0;
        for (int it = 0; it < 12; ++it) {
            const float v = (float)(it * 7 % 13);
            hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, st, buf[0], bytes[0] / 4, v);
            run();
            CK(hipMemsetAsync(bad, 0, 4, st));
            for (int s = 0; s < nst; ++s)  // stage s's whole output = its input region + 1
                hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, st, buf[s + 1], bytes[s] / 4, v + (float)(s + 1), bad);
            unsigned nb = 0;
            CK(hipMemcpyAsync(&nb, bad, 4, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            nbad_total += nb;
        }
        printf("%-28s mismatches over 12 runs: %u\n", name, nbad_total);
    };
    auto timeit = [&](const char* name, auto&& run) {
        for (int i = 0; i < 50; ++i) run();
        CK(hipStreamSynchronize(st));
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const float us = ms * 1000.f / iters;
            sum += us;
            if (us < best) best = us;
        }
        printf("%-28s %7.2f us (best of 5; mean %.2f)\n", name, best, sum / 5);
    };
    printf("stages %d, waves", nst);
    for (int s = 0; s < nst; ++s) printf(" %d", waves[s]);
    printf("\n");
    validate("separate launches", run_sep);
    validate("chain, fences", [&]() { run_chain(0); });
    validate("chain, no fences", [&]() { run_chain(1); });
    timeit("separate launches", run_sep);
    timeit("chain, fences", [&]() { run_chain(0); });
    timeit("chain, no fences", [&]() { run_chain(1); });
    for (int s = 0; s < nst; ++s) {
        char nm[64];
        snprintf(nm, sizeof nm, "stage %d alone", s);
        timeit(nm, [&]() { hipLaunchKernelGGL(stage_kernel, dim3((waves[s] + 3) / 4), dim3(256), 0, st, g, s); });
    }
    return 0;
}
