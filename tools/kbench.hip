// kbench.hip -- kernel micro-benchmark for tuning (developer tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pypwt_amd/csrc tools/kbench.hip -o build/kbench
// Times variants of the level kernels at a given size with HIP events, next to plain streaming
// kernels (copy / read / write) that calibrate what this GPU sustains at the same footprint.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "dwt2_fast_kernels.hpp"
#include "dwt2_kernels.hpp"
#include "dwt2_pyramid_kernels.hpp"
#include "dwt2_strip_kernels.hpp"

using namespace pdwt;

static const char* g_only = nullptr;
static bool skip(const char* tag) { return g_only && !strstr(tag, g_only); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
__global__ void read4(const float4* __restrict__ a, float* __restrict__ sink, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) sink[0] = s;
}
__global__ void write4(float4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static float time_it(const std::function<void()>& fn, int reps = 30, int warm = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; i++) fn();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    // back-to-back launches bracketed once: average kernel time incl. launch gaps
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

static const float DB4_LO[8] = {-0.010597401784997278f, 0.032883011666982945f, 0.030841381835986965f, -0.18703481171888114f,
                                -0.02798376941698385f, 0.6308807679295904f, 0.7148465705525415f, 0.23037781330885523f};
static const float DB4_HI[8] = {-0.23037781330885523f, 0.7148465705525415f, -0.6308807679295904f, -0.02798376941698385f,
                                0.18703481171888114f, 0.030841381835986965f, -0.032883011666982945f, -0.010597401784997278f};

template <int HLEN, int TX, int TY, int NT>
static void bench_fwd(const char* tag, const float* in, float* out4, int N, int batch) {
    if (skip(tag)) return;
    Fwd2DArgs a;
    a.in = in;
    const long long q = (long long)batch * (N / 2) * (N / 2);
    a.A = out4; a.H = out4 + q; a.V = out4 + 2 * q; a.D = out4 + 3 * q;
    a.Nr = N; a.Nc = N; a.Nr2 = N / 2; a.Nc2 = N / 2;
    a.in_bstride = (long long)N * N; a.out_bstride = (long long)(N / 2) * (N / 2);
    a.hlen = HLEN;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.lo[i] = DB4_LO[i]; a.fb.hi[i] = DB4_HI[i]; }
    const size_t lds = (size_t)fwd2d_lds_floats<TX, TY>(HLEN) * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_fwd_kernel<HLEN, TX, TY, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    dim3 grid((N / 2 + TX - 1) / TX, (N / 2 + TY - 1) / TY, batch);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, 0, a); });
    const double bytes = 8.0 * batch * N * N;
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  %7.1f GB/s (algorithmic)\n", tag, N, batch, lds, us, bytes / us / 1e3);
}

template <int HLEN, int TX, int TY, int NT>
static void bench_inv(const char* tag, const float* in4, float* out, int N, int batch) {
    if (skip(tag)) return;
    Inv2DArgs a;
    const long long q = (long long)batch * (N / 2) * (N / 2);
    a.A = in4; a.H = in4 + q; a.V = in4 + 2 * q; a.D = in4 + 3 * q; a.out = out;
    a.Nrc = N / 2; a.Ncc = N / 2; a.Nr = N; a.Nc = N;
    a.in_bstride = (long long)(N / 2) * (N / 2); a.out_bstride = (long long)N * N;
    a.hlen = HLEN;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.lo[i] = DB4_LO[7 - i]; a.fb.hi[i] = DB4_HI[7 - i]; }
    const size_t lds = (size_t)inv2d_lds_floats<TX, TY>(HLEN) * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_inv_kernel<HLEN, TX, TY, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    dim3 grid((N + 2 * TX - 1) / (2 * TX), (N + 2 * TY - 1) / (2 * TY), batch);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, 0, a); });
    const double bytes = 8.0 * batch * N * N;
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  %7.1f GB/s (algorithmic)\n", tag, N, batch, lds, us, bytes / us / 1e3);
}

template <int HLEN, int TX, int TY, int NT>
static void bench_fwd_fast(const char* tag, const float* in, float* out4, int N, int batch) {
    if (skip(tag)) return;
    Fwd2DFastArgs a;
    a.in = in;
    const long long q = (long long)batch * (N / 2) * (N / 2);
    a.A = out4; a.H = out4 + q; a.V = out4 + 2 * q; a.D = out4 + 3 * q;
    a.Nr = N; a.Nc = N; a.Nr2 = N / 2; a.Nc2 = N / 2;
    a.in_bstride = (long long)N * N; a.out_bstride = (long long)(N / 2) * (N / 2);
    a.tiles_x = (N / 2 + TX - 1) / TX; a.tiles_y = (N / 2 + TY - 1) / TY;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[i]; a.fb.t[i].y = DB4_HI[i]; }
    const size_t lds = (size_t)fwd2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_fwd_fast_kernel<HLEN, TX, TY, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    dim3 grid(8 * chunk, batch);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_fast_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, 0, a); });
    const double bytes = 8.0 * batch * N * N;
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  %7.1f GB/s (algorithmic)\n", tag, N, batch, lds, us, bytes / us / 1e3);
}

template <int HLEN, int TX, int TY, int NT>
static void bench_fwd_stream(const char* tag, const float* in, float* out4, int N, int batch, int wg_per_cu) {
    if (skip(tag)) return;
    Fwd2DFastArgs a;
    a.in = in;
    const long long q = (long long)batch * (N / 2) * (N / 2);
    a.A = out4; a.H = out4 + q; a.V = out4 + 2 * q; a.D = out4 + 3 * q;
    a.Nr = N; a.Nc = N; a.Nr2 = N / 2; a.Nc2 = N / 2;
    a.in_bstride = (long long)N * N; a.out_bstride = (long long)(N / 2) * (N / 2);
    a.tiles_x = (N / 2 + TX - 1) / TX; a.tiles_y = (N / 2 + TY - 1) / TY;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[i]; a.fb.t[i].y = DB4_HI[i]; }
    const size_t lds = (size_t)fwd2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(float);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    int nwg = 256 * wg_per_cu;
    if (nwg > 8 * chunk * batch) nwg = 8 * chunk * batch;
    nwg = (nwg + 7) & ~7;
    dim3 grid(nwg);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_fast_stream_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, 0, a, batch); });
    const double bytes = 8.0 * batch * N * N;
    printf("%-30s wg/cu=%d N=%d B=%d lds=%6zu  %8.2f us  %7.1f GB/s (algorithmic)\n", tag, wg_per_cu, N, batch, lds, us, bytes / us / 1e3);
}

template <int HLEN, int TX, int TY, int NT>
static void bench_inv_fast(const char* tag, const float* in4, float* out, int N, int batch) {
    if (skip(tag)) return;
    Inv2DFastArgs a;
    const long long q = (long long)batch * (N / 2) * (N / 2);
    a.A = in4; a.H = in4 + q; a.V = in4 + 2 * q; a.D = in4 + 3 * q; a.out = out;
    a.Nrc = N / 2; a.Ncc = N / 2; a.Nr = N; a.Nc = N;
    a.in_bstride = (long long)(N / 2) * (N / 2); a.out_bstride = (long long)N * N;
    a.tiles_x = (N + 2 * TX - 1) / (2 * TX); a.tiles_y = (N + 2 * TY - 1) / (2 * TY);
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[7 - i]; a.fb.t[i].y = DB4_HI[7 - i]; }
    const size_t lds = (size_t)inv2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_inv_fast_kernel<HLEN, TX, TY, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    dim3 grid(8 * chunk, batch);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_fast_kernel<HLEN, TX, TY, NT>), grid, dim3(NT), lds, 0, a); });
    const double bytes = 8.0 * batch * N * N;
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  %7.1f GB/s (algorithmic)\n", tag, N, batch, lds, us, bytes / us / 1e3);
}

template <int HLEN, int TX2, int TY2, int NT>
static void bench_fwd_pyr2(const char* tag, const float* in, float* out, int N, int batch) {
    if (skip(tag)) return;
    FwdPyr2Args a;
    const long long n1 = (long long)batch * (N / 2) * (N / 2), n2 = (long long)batch * (N / 4) * (N / 4);
    a.in = in; a.H1 = out; a.V1 = out + n1; a.D1 = out + 2 * n1;
    float* l2 = out + 3 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.N0r = N; a.N0c = N;
    a.in_bstride = (long long)N * N; a.l1_bstride = (long long)(N / 2) * (N / 2); a.l2_bstride = (long long)(N / 4) * (N / 4);
    a.tiles_x = (N / 4 + TX2 - 1) / TX2; a.tiles_y = (N / 4 + TY2 - 1) / TY2;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[i]; a.fb.t[i].y = DB4_HI[i]; }
    const size_t lds = (size_t)Pyr2Geom<HLEN, TX2, TY2>::LDS_FLOATS * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_fwd_pyr2_kernel<HLEN, TX2, TY2, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_pyr2_kernel<HLEN, TX2, TY2, NT>), dim3(8 * chunk, batch), dim3(NT), lds, 0, a); });
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  (two levels in one launch)\n", tag, N, batch, lds, us);
}

template <int HLEN, int TX, int TY, int NT>
static void bench_inv_pyr2(const char* tag, const float* in, float* out, int N, int batch) {
    if (skip(tag)) return;
    InvPyr2Args a;
    const long long n1 = (long long)batch * (N / 2) * (N / 2), n2 = (long long)batch * (N / 4) * (N / 4);
    a.H1 = in; a.V1 = in + n1; a.D1 = in + 2 * n1;
    const float* l2 = in + 3 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.out = out; a.N0r = N; a.N0c = N;
    a.out_bstride = (long long)N * N; a.l1_bstride = (long long)(N / 2) * (N / 2); a.l2_bstride = (long long)(N / 4) * (N / 4);
    a.tiles_x = (N + 2 * TX - 1) / (2 * TX); a.tiles_y = (N + 2 * TY - 1) / (2 * TY);
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[7 - i]; a.fb.t[i].y = DB4_HI[7 - i]; }
    const size_t lds = (size_t)InvPyr2Geom<HLEN, TX, TY>::LDS_FLOATS * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_inv_pyr2_kernel<HLEN, TX, TY, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_pyr2_kernel<HLEN, TX, TY, NT>), dim3(8 * chunk, batch), dim3(NT), lds, 0, a); });
    printf("%-34s N=%d B=%d lds=%6zu  %8.2f us  (two levels in one launch)\n", tag, N, batch, lds, us);
}

template <int HLEN, int TX2, int NT, int PF, int CHUNK2 = 4>
static void bench_fwd_strip2(const char* tag, const float* in, float* out, int N, int batch, int seg2) {
    if (skip(tag)) return;
    FwdStrip2Args a;
    const long long n1 = (long long)batch * (N / 2) * (N / 2), n2 = (long long)batch * (N / 4) * (N / 4);
    a.in = in; a.H1 = out; a.V1 = out + n1; a.D1 = out + 2 * n1;
    float* l2 = out + 3 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.N0r = N; a.N0c = N; a.seg2 = seg2;
    a.in_bstride = (long long)N * N; a.l1_bstride = (long long)(N / 2) * (N / 2); a.l2_bstride = (long long)(N / 4) * (N / 4);
    a.strips = (N / 4 + TX2 - 1) / TX2; a.segs = (N / 4 + seg2 - 1) / seg2;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[i]; a.fb.t[i].y = DB4_HI[i]; }
    const size_t lds = (size_t)Strip2Geom<HLEN, TX2, CHUNK2>::LDS_FLOATS * sizeof(float);
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dwt2_fwd_strip2_kernel<HLEN, TX2, NT, PF, CHUNK2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_strip2_kernel<HLEN, TX2, NT, PF, CHUNK2>), dim3(a.strips * a.segs, batch), dim3(NT), lds, 0, a); });
    printf("%-24s PF=%d seg2=%3d wgs=%5d N=%d B=%d lds=%6zu  %8.2f us  (two levels in one launch)\n", tag, PF, seg2, a.strips * a.segs, N, batch, lds, us);
}

template <int HLEN, int TX, int NT>
static void bench_inv_strip2(const char* tag, const float* in, float* out, int N, int batch, int seg_rows) {
    if (skip(tag)) return;
    InvStrip2Args a;
    const long long n1 = (long long)batch * (N / 2) * (N / 2), n2 = (long long)batch * (N / 4) * (N / 4);
    a.H1 = in; a.V1 = in + n1; a.D1 = in + 2 * n1;
    const float* l2 = in + 3 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.out = out; a.N0r = N; a.N0c = N; a.seg_rows = seg_rows;
    a.out_bstride = (long long)N * N; a.l1_bstride = (long long)(N / 2) * (N / 2); a.l2_bstride = (long long)(N / 4) * (N / 4);
    a.strips = (N / 2 + TX - 1) / TX; a.segs = (N + seg_rows - 1) / seg_rows;
    memset(&a.fb, 0, sizeof(a.fb));
    for (int i = 0; i < 8; i++) { a.fb.t[i].x = DB4_LO[7 - i]; a.fb.t[i].y = DB4_HI[7 - i]; }
    const size_t lds = (size_t)InvStrip2Geom<HLEN, TX>::LDS_FLOATS * sizeof(float);
    float us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_strip2_kernel<HLEN, TX, NT>), dim3(a.strips * a.segs, batch), dim3(NT), lds, 0, a); });
    printf("%-24s seg_rows=%4d wgs=%5d N=%d B=%d lds=%6zu  %8.2f us  (two levels in one launch)\n", tag, seg_rows, a.strips * a.segs, N, batch, lds, us);
}

// Six dependent copy kernels sized like the launches of one cfg2 step (64, 16, 5, 5, 16, 64 MiB of
// traffic halves), issued as plain stream launches and as one replayed hipGraph: what a launch boundary
// costs on this GPU and whether a graph shortens it.
static void bench_launch_chain(const float* a, float* b) {
    if (skip("CHAIN")) return;
    const long long n4s[6] = {4194304, 1048576, 327680, 327680, 1048576, 4194304};  // float4 groups per kernel
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto chain = [&](hipStream_t s) {
        for (int k = 0; k < 6; k++)
            hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, s, (const float4*)a, (float4*)b, n4s[k]);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    for (int i = 0; i < 5; i++) chain(st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; i++) chain(st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float plain = ms * 1e3f / 50;
    // sum of the kernels alone
    float sum = 0.f;
    for (int k = 0; k < 6; k++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, st, (const float4*)a, (float4*)b, n4s[k]);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms * 1e3f / 50;
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("CHAIN of 6 dependent copies: back-to-back same-size launches sum %.2f us ; chain as stream launches %.2f us ; "
           "as one hipGraph %.2f us\n", sum, plain, ms * 1e3f / 50);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    const int B = argc > 2 ? atoi(argv[2]) : 1;
    if (argc > 3) g_only = argv[3];
    const long long n = (long long)B * N * N;
    float *a, *b;
    CK(hipMalloc((void**)&a, n * sizeof(float)));
    CK(hipMalloc((void**)&b, n * sizeof(float)));
    CK(hipMemset(a, 0, n * sizeof(float)));
    CK(hipMemset(b, 0, n * sizeof(float)));
    {
        std::vector<float> h((size_t)N * N);
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 257.0f;
        for (int i = 0; i < B; i++) CK(hipMemcpy(a + (long long)i * N * N, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const long long n4 = n / 4;
    for (int grid : {1024, 2048, 8192}) {
        float us = time_it([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
        printf("copy4  grid=%5d                    %8.2f us  %7.1f GB/s (r+w)\n", grid, us, 8.0 * n / us / 1e3);
    }
    {
        float us = time_it([&] { hipLaunchKernelGGL(read4, dim3(2048), dim3(256), 0, 0, (const float4*)a, b, n4); });
        printf("read4  grid= 2048                    %8.2f us  %7.1f GB/s (r)\n", us, 4.0 * n / us / 1e3);
        us = time_it([&] { hipLaunchKernelGGL(write4, dim3(2048), dim3(256), 0, 0, (float4*)b, n4); });
        printf("write4 grid= 2048                    %8.2f us  %7.1f GB/s (w)\n", us, 4.0 * n / us / 1e3);
    }
    bench_launch_chain(a, b);
    for (int w : {2, 4, 6, 8, 12}) bench_fwd_stream<8, 64, 8, 256>("STREAM fwd db4 TX64 TY8 NT256", a, b, N, B, w);
    for (int w : {2, 4, 6, 8}) bench_fwd_stream<8, 64, 16, 256>("STREAM fwd db4 TX64 TY16 NT256", a, b, N, B, w);
    for (int w : {2, 4, 8}) bench_fwd_stream<8, 64, 4, 128>("STREAM fwd db4 TX64 TY4 NT128", a, b, N, B, w);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 256, 1, 8>("STRIP2 fwd db4 TX2=32 NT256 CH2=8", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 512, 1, 8>("STRIP2 fwd db4 TX2=32 NT512 CH2=8", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 512, 1, 16>("STRIP2 fwd db4 TX2=32 NT512 CH2=16", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 192, 2>("STRIP2 fwd db4 TX2=32 NT192", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 320, 2>("STRIP2 fwd db4 TX2=32 NT320", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 256, 1>("STRIP2 fwd db4 TX2=32 NT256", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 256, 2>("STRIP2 fwd db4 TX2=32 NT256", a, b, N, B, sg);
    for (int sg : {16, 32, 64, 128}) bench_fwd_strip2<8, 32, 256, 3>("STRIP2 fwd db4 TX2=32 NT256", a, b, N, B, sg);
    for (int sg : {16, 32, 64}) bench_fwd_strip2<8, 64, 512, 2>("STRIP2 fwd db4 TX2=64 NT512", a, b, N, B, sg);
    for (int sg : {16, 32, 64}) bench_fwd_strip2<8, 64, 256, 2>("STRIP2 fwd db4 TX2=64 NT256", a, b, N, B, sg);
    for (int sg : {128, 256, 512}) bench_inv_strip2<8, 64, 256>("STRIP2 inv db4 TX=64 NT256", b, a, N, B, sg);
    for (int sg : {128, 256, 512}) bench_inv_strip2<8, 64, 512>("STRIP2 inv db4 TX=64 NT512", b, a, N, B, sg);
    for (int sg : {128, 256, 512}) bench_inv_strip2<8, 56, 256>("STRIP2 inv db4 TX=56 NT256", b, a, N, B, sg);
    for (int sg : {128, 256, 512}) bench_inv_strip2<8, 120, 512>("STRIP2 inv db4 TX=120 NT512", b, a, N, B, sg);
    for (int sg : {32, 64, 128}) bench_fwd_strip2<8, 28, 256, 2>("STRIP2 fwd db4 TX2=28 NT256", a, b, N, B, sg);
    for (int sg : {32, 64, 128}) bench_fwd_strip2<8, 60, 512, 2>("STRIP2 fwd db4 TX2=60 NT512", a, b, N, B, sg);
    bench_fwd_pyr2<8, 32, 4, 256>("PYR2 fwd db4 TX2=32 TY2=4 NT256", a, b, N, B);
    bench_fwd_pyr2<8, 32, 8, 256>("PYR2 fwd db4 TX2=32 TY2=8 NT256", a, b, N, B);
    bench_fwd_pyr2<8, 32, 8, 512>("PYR2 fwd db4 TX2=32 TY2=8 NT512", a, b, N, B);
    bench_fwd_pyr2<8, 64, 4, 256>("PYR2 fwd db4 TX2=64 TY2=4 NT256", a, b, N, B);
    bench_inv_pyr2<8, 64, 8, 256>("PYR2 inv db4 TX=64 TY=8 NT256", b, a, N, B);
    bench_inv_pyr2<8, 64, 16, 256>("PYR2 inv db4 TX=64 TY=16 NT256", b, a, N, B);
    bench_inv_pyr2<8, 64, 16, 512>("PYR2 inv db4 TX=64 TY=16 NT512", b, a, N, B);
    bench_fwd_fast<8, 64, 16, 512>("FAST fwd db4 TX64 TY16 NT512", a, b, N, B);
    bench_fwd_fast<8, 128, 8, 512>("FAST fwd db4 TX128 TY8 NT512", a, b, N, B);
    bench_fwd_fast<8, 64, 4, 128>("FAST fwd db4 TX64 TY4 NT128", a, b, N, B);
    bench_fwd_fast<8, 32, 8, 128>("FAST fwd db4 TX32 TY8 NT128", a, b, N, B);
    bench_inv_fast<8, 64, 16, 512>("FAST inv db4 TX64 TY16 NT512", b, a, N, B);
    bench_inv_fast<8, 64, 8, 512>("FAST inv db4 TX64 TY8 NT512", b, a, N, B);
    bench_inv_fast<8, 64, 4, 128>("FAST inv db4 TX64 TY4 NT128", b, a, N, B);
    bench_inv_fast<8, 64, 4, 256>("FAST inv db4 TX64 TY4 NT256", b, a, N, B);
    bench_inv_fast<8, 32, 8, 128>("FAST inv db4 TX32 TY8 NT128", b, a, N, B);
    bench_inv_fast<8, 128, 8, 256>("FAST inv db4 TX128 TY8 NT256", b, a, N, B);
    bench_fwd_fast<8, 64, 16, 256>("FAST fwd db4 TX64 TY16 NT256", a, b, N, B);
    bench_fwd_fast<8, 64, 32, 256>("FAST fwd db4 TX64 TY32 NT256", a, b, N, B);
    bench_fwd_fast<8, 64, 32, 512>("FAST fwd db4 TX64 TY32 NT512", a, b, N, B);
    bench_fwd_fast<8, 64, 8, 256>("FAST fwd db4 TX64 TY8  NT256", a, b, N, B);
    bench_fwd_fast<8, 64, 8, 128>("FAST fwd db4 TX64 TY8  NT128", a, b, N, B);
    bench_fwd_fast<8, 128, 16, 512>("FAST fwd db4 TX128 TY16 NT512", a, b, N, B);
    bench_fwd_fast<8, 32, 16, 128>("FAST fwd db4 TX32 TY16 NT128", a, b, N, B);
    bench_fwd_fast<2, 64, 16, 256>("FAST fwd hlen2 TX64 TY16 NT256", a, b, N, B);
    bench_inv_fast<8, 64, 16, 256>("FAST inv db4 TX64 TY16 NT256", b, a, N, B);
    bench_inv_fast<8, 64, 8, 256>("FAST inv db4 TX64 TY8  NT256", b, a, N, B);
    bench_inv_fast<8, 64, 32, 256>("FAST inv db4 TX64 TY32 NT256", b, a, N, B);
    bench_inv_fast<8, 64, 32, 512>("FAST inv db4 TX64 TY32 NT512", b, a, N, B);
    bench_inv_fast<8, 128, 16, 512>("FAST inv db4 TX128 TY16 NT512", b, a, N, B);
    bench_inv_fast<8, 32, 16, 128>("FAST inv db4 TX32 TY16 NT128", b, a, N, B);
    bench_inv_fast<2, 64, 16, 256>("FAST inv hlen2 TX64 TY16 NT256", b, a, N, B);
    // long filters (16 taps): tile shapes of the LDS kernels (run with the name filter "h16")
    for (int w : {2, 3, 4, 6}) bench_fwd_stream<16, 64, 16, 256>("STREAM fwd h16 TX64 TY16 NT256", a, b, N, B, w);
    for (int w : {2, 4, 6}) bench_fwd_stream<16, 64, 8, 256>("STREAM fwd h16 TX64 TY8 NT256", a, b, N, B, w);
    for (int w : {1, 2}) bench_fwd_stream<16, 64, 32, 256>("STREAM fwd h16 TX64 TY32 NT256", a, b, N, B, w);
    bench_fwd_fast<16, 64, 8, 256>("FAST fwd h16 TX64 TY8 NT256", a, b, N, B);
    bench_fwd_fast<16, 64, 16, 256>("FAST fwd h16 TX64 TY16 NT256", a, b, N, B);
    bench_fwd_fast<16, 64, 16, 512>("FAST fwd h16 TX64 TY16 NT512", a, b, N, B);
    bench_fwd_fast<16, 64, 32, 512>("FAST fwd h16 TX64 TY32 NT512", a, b, N, B);
    bench_fwd_fast<16, 128, 16, 512>("FAST fwd h16 TX128 TY16 NT512", a, b, N, B);
    bench_fwd_fast<16, 128, 8, 512>("FAST fwd h16 TX128 TY8 NT512", a, b, N, B);
    bench_inv_fast<16, 64, 8, 256>("FAST inv h16 TX64 TY8 NT256", b, a, N, B);
    bench_inv_fast<16, 64, 16, 256>("FAST inv h16 TX64 TY16 NT256", b, a, N, B);
    bench_inv_fast<16, 64, 16, 512>("FAST inv h16 TX64 TY16 NT512", b, a, N, B);
    bench_inv_fast<16, 64, 32, 512>("FAST inv h16 TX64 TY32 NT512", b, a, N, B);
    bench_inv_fast<16, 128, 16, 512>("FAST inv h16 TX128 TY16 NT512", b, a, N, B);
    bench_inv_fast<16, 128, 8, 512>("FAST inv h16 TX128 TY8 NT512", b, a, N, B);
    bench_inv_fast<16, 128, 8, 256>("FAST inv h16 TX128 TY8 NT256", b, a, N, B);
    bench_fwd<8, 64, 16, 256>("fwd db4 TX64 TY16 NT256", a, b, N, B);
    bench_fwd<8, 64, 32, 256>("fwd db4 TX64 TY32 NT256", a, b, N, B);
    bench_fwd<8, 64, 8, 256>("fwd db4 TX64 TY8  NT256", a, b, N, B);
    bench_fwd<8, 128, 16, 256>("fwd db4 TX128 TY16 NT256", a, b, N, B);
    bench_fwd<8, 64, 16, 128>("fwd db4 TX64 TY16 NT128", a, b, N, B);
    bench_fwd<8, 64, 32, 512>("fwd db4 TX64 TY32 NT512", a, b, N, B);
    bench_fwd<8, 32, 32, 256>("fwd db4 TX32 TY32 NT256", a, b, N, B);
    bench_fwd<2, 64, 16, 256>("fwd haar(taps of db4) TX64 TY16", a, b, N, B);
    bench_inv<8, 64, 16, 256>("inv db4 TX64 TY16 NT256", b, a, N, B);
    bench_inv<8, 64, 32, 256>("inv db4 TX64 TY32 NT256", b, a, N, B);
    bench_inv<8, 64, 8, 256>("inv db4 TX64 TY8  NT256", b, a, N, B);
    bench_inv<8, 128, 16, 256>("inv db4 TX128 TY16 NT256", b, a, N, B);
    bench_inv<8, 32, 32, 256>("inv db4 TX32 TY32 NT256", b, a, N, B);
    CK(hipFree(a));
    CK(hipFree(b));
    return 0;
}
