// wbench.hip -- micro-benchmark of the wave-per-tile level kernels (dwt2_wave_kernels.hpp) next to the
// LDS-tiled kernels they replace and to plain copies of the same byte count (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pypwt_amd/csrc tools/wbench.hip -o tools/bin/wbench
//   tools/bin/wbench [batch]        sweeps N = 4096, 2048, 1024, 512 (db4)
// Every variant is first checked against the LDS-tiled kernel's output on the same input.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "dwt2_fast_kernels.hpp"
#include "dwt2_wave_kernels.hpp"

using namespace pdwt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 12345) *p = 1;
}

static float time_it(const std::function<void()>& fn, int reps = 40, int warm = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; i++) fn();
    CK(hipDeviceSynchronize());
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

static const float DB4_DLO[8] = {-0.010597401784997278f, 0.032883011666982945f, 0.030841381835986965f, -0.18703481171888114f,
                                 -0.02798376941698385f, 0.6308807679295904f, 0.7148465705525415f, 0.23037781330885523f};
static const float DB4_DHI[8] = {-0.23037781330885523f, 0.7148465705525415f, -0.6308807679295904f, -0.02798376941698385f,
                                 0.18703481171888114f, 0.030841381835986965f, -0.032883011666982945f, -0.010597401784997278f};

static void bank(FilterBankI& fb, bool rec) {
    memset(&fb, 0, sizeof(fb));
    for (int i = 0; i < 8; i++) {
        // reconstruction filters of an orthogonal wavelet: the decomposition filters reversed
        fb.t[i].x = rec ? DB4_DLO[7 - i] : DB4_DLO[i];
        fb.t[i].y = rec ? DB4_DHI[7 - i] : DB4_DHI[i];
    }
}

static double max_abs_diff(const float* d_a, const float* d_b, long long n) {
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), d_a, n * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_b, n * sizeof(float), hipMemcpyDeviceToHost));
    double m = 0;
    for (long long i = 0; i < n; i++) {
        const double d = std::fabs((double)a[i] - (double)b[i]);
        if (!(d <= m)) m = d;  // also catches NaN
    }
    return m;
}

template <int NT>
static void launch_fwd_wave(const FwdWaveArgs& a, int batch) {
    const int nblk = (a.strips * a.segs + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL((dwt2_fwd_wave_kernel<8, false, NT>), dim3(8 * ((nblk + 7) / 8), batch), dim3(NT), 0, 0, a);
}
static void launch_fwd2_wave(const FwdWave2Args& a, int batch) {
    const int nblk = (a.strips * a.segs + 3) / 4;
    hipLaunchKernelGGL((dwt2_fwd2_wave_kernel<8, 256>), dim3(8 * ((nblk + 7) / 8), batch), dim3(256), 0, 0, a);
}
template <int NT>
static void launch_inv_wave(const InvWaveArgs& a, int batch) {
    const int nblk = (a.strips * a.segs + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL((dwt2_inv_wave_kernel<8, false, NT>), dim3(8 * ((nblk + 7) / 8), batch), dim3(NT), 0, 0, a);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1;
    const int nmin = argc > 2 ? atoi(argv[2]) : 512;  // smallest level input size swept
    const int NMAX = 4096;
    const long long nmax = (long long)B * NMAX * NMAX;
    float *img, *co, *co2, *img2;
    CK(hipMalloc((void**)&img, nmax * sizeof(float)));
    CK(hipMalloc((void**)&img2, nmax * sizeof(float)));
    CK(hipMalloc((void**)&co, nmax * sizeof(float)));
    CK(hipMalloc((void**)&co2, nmax * sizeof(float)));
    {
        std::vector<float> h((size_t)nmax);
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 257.0f;
        CK(hipMemcpy(img, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; batch %d; db4 (hlen 8); back-to-back launches between two HIP events\n", prop.name,
           prop.multiProcessorCount, B);
    {
        float us = time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int*)nullptr); }, 200);
        printf("EMPTY kernel, 256 workgroups: %.2f us per launch\n", us);
    }
    for (int N : {4096, 2048, 1024, 512}) {
        if (N < nmin) continue;
        const long long n = (long long)B * N * N, q = n / 4;
        const double bytes = 8.0 * n;
        printf("---- N = %d (level input %d x %d, %.1f MB moved per direction)\n", N, N, N, bytes / 1e6);
        for (int grid : {256, 1024, 2048}) {
            float us = time_it([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, (const float4*)img, (float4*)co, n / 4); });
            printf("copy4 grid=%4d                          %8.2f us  %7.1f GB/s\n", grid, us, bytes / us / 1e3);
        }
        // ---------------- forward: reference = LDS-tiled kernel
        Fwd2DFastArgs f;
        f.in = img; f.A = co; f.H = co + q; f.V = co + 2 * q; f.D = co + 3 * q;
        f.Nr = N; f.Nc = N; f.Nr2 = N / 2; f.Nc2 = N / 2;
        f.in_bstride = (long long)N * N; f.out_bstride = (long long)(N / 2) * (N / 2);
        f.tiles_x = (N / 2 + 63) / 64; f.tiles_y = (N / 2 + 7) / 8;
        bank(f.fb, false);
        {
            constexpr size_t lds = (size_t)fwd2d_fast_lds_floats<8, 64, 8>() * sizeof(float);
            const int chunk = (f.tiles_x * f.tiles_y + 7) / 8;
            float us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_fast_kernel<8, 64, 8, 256>), dim3(8 * chunk, B), dim3(256), lds, 0, f); });
            printf("LDS-tile fwd 64x8 NT256                   %8.2f us  %7.1f GB/s\n", us, bytes / us / 1e3);
            {
                int nwg = 256 * 6;
                if (nwg > 8 * chunk * B) nwg = 8 * chunk * B;
                nwg = (nwg + 7) & ~7;
                us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_fast_stream_kernel<8, 64, 8, 256>), dim3(nwg), dim3(256), lds, 0, f, B); });
                printf("LDS-tile fwd 64x8 NT256 streaming         %8.2f us  %7.1f GB/s\n", us, bytes / us / 1e3);
            }
            if (N / 2 >= 64) {
                Fwd2DFastArgs g = f;
                g.tiles_y = (N / 2 + 31) / 32;
                constexpr size_t lds2 = (size_t)fwd2d_fast_lds_floats<8, 64, 32>() * sizeof(float);
                const int chunk2 = (g.tiles_x * g.tiles_y + 7) / 8;
                us = time_it([&] { hipLaunchKernelGGL((dwt2_fwd_fast_kernel<8, 64, 32, 512>), dim3(8 * chunk2, B), dim3(512), lds2, 0, g); });
                printf("LDS-tile fwd 64x32 NT512                  %8.2f us  %7.1f GB/s\n", us, bytes / us / 1e3);
                hipLaunchKernelGGL((dwt2_fwd_fast_kernel<8, 64, 8, 256>), dim3(8 * chunk, B), dim3(256), lds, 0, f);
            }
        }
        FwdWaveArgs w;
        w.in = img; w.A = co2; w.H = co2 + q; w.V = co2 + 2 * q; w.D = co2 + 3 * q;
        w.Nr = N; w.Nc = N; w.Nr2 = N / 2; w.Nc2 = N / 2;
        w.in_bstride = f.in_bstride; w.out_bstride = f.out_bstride;
        w.strips = (N + 255) / 256;
        bank(w.fb, false);
        for (int seg : {4, 8, 16, 32, 64}) {
            if (seg > N / 2) continue;
            w.seg_out = seg; w.segs = (N / 2 + seg - 1) / seg;
            CK(hipMemset(co2, 0xff, n * sizeof(float)));
            launch_fwd_wave<256>(w, B);
            const double err = max_abs_diff(co, co2, n);
            float us = time_it([&] { launch_fwd_wave<256>(w, B); });
            float us1 = time_it([&] { launch_fwd_wave<64>(w, B); });
            float us2 = time_it([&] { launch_fwd_wave<128>(w, B); });
            printf("WAVE fwd seg_out=%2d waves=%6d   NT256 %8.2f us %7.1f GB/s | NT128 %8.2f | NT64 %8.2f   max|diff| %.3g\n",
                   seg, w.strips * w.segs * B, us, bytes / us / 1e3, us2, us1, err);
        }
        // ---------------- two forward levels per wavefront vs. the two single-level launches
        if (N >= 64) {
            // reference for level 2: the LDS-tiled kernel on the A band of level 1 (co) -> img2 (4 planes of (N/4)^2)
            const long long q2 = n / 16;
            Fwd2DFastArgs f2 = f;
            f2.in = co; f2.A = img2; f2.H = img2 + q2; f2.V = img2 + 2 * q2; f2.D = img2 + 3 * q2;
            f2.Nr = N / 2; f2.Nc = N / 2; f2.Nr2 = N / 4; f2.Nc2 = N / 4;
            f2.in_bstride = f.out_bstride; f2.out_bstride = (long long)(N / 4) * (N / 4);
            f2.tiles_x = (N / 4 + 63) / 64; f2.tiles_y = (N / 4 + 7) / 8;
            constexpr size_t lds = (size_t)fwd2d_fast_lds_floats<8, 64, 8>() * sizeof(float);
            hipLaunchKernelGGL((dwt2_fwd_fast_kernel<8, 64, 8, 256>), dim3(8 * ((f2.tiles_x * f2.tiles_y + 7) / 8), B), dim3(256), lds, 0, f2);
            FwdWave2Args w2;
            float* l2out = co2 + 3 * q;  // A2 H2 V2 D2 packed into the (unused) fourth plane of co2
            w2.in = img; w2.H1 = co2; w2.V1 = co2 + q; w2.D1 = co2 + 2 * q;
            w2.A2 = l2out; w2.H2 = l2out + q2; w2.V2 = l2out + 2 * q2; w2.D2 = l2out + 3 * q2;
            w2.N0r = N; w2.N0c = N;
            w2.in_bstride = f.in_bstride; w2.l1_bstride = f.out_bstride; w2.l2_bstride = f2.out_bstride;
            w2.strips = (N + 239) / 240;
            bank(w2.fb, false);
            for (int seg2 : {8, 12, 24, 32, 64, 128, 256, 512}) {
                if (seg2 > N / 4) continue;
                w2.seg2_out = seg2; w2.segs = (N / 4 + seg2 - 1) / seg2;
                CK(hipMemset(co2, 0xff, n * sizeof(float)));
                launch_fwd2_wave(w2, B);
                // co: A H V D ; co2: H1 V1 D1 [A2 H2 V2 D2]
                const double e1 = fmax(fmax(max_abs_diff(co + q, co2, q), max_abs_diff(co + 2 * q, co2 + q, q)),
                                       max_abs_diff(co + 3 * q, co2 + 2 * q, q));
                const double e2 = max_abs_diff(img2, l2out, 4 * q2);
                float us = time_it([&] { launch_fwd2_wave(w2, B); });
                printf("WAVE2 fwd (L+1 fused) seg2_out=%2d waves=%6d  %8.2f us  %7.1f GB/s (8 B/sample)   max|diff| L1 %.3g L2 %.3g\n",
                       seg2, w2.strips * w2.segs * B, us, bytes / us / 1e3, e1, e2);
            }
        }
        // ---------------- inverse: reference = LDS-tiled kernel on the coefficients of the forward
        Inv2DFastArgs v;
        v.A = co; v.H = co + q; v.V = co + 2 * q; v.D = co + 3 * q; v.out = img2;
        v.Nrc = N / 2; v.Ncc = N / 2; v.Nr = N; v.Nc = N;
        v.in_bstride = f.out_bstride; v.out_bstride = f.in_bstride;
        v.tiles_x = (N + 127) / 128; v.tiles_y = (N + 15) / 16;
        bank(v.fb, true);
        {
            constexpr size_t lds = (size_t)inv2d_fast_lds_floats<8, 64, 8>() * sizeof(float);
            const int chunk = (v.tiles_x * v.tiles_y + 7) / 8;
            float us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_fast_kernel<8, 64, 8, 256>), dim3(8 * chunk, B), dim3(256), lds, 0, v); });
            printf("LDS-tile inv 64x8 NT256                   %8.2f us  %7.1f GB/s   round trip max|x - inv(fwd(x))| %.3g\n", us,
                   bytes / us / 1e3, max_abs_diff(img, img2, n));
            if (N / 2 >= 128) {
                Inv2DFastArgs g = v;
                g.tiles_x = (N + 255) / 256; g.tiles_y = (N + 31) / 32;
                constexpr size_t lds2 = (size_t)inv2d_fast_lds_floats<8, 128, 16>() * sizeof(float);
                const int chunk2 = (g.tiles_x * g.tiles_y + 7) / 8;
                us = time_it([&] { hipLaunchKernelGGL((dwt2_inv_fast_kernel<8, 128, 16, 512>), dim3(8 * chunk2, B), dim3(512), lds2, 0, g); });
                printf("LDS-tile inv 128x16 NT512                 %8.2f us  %7.1f GB/s\n", us, bytes / us / 1e3);
            }
        }
        InvWaveArgs iw;
        iw.A = v.A; iw.H = v.H; iw.V = v.V; iw.D = v.D; iw.out = img2;
        iw.Nrc = N / 2; iw.Ncc = N / 2; iw.Nr = N; iw.Nc = N;
        iw.in_bstride = v.in_bstride; iw.out_bstride = v.out_bstride;
        iw.strips = (N / 2 + 127) / 128;
        bank(iw.fb, true);
        for (int d = 0; d < 4; d++) {
            iw.pl[d].x = iw.fb.t[6 - 2 * d].x; iw.pl[d].y = iw.fb.t[7 - 2 * d].x;
            iw.ph[d].x = iw.fb.t[6 - 2 * d].y; iw.ph[d].y = iw.fb.t[7 - 2 * d].y;
        }
        for (int seg : {4, 8, 16, 32, 64}) {
            if (seg > N / 2) continue;
            iw.seg_pairs = seg; iw.segs = (N / 2 + seg - 1) / seg;
            CK(hipMemset(img2, 0xff, n * sizeof(float)));
            launch_inv_wave<256>(iw, B);
            const double err = max_abs_diff(img, img2, n);
            float us = time_it([&] { launch_inv_wave<256>(iw, B); });
            float us1 = time_it([&] { launch_inv_wave<64>(iw, B); });
            float us2 = time_it([&] { launch_inv_wave<128>(iw, B); });
            printf("WAVE inv seg_pairs=%2d waves=%6d NT256 %8.2f us %7.1f GB/s | NT128 %8.2f | NT64 %8.2f   round trip %.3g\n",
                   seg, iw.strips * iw.segs * B, us, bytes / us / 1e3, us2, us1, err);
        }
    }
    return 0;
}
