// longbench.hip -- timing of the strip-streaming long-filter level kernels (dwt2_long_kernels.hpp) next to the LDS tiles they
// replace, same harness, same box, results cross-checked (developer tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pypwt_amd/csrc -DLB_HLEN=40 [-DLB_TXC=64 -DLB_TY=16 -DLB_NT=256 -DLB_KB=2
//         -DLB_M=4 -DLB_XB=1 -DLB_MINB=2  and the same with the prefix LBF_ for the forward] tools/longbench.hip -o tools/bin/longbench_40
//   longbench [N=4096] [seg=0: 512 workgroups] [batch=1]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "dwt2_fast_kernels.hpp"
#include "dwt2_long_kernels.hpp"
using namespace pdwt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef LB_HLEN
#define LB_HLEN 40
#endif
#ifndef LB_TXC
#define LB_TXC 64
#endif
#ifndef LB_TY
#define LB_TY 16
#endif
#ifndef LB_NT
#define LB_NT 256
#endif
#ifndef LB_KB
#define LB_KB 4
#endif
#ifndef LB_M
#define LB_M 4
#endif
#ifndef LB_XB
#define LB_XB 1
#endif
#ifndef LB_MINB
#define LB_MINB 2
#endif
#ifndef LBF_TXC
#define LBF_TXC 64
#endif
#ifndef LBF_TY
#define LBF_TY 16
#endif
#ifndef LBF_NT
#define LBF_NT 256
#endif
#ifndef LBF_KB
#define LBF_KB 2
#endif
#ifndef LBF_M
#define LBF_M 4
#endif
#ifndef LBF_XB
#define LBF_XB 1
#endif
#ifndef LBF_MINB
#define LBF_MINB 2
#endif

#if PDWT_LONG_DIAG & 32
__device__ unsigned long long pdwt::pdwt_long_prof[8 * 65536];
#endif
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xffff) * (1.0f / 65536.0f) - 0.5f;
    }
}

static float time_it(const std::function<void()>& fn, int reps = 40) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

static double max_diff(const float* a, const float* b, size_t n, double* scale) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double d = 0, s = 0;
    for (size_t i = 0; i < n; i++) {
        const double e = std::fabs((double)ha[i] - (double)hb[i]);
        if (!(e <= d)) d = e;  // NaN propagates
        if (std::fabs(hb[i]) > s) s = std::fabs(hb[i]);
    }
    *scale = s;
    return d;
}

static int pick_seg(int rows, int strips, int batch, int ty, int hint) {
    if (hint > 0) return (hint + ty - 1) / ty * ty;
    long long units = (long long)strips * batch;
    int segs = (int)((512 + units - 1) / units);  // two workgroups per CU
    if (segs < 1) segs = 1;
    int seg = (rows + segs - 1) / segs;
    seg = (seg + ty - 1) / ty * ty;
    return seg < ty ? ty : seg;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, seg_hint = argc > 2 ? atoi(argv[2]) : 0, B = argc > 3 ? atoi(argv[3]) : 1;
    float *img, *img2, *coef, *coef2;
    const size_t plane = (size_t)N * N, q = plane / 4;
    CK(hipMalloc(&img, plane * B * 4)); CK(hipMalloc(&img2, plane * B * 4));
    CK(hipMalloc(&coef, plane * B * 4)); CK(hipMalloc(&coef2, plane * B * 4));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, img, plane * B, 1u);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, coef, plane * B, 2u);
    float *A = coef, *H = coef + q * B, *V = coef + 2 * q * B, *D = coef + 3 * q * B;
    float *A2 = coef2, *H2 = coef2 + q * B, *V2 = coef2 + 2 * q * B, *D2 = coef2 + 3 * q * B;
    FilterBankI fb;
    for (int i = 0; i < kMaxTaps; i++) { fb.t[i].x = 0.05f * ((i * 7) % 5 - 2) + 0.01f * i; fb.t[i].y = -0.04f * ((i * 3) % 7 - 3) - 0.007f * i; }
    constexpr int HL = LB_HLEN;

    // ---- inverse: coef -> img (long) / img2 (tile)
    using GI = InvLongGeom<HL, LB_TXC, LB_TY>;
    InvLongArgs ia;
    ia.A = A; ia.H = H; ia.V = V; ia.D = D; ia.out = img;
    ia.Nrc = N / 2; ia.Ncc = N / 2; ia.Nr = N; ia.Nc = N; ia.in_bstride = q; ia.out_bstride = plane;
    ia.strips = (N / 2 + LB_TXC - 1) / LB_TXC;
    ia.seg = pick_seg(N / 2, ia.strips, B, LB_TY, seg_hint);
    ia.segs = (N / 2 + ia.seg - 1) / ia.seg;
    {
        float lo[kMaxTaps], hi[kMaxTaps];
        for (int i = 0; i < kMaxTaps; i++) { lo[i] = fb.t[i].x; hi[i] = fb.t[i].y; }
        long_syn_tables<HL>(ia, lo, hi);
    }
    constexpr size_t ilds_long = (size_t)GI::LDS_REALS * 4;
    auto kinv = &dwt2_inv_long_kernel<HL, LB_TXC, LB_TY, LB_NT, LB_KB, LB_M, LB_XB, LB_MINB>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kinv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds_long));
    const int iblk = 8 * ((ia.strips * ia.segs + 7) / 8);
    auto long_inv = [&] { hipLaunchKernelGGL(kinv, dim3(iblk, B), dim3(LB_NT), ilds_long, 0, ia); };

    // the tile shapes the plan launches for this length (launch_dwt2_fast.hip)
    constexpr int ITX = 32, ITY = (HL == 26 || HL == 30) ? 16 : 8, INT = (HL == 26 || HL == 30) ? 512 : 256;
    Inv2DFastArgs ti;
    ti.A = A; ti.H = H; ti.V = V; ti.D = D; ti.out = img2; ti.Nrc = N / 2; ti.Ncc = N / 2; ti.Nr = N; ti.Nc = N;
    ti.in_bstride = q; ti.out_bstride = plane; ti.tiles_x = (N + 2 * ITX - 1) / (2 * ITX); ti.tiles_y = (N + 2 * ITY - 1) / (2 * ITY); ti.fb = fb;
    constexpr size_t ilds = (size_t)inv2d_fast_lds_floats<HL, ITX, ITY>() * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt2_inv_fast_kernel<HL, ITX, ITY, INT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds));
    const int ichunk = (ti.tiles_x * ti.tiles_y + 7) / 8;
    auto tile_inv = [&] { hipLaunchKernelGGL((dwt2_inv_fast_kernel<HL, ITX, ITY, INT>), dim3(8 * ichunk, B), dim3(INT), ilds, 0, ti); };

    // ---- forward: img2 (the tile inverse's output) -> coef2 (long) / coef (tile, overwritten LAST)
    using GF = FwdLongGeom<HL, LBF_TXC, LBF_TY>;
    FwdLongArgs fa;
    fa.in = img2; fa.A = A2; fa.H = H2; fa.V = V2; fa.D = D2;
    fa.Nr = N; fa.Nc = N; fa.Nr2 = N / 2; fa.Nc2 = N / 2; fa.in_bstride = plane; fa.out_bstride = q;
    fa.strips = (N / 2 + LBF_TXC - 1) / LBF_TXC;
    fa.seg = pick_seg(N / 2, fa.strips, B, LBF_TY, seg_hint);
    fa.segs = (N / 2 + fa.seg - 1) / fa.seg;
    fa.fb = fb;
    constexpr size_t flds_long = (size_t)GF::LDS_REALS * 4;
    auto kfwd = &dwt2_fwd_long_kernel<HL, LBF_TXC, LBF_TY, LBF_NT, LBF_KB, LBF_M, LBF_XB, LBF_MINB>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds_long));
    const int fblk = 8 * ((fa.strips * fa.segs + 7) / 8);
    auto long_fwd = [&] { hipLaunchKernelGGL(kfwd, dim3(fblk, B), dim3(LBF_NT), flds_long, 0, fa); };

    constexpr int FTX = 32, FTY = 32, FNT = 512;
    float* T = nullptr;
    CK(hipMalloc(&T, plane * B * 4));
    float *TA = T, *TH = T + q * B, *TV = T + 2 * q * B, *TD = T + 3 * q * B;
    Fwd2DFastArgs tf;
    tf.in = img2; tf.A = TA; tf.H = TH; tf.V = TV; tf.D = TD; tf.Nr = N; tf.Nc = N; tf.Nr2 = N / 2; tf.Nc2 = N / 2;
    tf.in_bstride = plane; tf.out_bstride = q; tf.tiles_x = (N / 2 + FTX - 1) / FTX; tf.tiles_y = (N / 2 + FTY - 1) / FTY; tf.fb = fb;
    constexpr size_t flds = (size_t)fwd2d_fast_lds_floats<HL, FTX, FTY>() * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt2_fwd_fast_kernel<HL, FTX, FTY, FNT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    const int fchunk = (tf.tiles_x * tf.tiles_y + 7) / 8;
    auto tile_fwd = [&] { hipLaunchKernelGGL((dwt2_fwd_fast_kernel<HL, FTX, FTY, FNT>), dim3(8 * fchunk, B), dim3(FNT), flds, 0, tf); };

    printf("hlen=%d N=%d B=%d | inv <TXC %d TY %d NT %d KB %d M %d XB %d minb %d> lds %zu KB seg %d (%d wg) | fwd <TXC %d TY %d NT %d KB %d M %d XB %d minb %d> lds %zu KB seg %d (%d wg)\n",
           HL, N, B, LB_TXC, LB_TY, LB_NT, LB_KB, LB_M, LB_XB, LB_MINB, ilds_long / 1024, ia.seg, ia.strips * ia.segs * B,
           LBF_TXC, LBF_TY, LBF_NT, LBF_KB, LBF_M, LBF_XB, LBF_MINB, flds_long / 1024, fa.seg, fa.strips * fa.segs * B);
    // correctness: the two inverses on the same coefficients, the two forwards on the same image
    CK(hipMemset(img, 0xff, plane * B * 4)); CK(hipMemset(img2, 0xff, plane * B * 4));
    long_inv(); tile_inv();
    CK(hipDeviceSynchronize());
    double sc;
    double d = max_diff(img, img2, plane * B, &sc);
    printf("  inverse: max |long - tile| = %.3g (scale %.3g)%s\n", d, sc, d <= 2e-5 * sc ? "" : "  MISMATCH");
    CK(hipMemset(coef2, 0xff, plane * B * 4)); CK(hipMemset(T, 0xff, plane * B * 4));
    long_fwd(); tile_fwd();
    CK(hipDeviceSynchronize());
    d = max_diff(coef2, T, plane * B, &sc);
    printf("  forward: max |long - tile| = %.3g (scale %.3g)%s\n", d, sc, d <= 2e-5 * sc ? "" : "  MISMATCH");
#if PDWT_LONG_DIAG & 32
    {
        long_inv();
        CK(hipDeviceSynchronize());
        const int nw = iblk * B * (LB_NT / 64);
        std::vector<unsigned long long> h(8 * (size_t)nw);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(pdwt::pdwt_long_prof), h.size() * 8));
        double s[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tmin = ~0ull, tmax = 0;
        int cnt = 0;
        for (int w = 0; w < nw; w++) {
            if (!h[8 * w + 7]) continue;
            for (int k = 0; k < 6; k++) s[k] += (double)h[8 * w + k];
            if (h[8 * w + 6] < tmin) tmin = h[8 * w + 6];
            if (h[8 * w + 6] + h[8 * w + 5] > tmax) tmax = h[8 * w + 6] + h[8 * w + 5];
            cnt++;
        }
        printf("  inverse phase clocks per wavefront (%d wavefronts, %llu steps): rows %.0f  barrier %.0f  cols+stores %.0f  carry/stage/issue %.0f  barrier %.0f | loop %.0f ticks; first start to last end %llu ticks\n",
               cnt, h[7], s[0] / cnt, s[1] / cnt, s[2] / cnt, s[3] / cnt, s[4] / cnt, s[5] / cnt, tmax - tmin);
    }
#endif
    for (int rep = 0; rep < 2; rep++) {
        const float a = time_it(tile_fwd), b = time_it(tile_inv), c = time_it(long_fwd), e = time_it(long_inv);
        printf("  tile fwd %7.2f  tile inv %7.2f | long fwd %7.2f  long inv %7.2f us\n", a, b, c, e);
    }
    return 0;
}
