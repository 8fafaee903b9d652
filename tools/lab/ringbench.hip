// ringbench.hip -- timing of the register-ring level kernels next to the LDS tiles they replace, same harness, same box
// (developer tool): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pypwt_amd/csrc [-DPDWT_RING_DIAG=<mask>] tools/ringbench.hip -o tools/bin/ringbench
//   ringbench [N=4096] [seg=16] [batch=1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "dwt2_fast_kernels.hpp"
#include "dwt2_ring_kernels.hpp"
using namespace pdwt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef RB_HLEN
#define RB_HLEN 16
#endif
#ifndef RB_CPL
#define RB_CPL 4
#endif
#ifndef RB_MINB
#define RB_MINB 3
#endif
#ifndef RB_MINB_INV
#define RB_MINB_INV 2
#endif

__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xffff) * (1.0f / 256.0f);
    }
}

static float time_it(const std::function<void()>& fn, int reps = 60) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; i++) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, seg = argc > 2 ? atoi(argv[2]) : 16, B = argc > 3 ? atoi(argv[3]) : 1;
    float *img, *coef;
    const size_t plane = (size_t)N * N, q = plane / 4;
    CK(hipMalloc(&img, plane * B * 4));
    CK(hipMalloc(&coef, plane * B * 4));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, img, plane * B, 1u);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, coef, plane * B, 2u);
    float *A = coef, *H = coef + q * B, *V = coef + 2 * q * B, *D = coef + 3 * q * B;
    FilterBankI fb;
    for (int i = 0; i < kMaxTaps; i++) { fb.t[i].x = 0.05f * ((i * 7) % 5 - 2); fb.t[i].y = -0.04f * ((i * 3) % 7 - 3); }
    constexpr int HL = RB_HLEN;

    using GF = FwdRingGeom<HL, RB_CPL>;
    FwdWaveArgs fa;
    fa.in = img; fa.A = A; fa.H = H; fa.V = V; fa.D = D;
    fa.Nr = N; fa.Nc = N; fa.Nr2 = N / 2; fa.Nc2 = N / 2;
    fa.in_bstride = plane; fa.out_bstride = q;
    fa.strips = (N + GF::W - 1) / GF::W; fa.seg_out = seg; fa.segs = (fa.Nr2 + seg - 1) / seg;
    fa.fb = fb;
    const int fblk = (fa.strips * fa.segs + 3) / 4;
    auto ring_fwd = [&] { hipLaunchKernelGGL((dwt2_fwd_ring_kernel<HL, RB_CPL, 256, RB_MINB>), dim3(8 * ((fblk + 7) / 8), B), dim3(256), 4 * GF::LDS_REALS * sizeof(float), 0, fa); };

    using GI = InvRingGeom<HL, RB_CPL>;
    InvRingArgs ia;
    ia.A = A; ia.H = H; ia.V = V; ia.D = D; ia.out = img;
    ia.Nrc = N / 2; ia.Ncc = N / 2; ia.Nr = N; ia.Nc = N;
    ia.in_bstride = q; ia.out_bstride = plane;
    ia.strips = (N / 2 + GI::WC - 1) / GI::WC; ia.seg_pairs = seg; ia.segs = (N / 2 + seg - 1) / seg;
    ia.fb = fb;
    for (int d = 0; d < HL / 2; d++) { ia.pl[d].x = fb.t[HL - 2 - 2 * d].x; ia.pl[d].y = fb.t[HL - 1 - 2 * d].x; ia.ph[d].x = fb.t[HL - 2 - 2 * d].y; ia.ph[d].y = fb.t[HL - 1 - 2 * d].y; }
    const int iblk = (ia.strips * ia.segs + 3) / 4;
    auto ring_inv = [&] { hipLaunchKernelGGL((dwt2_inv_ring_kernel<HL, RB_CPL, 256, RB_MINB_INV>), dim3(8 * ((iblk + 7) / 8), B), dim3(256), 4 * GI::LDS_REALS * sizeof(float), 0, ia); };

    // the LDS tiles the plan launches for this length (launch_dwt2_fast.hip: 64 x 16 outputs, 512 threads)
    constexpr int TX = 64, TY = 16, NT = 512;
    Fwd2DFastArgs tf;
    tf.in = img; tf.A = A; tf.H = H; tf.V = V; tf.D = D; tf.Nr = N; tf.Nc = N; tf.Nr2 = N / 2; tf.Nc2 = N / 2;
    tf.in_bstride = plane; tf.out_bstride = q; tf.tiles_x = (N / 2 + TX - 1) / TX; tf.tiles_y = (N / 2 + TY - 1) / TY; tf.fb = fb;
    constexpr size_t flds = (size_t)fwd2d_fast_lds_floats<HL, TX, TY>() * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt2_fwd_fast_kernel<HL, TX, TY, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    const int fchunk = (tf.tiles_x * tf.tiles_y + 7) / 8;
    auto tile_fwd = [&] { hipLaunchKernelGGL((dwt2_fwd_fast_kernel<HL, TX, TY, NT>), dim3(8 * fchunk, B), dim3(NT), flds, 0, tf); };
    Inv2DFastArgs ti;
    ti.A = A; ti.H = H; ti.V = V; ti.D = D; ti.out = img; ti.Nrc = N / 2; ti.Ncc = N / 2; ti.Nr = N; ti.Nc = N;
    ti.in_bstride = q; ti.out_bstride = plane; ti.tiles_x = (N + 2 * TX - 1) / (2 * TX); ti.tiles_y = (N + 2 * TY - 1) / (2 * TY); ti.fb = fb;
    constexpr size_t ilds = (size_t)inv2d_fast_lds_floats<HL, TX, TY>() * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt2_inv_fast_kernel<HL, TX, TY, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds));
    const int ichunk = (ti.tiles_x * ti.tiles_y + 7) / 8;
    auto tile_inv = [&] { hipLaunchKernelGGL((dwt2_inv_fast_kernel<HL, TX, TY, NT>), dim3(8 * ichunk, B), dim3(NT), ilds, 0, ti); };

    printf("hlen=%d cpl=%d minb=%d/%d nr=%d/%d diag=%d N=%d seg=%d B=%d (%d fwd wavefronts)\n", HL, RB_CPL, RB_MINB, RB_MINB_INV, GF::NR, GI::NR, PDWT_RING_DIAG, N,
           seg, B, fa.strips * fa.segs * B);
    for (int rep = 0; rep < 2; rep++) {
        printf("  tile fwd %6.2f  tile inv %6.2f  tile fwd+inv %6.2f | ring fwd %6.2f  ring inv %6.2f  ring fwd+inv %6.2f us\n", time_it(tile_fwd), time_it(tile_inv),
               time_it([&] { tile_fwd(); tile_inv(); }), time_it(ring_fwd), time_it(ring_inv), time_it([&] { ring_fwd(); ring_inv(); }));
    }
    return 0;
}
