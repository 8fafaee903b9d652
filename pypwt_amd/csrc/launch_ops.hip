// launch_ops.hip -- launchers of the streaming coefficient operators (gfx950).
#include "launch.hpp"
#include "launch_util.hpp"
#include "ops_kernels.hpp"

namespace pdwt {

hipError_t launch_ew(int op, float* p, long long n, float b, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    const int grid = stream_grid(n4, 256);
    float4* p4 = reinterpret_cast<float4*>(p);
    switch (op) {
        case EW_SOFT: hipLaunchKernelGGL((ew_kernel<EW_SOFT>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_HARD: hipLaunchKernelGGL((ew_kernel<EW_HARD>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_LINF: hipLaunchKernelGGL((ew_kernel<EW_LINF>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_SCALE: hipLaunchKernelGGL((ew_kernel<EW_SCALE>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_group_soft(float* d0, float* d1, float* d2, float* ap, long long n, float beta, int nb,
                             hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_soft_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, d0, d1, d2, ap, n, beta, nb);
    return hipGetLastError();
}

hipError_t launch_axpy(float* dst, const float* src, long long n, float alpha, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, s, reinterpret_cast<float4*>(dst),
                       reinterpret_cast<const float4*>(src), n4, alpha);
    return hipGetLastError();
}

hipError_t launch_norms(const float* p, long long n, double* out2, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(norms_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(p), n4, out2);
    return hipGetLastError();
}

hipError_t launch_circshift(const float* in, float* out, int batch, int Nr, int Nc, int sr, int sc,
                            hipStream_t s) {
    const long long plane = (long long)Nr * Nc;
    hipLaunchKernelGGL(circshift_kernel, dim3((unsigned)cdivll(plane, 256), batch), dim3(256), 0, s, in, out, Nr,
                       Nc, sr, sc);
    return hipGetLastError();
}

hipError_t launch_fill_hash(float* x, long long n, uint32_t seed, float scale, long long index_offset,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_hash_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, x, n, seed, scale,
                       index_offset);
    return hipGetLastError();
}

}  // namespace pdwt
