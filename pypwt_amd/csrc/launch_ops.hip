// launch_ops.hip -- launchers of the streaming coefficient operators (gfx950).
#include "launch.hpp"
#include "launch_util.hpp"
#include "ops_kernels.hpp"

namespace pdwt {

hipError_t launch_ew(int op, real_t* p, long long n, real_t b, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    const int grid = stream_grid(n4, 256);
    real4_t* p4 = reinterpret_cast<real4_t*>(p);
    switch (op) {
        case EW_SOFT: hipLaunchKernelGGL((ew_kernel<EW_SOFT>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_HARD: hipLaunchKernelGGL((ew_kernel<EW_HARD>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_LINF: hipLaunchKernelGGL((ew_kernel<EW_LINF>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        case EW_SCALE: hipLaunchKernelGGL((ew_kernel<EW_SCALE>), dim3(grid), dim3(256), 0, s, p4, n4, b); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_group_soft(real_t* d0, real_t* d1, real_t* d2, real_t* ap, long long n, real_t beta, int nb,
                             hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_soft_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, d0, d1, d2, ap, n, beta, nb);
    return hipGetLastError();
}

hipError_t launch_axpy(real_t* dst, const real_t* src, long long n, real_t alpha, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, s, reinterpret_cast<real4_t*>(dst),
                       reinterpret_cast<const real4_t*>(src), n4, alpha);
    return hipGetLastError();
}

// scratch: 2 + 2 * kNormsMaxBlocks doubles ([0], [1]: a result slot of its own, the rest per-block partial sums); out: where the
// two results go (device memory; may be scratch itself)
hipError_t launch_norms(const real_t* p, long long n, double* scratch, double* out, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(out, 0, 2 * sizeof(double), s);
    const long long n4 = n / 4;
    int grid = stream_grid(n4, 256);
    if (grid > kNormsMaxBlocks) grid = kNormsMaxBlocks;
    hipLaunchKernelGGL(norms_partial_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const real4_t*>(p), n4, scratch + 2);
    hipLaunchKernelGGL(norms_final_kernel, dim3(1), dim3(256), 0, s, scratch + 2, grid, out);
    return hipGetLastError();
}
int norms_scratch_doubles() { return 2 + 2 * kNormsMaxBlocks; }

// one fused sweep of soft_norms_partial_kernel over n values from p; its partial sums go to scratch + 2 + 2 * first_block;
// returns the blocks it used through *blocks (at most max_blocks)
hipError_t launch_soft_norms(real_t* p, long long n, long long split, real_t b_lo, real_t b_hi, bool keep_lo, bool store,
                             double* scratch, int first_block, int max_blocks, int* blocks, hipStream_t s) {
    *blocks = 0;
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    int grid = stream_grid(n4, 256);
    if (grid > max_blocks) grid = max_blocks;
    if (grid < 1 || first_block + grid > kNormsMaxBlocks) return hipErrorInvalidValue;
    double* part = scratch + 2 + 2 * first_block;
    real4_t* p4 = reinterpret_cast<real4_t*>(p);
    if (store) hipLaunchKernelGGL((soft_norms_partial_kernel<true>), dim3(grid), dim3(256), 0, s, p4, n4, split / 4, b_lo, b_hi, keep_lo ? 1 : 0, part);
    else hipLaunchKernelGGL((soft_norms_partial_kernel<false>), dim3(grid), dim3(256), 0, s, p4, n4, split / 4, b_lo, b_hi, keep_lo ? 1 : 0, part);
    *blocks = grid;
    return hipGetLastError();
}
hipError_t launch_norms_final(const double* scratch, int nblocks, double* out, hipStream_t s) {
    hipLaunchKernelGGL(norms_final_kernel, dim3(1), dim3(256), 0, s, scratch + 2, nblocks, out);
    return hipGetLastError();
}
int norms_max_blocks() { return kNormsMaxBlocks; }

hipError_t launch_circshift(const real_t* in, real_t* out, int batch, int Nr, int Nc, int sr, int sc,
                            hipStream_t s) {
    const long long plane = (long long)Nr * Nc;
    hipLaunchKernelGGL(circshift_kernel, dim3((unsigned)cdivll(plane, 256), batch), dim3(256), 0, s, in, out, Nr,
                       Nc, sr, sc);
    return hipGetLastError();
}

hipError_t launch_fill_hash(real_t* x, long long n, uint32_t seed, real_t scale, long long index_offset,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_hash_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, x, n, seed, scale,
                       index_offset);
    return hipGetLastError();
}

hipError_t launch_copy(const real_t* src, real_t* dst, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long n4 = n / 4;
    // 1024 workgroups: the best grid for this copy on MI355X at every footprint measured (profiles/r02t_pbench_b{1,8}.txt:
    // 134 MB 16.4 us against 17.8 with 2048; 1 GB 186.5 us = 5.76 TB/s against 216.3 = 4.96)
    int grid = stream_grid(n4, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const real4_t*>(src),
                       reinterpret_cast<real4_t*>(dst), n4);
    return hipGetLastError();
}

}  // namespace pdwt
