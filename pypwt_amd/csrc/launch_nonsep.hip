// launch_nonsep.hip -- launchers of the non-separable 2D level kernels (gfx950).
#include "launch.hpp"
#include "launch_util.hpp"
#include "nonsep_kernels.hpp"

namespace pdwt {

hipError_t launch_nonsep_fwd(const NonsepArgs& a, int batch, hipStream_t s) {
    const long long total = (long long)a.Nrc * a.Ncc;
    const size_t lds = (size_t)nonsep_lds_floats(a.hlen) * sizeof(real_t);
    hipLaunchKernelGGL((nonsep_fwd_kernel<256>), dim3((unsigned)cdivll(total, 256), batch), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_nonsep_inv(const NonsepArgs& a, int batch, hipStream_t s) {
    const long long total = (long long)a.Nr * a.Nc;
    const size_t lds = (size_t)nonsep_lds_floats(a.hlen) * sizeof(real_t);
    hipLaunchKernelGGL((nonsep_inv_kernel<256>), dim3((unsigned)cdivll(total, 256), batch), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace pdwt
