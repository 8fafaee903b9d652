// swt_kernels_args.hpp -- argument block of the direct one-pass SWT kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

struct SwtPassArgs {
    const float* in0;  // forward: input ; inverse: approximation-like operand (rlo)
    const float* in1;  // inverse: detail-like operand (rhi); forward: unused
    float* out0;       // forward: low output ; inverse: output
    float* out1;       // forward: high output
    int Nr, Nc, f, along_y;
    int hlen;
    FilterBank fb;
};

}  // namespace pdwt
