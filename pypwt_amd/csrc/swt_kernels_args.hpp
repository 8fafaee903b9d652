// swt_kernels_args.hpp -- argument block of the direct one-pass SWT kernels.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

struct SwtPassArgs {
    const real_t* in0;  // forward: input ; inverse: approximation-like operand (rlo)
    const real_t* in1;  // inverse: detail-like operand (rhi); forward: unused
    real_t* out0;       // forward: low output ; inverse: output
    real_t* out1;       // forward: high output
    int Nr, Nc, f, along_y;
    int hlen;
    FilterBank fb;
    int images = 1;     // the direct one-output-per-thread kernels only: that many (Nr, Nc) planes back to back behind every pointer
};

}  // namespace pdwt
