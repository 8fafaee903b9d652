// strip_walk.hpp -- host-side geometry of the strip walks (no HIP dependency: the CPU emulation build and its tests include it too)
#pragma once

namespace pdwt {

// Rows of a chain per segment for the strip walks (swt_fwdstream / swt_invstream / swt_colstream kernels): k steps of `ty` rows.  A launch of
// units x segments workgroups runs in rounds of `slots` resident workgroups (256 CUs x the kernel's occupancy), a workgroup takes warm + k
// steps: the k with the fewest rounds x steps; ties go to the longer segment (fewer warm-up rows).  Round 6: the first version cut the
// rows into ceil(target / units) segments and rounded UP to whole steps -- 1040 rows became 17 segments of 64 where 1024 rows are 32 of 32:
// half the workgroups, 1.3x the time per sample (profiles/r06_sizes_cliff.txt).
static inline int strip_walk_seg(int rows, long long units, int ty, int warm, int slots) {
    const int kmax = (rows + ty - 1) / ty;
    int best_k = kmax;
    long long best = -1;
    for (int k = kmax; k >= 1; --k) {
        const long long wgs = units * ((rows + ty * k - 1) / (ty * k));
        const long long cost = ((wgs + slots - 1) / slots) * (warm + k);
        if (best < 0 || cost < best) {
            best = cost;
            best_k = k;
        }
    }
    return best_k * ty;
}

}  // namespace pdwt
