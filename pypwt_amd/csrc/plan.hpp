// plan.hpp -- host-side transform plan (the MI355X counterpart of the reference's
// C++ class Wavelets, pdwt/src/wt.h:20-76).
//
// A plan owns ONE device arena:
//
//   [ band 0 = A_L | band 1 | band 2 | ...  ]  coefficient region, reference order
//   [ A_1 | A_2 | ... | A_{L-1} ]              intermediate approximations (DWT)
//     or [ ping | pong ]                        two full-size planes (SWT, L >= 2)
//   [ image ]
//
// Every band is [batch][rows][cols] float32, starts on a 256-B boundary and is
// zero-padded to a multiple of 64 floats, so element-wise operators sweep whole
// regions with 16-B accesses.  The forward transform writes A_l to its own slot
// and the last level writes band 0 directly: there is no ping-pong fix-up copy
// (reference: pdwt/src/separable.cu:234,389,535,667, haar.cu:83,112,186,214) and
// the inverse does not overwrite band 0 (reference: wt.cu:272-275).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/pypwt_amd.h"
#include "../../include/pypwt_amd_bench.h"
#include "kernels_common.hpp"
#include "launch.hpp"

namespace pdwt {

struct Band {
    long long off;   // offset in floats from the arena base
    int rows, cols;  // per image
    long long elems(int batch) const { return (long long)batch * rows * cols; }
};

// one launch of a plan's forward / inverse launch list (plan.cpp: build_schedule)
struct Step {
    enum Kind { LEVEL = 0, PYR2 = 1, STRIP2 = 2, FUSED1D = 3, WAVE2 = 4, REG1D = 5, SWTF = 6, PYR3 = 7, CHAIN = 8, TAIL = 9 };
    int kind;
    int level;  // first (finest) level the launch works on
    int K;      // number of levels it covers
};

struct KernelStamp {
    hipEvent_t start, stop;
    char name[48];
    char family[16];  // which of a step's alternative kernels ran (launch.hpp: note_family); "" when the step has only one
};

}  // namespace pdwt

struct pdwt_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    int batch = 1;
    pdwt_info info{};  // ndims, Nr, Nc, nlevels, do_swt, hlen   (w_info, utils.h:9-19)
    int do_separable = 1;
    int do_cycle_spinning = 0;
    int state = PDWT_INIT;
    char wname[128] = {0};
    int shift_r = 0, shift_c = 0;  // current cycle-spinning shift (wt.h:27-28)

    real_t* arena = nullptr;
    long long arena_elems = 0;
    size_t arena_bytes = 0;  // size of the device block behind `arena` (it may come from the pool, up to 1.25x larger)
    std::vector<pdwt::Band> bands;      // coefficient bands, index = `num`
    long long coeff_elems = 0;          // size of the (padded) coefficient region
    std::vector<long long> approx_off;  // DWT: [l] for 1 <= l < L ; SWT: [0], [1] ping-pong
    long long image_off = 0;
    std::vector<int> lr, lc;  // per-level image dims, lr[0] = Nr ... lr[L]

    real_t* tmp = nullptr;  // lazily allocated scratch (circshift, SWT fallback)
    long long tmp_elems = 0;
    double* d_red = nullptr;  // two fp64 accumulators for the norms
    double* h_red = nullptr;  // pinned host landing zone of the two results (a pageable destination makes the copy ~20 us slower)
    size_t d_red_bytes = 0;   // size of the block behind them (a pool block may be larger than the 256 B asked for)

    pdwt::FilterBank dec{}, rec{};
    real_t* d_f2d = nullptr;  // non-separable banks: fwd LL,LH,HL,HH then inv, each hlen*hlen
    bool f2d_custom = false;

    // A soft_threshold that has been requested but not yet applied: the fused SWT inverse applies
    // it while it loads the detail bands (saves one read+write sweep of 3L full-size planes); any
    // other consumer of the coefficients materialises it first (plan.cpp: materialize_pending).
    bool pend_soft = false;
    real_t pend_beta = 0.f;
    int pend_normalize = 0;
    // ... and one that the fused inverse has applied on the fly without writing the thresholded details
    // back (plan.cpp: materialize_consumed)
    bool soft_consumed = false;
    real_t consumed_beta = 0.f;
    int consumed_normalize = 0;

    std::vector<pdwt::Step> sched_fwd, sched_inv;  // launch lists, in execution order
    pdwt::Tuning tune{};  // the dispatch knobs as they stood when the plan was created (launch.hpp: ActiveTuning)
    // Step::CHAIN (several levels in one launch, dwt2_chain_kernels.hpp): per-tile hand-off flags of the forward and of the
    // inverse chain, batch x chain_tiles words each, zeroed once; every launch stamps them with a new epoch
    unsigned* chain_flags = nullptr;
    long long chain_words = 0;  // per direction
    unsigned chain_epoch = 0;

    bool timing = false;
    std::vector<pdwt::KernelStamp> stamps;

    real_t* image_ext = nullptr;  // pdwt_bind_image: the image lives in memory the caller owns (another plan's band, say)
    real_t* image() const { return image_ext ? image_ext : arena + image_off; }
    real_t* band(int num) const { return arena + bands[num].off; }
};
