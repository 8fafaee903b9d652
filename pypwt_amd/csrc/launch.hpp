// launch.hpp -- host-side launch wrappers around the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_common.hpp"
#include "swt_kernels_args.hpp"

namespace pdwt {

// every even filter length of the built-in table gets its own fully unrolled instantiation
#define PDWT_EVEN_HLENS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) \
    X(28) X(30) X(32) X(34) X(36) X(38) X(40)

// element-wise operator codes of launch_ew
enum EwOp { EW_SOFT = 0, EW_HARD = 1, EW_LINF = 2, EW_SCALE = 3 };

// Dispatch knobs that a level LAUNCH consults (pdwt_set_tuning).  A plan takes a snapshot when it is created and makes it
// the calling thread's active set around its launches, so that two threads driving plans with different settings -- or a
// third thread calling pdwt_set_tuning -- cannot change each other's kernel choice in mid-transform; with no active set
// (direct calls of the launchers: tools, emulation) the process-wide values apply.
struct Tuning {
    int wave_min_log2, lds_max_log2, swt_split_fwd, swt_split_inv, dwt_split_fwd, dwt_split_inv, ring_min_log2, long_fwd, long_inv, swt_colstream, swt_fwdstream, swt_invstream;
    int wave2, swt_fused, chain, reg1d;  // read by build_schedule: a clone rebuilds its launch lists from its source's values
};
Tuning current_tuning();                  // the process-wide values now
void set_active_tuning(const Tuning* t);  // thread-local; nullptr = the process-wide values
const Tuning* active_tuning();
int get_swt_split_min(int inverse);
int get_dwt_split_min(int inverse);
struct ActiveTuning {  // RAII: the plan's snapshot for the duration of a forward / inverse
    const Tuning* prev;
    explicit ActiveTuning(const Tuning* t) : prev(active_tuning()) { set_active_tuning(t); }
    ~ActiveTuning() { set_active_tuning(prev); }
};

// Which kernel FAMILY the last level launch of the calling thread went to ("tile", "wave", "ring", "generic", ...): the launch
// names of a plan (pdwt_kernel_times) say which step ran, not which of the alternative kernels served a LEVEL step; the
// dispatch-coverage test (tests/test_gpu_dispatch.py) reads this through pdwt_kernel_families.  Diagnostics only.
void note_family(const char* family);
const char* last_family();
hipError_t launch_dwt2_fwd(const Fwd2DArgs& a, int batch, hipStream_t s);
hipError_t launch_dwt2_inv(const Inv2DArgs& a, int batch, hipStream_t s);
// tuned kernels; hipErrorNotSupported = preconditions not met, use the generic launcher
hipError_t try_launch_dwt2_fwd_fast(const Fwd2DArgs& a, int batch, hipStream_t s);
hipError_t try_launch_dwt2_inv_fast(const Inv2DArgs& a, int batch, hipStream_t s);
// wave-per-tile kernels (registers + DPP, no LDS): hlen <= 8; seg_hint > 0 forces the rows per wavefront
hipError_t try_launch_dwt2_fwd_wave(const Fwd2DArgs& a, int batch, hipStream_t s, int seg_hint = 0);
hipError_t try_launch_dwt2_inv_wave(const Inv2DArgs& a, int batch, hipStream_t s, int seg_hint = 0);
// register-ring kernels for 10-20 taps (dwt2_ring_kernels.hpp): cpl = image columns per lane (4; 2 in the lab library only); seg_hint > 0
// forces the rows per wavefront
hipError_t try_launch_dwt2_fwd_ring(const Fwd2DArgs& a, int batch, hipStream_t s, int cpl = 4, int seg_hint = 0);
hipError_t try_launch_dwt2_inv_ring(const Inv2DArgs& a, int batch, hipStream_t s, int cpl = 4, int seg_hint = 0);
// strip-streaming kernels for long filters (dwt2_long_kernels.hpp): even hlen 10-40, even sides, rows of whole 16-B groups; seg_hint > 0
// forces the rows per segment
hipError_t try_launch_dwt2_fwd_long(const Fwd2DArgs& a, int batch, hipStream_t s, int seg_hint = 0);
hipError_t try_launch_dwt2_inv_long(const Inv2DArgs& a, int batch, hipStream_t s, int seg_hint = 0);
int set_long_min_taps(int inverse, int taps);  // shortest filter on them (0: never; 100 + n: n taps at every size they take); returns the previous value
int get_long_min_taps(int inverse);
int set_swt_colstream_min(int taps);  // column pass of the two-launch SWT levels streamed through an LDS history (swt_colstream_kernels.hpp) from `taps` taps; 0: never
int get_swt_colstream_min();
int set_swt_fwdstream_min(int taps);  // forward SWT levels in one launch (swt_fwdstream_kernels.hpp) from `taps` taps; 0: never; 100 + n: n taps at every size
int get_swt_fwdstream_min();
int set_swt_invstream_min(int taps);  // ... inverse SWT levels in one launch (swt_invstream_kernels.hpp)
int get_swt_invstream_min();
int set_ring_min_log2(int value);  // 2D DWT levels of at least 2^value samples with 12-20 taps run on them (63 = never; below the default: 10-20 taps, tests)
int get_ring_min_log2();
int set_wave_min_log2(int value);  // returns the previous threshold
int get_wave_min_log2();
int set_lds_max_log2(int value);   // 2D DWT levels of at most 2^value samples prefer the LDS tiles to the wave kernels (0 = never)
int get_lds_max_log2();
int set_wave2_enabled(int value);  // two-levels-per-wavefront forward (opt-in); returns the previous setting
int get_wave2_enabled();
int set_swt_fused_enabled(int value);  // 2-tap 2D SWT levels 1-3 / 4-6 in one launch each (swt2_fused_kernels.hpp); read when a plan is built
int get_swt_fused_enabled();
int set_reg1d_enabled(int value);   // 1D levels three at a time in registers (dwt1_reg_kernels.hpp); bit 0 forward, bit 1 inverse; read when a plan is built
int get_reg1d_enabled();
// two forward levels per wavefront (A_l stays in registers); contract of launch_dwt2_fwd_pyr2
bool dwt2_wave2_supported(int hlen, int N0r, int N0c);
hipError_t launch_dwt2_fwd_wave2(const real_t* in, real_t* const det1[3], real_t* const band2[4], int N0r, int N0c,
                                 int hlen, const FilterBank& fb, int batch, hipStream_t s, int seg_hint = 0);
// two consecutive 2D levels in one launch (small levels only, see launch_dwt2_pyramid.hip)
bool dwt2_pyramid_supported(int hlen, int N0r, int N0c, bool inverse);   // tile pyramid: even filters of at most 16 taps
bool dwt2_strip_supported(int hlen, int N0r, int N0c);     // streaming strips: at most 8 taps
hipError_t launch_dwt2_fwd_pyr2(const real_t* in, real_t* const det1[3], real_t* const band2[4], int N0r, int N0c,
                                int hlen, const FilterBank& fb, int batch, hipStream_t s);
// same contract, streaming-strip kernel for large inputs (forward only)
hipError_t launch_dwt2_fwd_strip2(const real_t* in, real_t* const det1[3], real_t* const band2[4], int N0r, int N0c,
                                  int hlen, const FilterBank& fb, int batch, hipStream_t s);
hipError_t launch_dwt2_inv_strip2(const real_t* const band2[4], const real_t* const det1[3], real_t* out, int N0r,
                                  int N0c, int hlen, const FilterBank& fb, int batch, hipStream_t s);
hipError_t launch_dwt2_inv_pyr2(const real_t* const band2[4], const real_t* const det1[3], real_t* out, int N0r, int N0c,
                                int hlen, const FilterBank& fb, int batch, hipStream_t s);
// K consecutive 2D levels in ONE launch with in-launch hand-offs between levels (dwt2_chain_kernels.hpp): (Nr, Nc) enter the
// finest level of the group; det[3 k + b] = band b (H, V, D) of the group's k-th level (finest first), app[k] its approximation
// plane (forward: outputs; inverse: app[K-1] is the input, app[K-2..0] are intermediates); flags: batch x dwt2_chain_tiles()
// zero-initialised words owned by the plan; epoch: a counter that grows with every launch on those flags
bool dwt2_chain_supported(int hlen, int Nr, int Nc, int K);
int dwt2_chain_tiles(int Nr, int Nc, int K);
hipError_t launch_dwt2_fwd_chain(const real_t* in, real_t* const* det, real_t* const* app, int Nr, int Nc, int K, int hlen,
                                 const FilterBank& fb, int batch, unsigned* flags, unsigned epoch, hipStream_t s);
hipError_t launch_dwt2_inv_chain(real_t* out, real_t* const* det, real_t* const* app, int Nr, int Nc, int K, int hlen,
                                 const FilterBank& fb, int batch, unsigned* flags, unsigned epoch, hipStream_t s);
int set_chain_enabled(int value);  // 0 never (the default: opt-in, and the product build stubs the chain kernels out), 1 one cache-resident image + batch inverses, 2 wherever supported (tests), 3 = 2 + batch forwards too
int get_chain_enabled();
int set_chain_timeout(int ticks);  // s_memrealtime ticks (100 MHz) a chained tile waits for a producer before computing it itself
// three consecutive 2D levels in one launch, small images (launch_dwt2_pyr3.hip): det[3 k + b] = band b (H, V, D) of the
// k-th level of the group (finest first); rows and columns multiples of 8, even filters of at most 16 taps (fp64: 8)
bool dwt2_pyr3_supported(int hlen, int N0r, int N0c);
hipError_t launch_dwt2_fwd_pyr3(const real_t* in, real_t* const det[9], real_t* out, int N0r, int N0c, int hlen,
                                const FilterBank& fb, int batch, hipStream_t s);
hipError_t launch_dwt2_inv_pyr3(const real_t* app, real_t* const det[9], real_t* out, int N0r, int N0c, int hlen,
                                const FilterBank& fb, int batch, hipStream_t s);
// ALL remaining levels of small approximations in one launch, one workgroup per image (launch_dwt2_tail.hip): (R0, C0) enter
// the group's finest level (at most 16384 samples, fp64 8192; even sizes at every level's input: powers of two or not), det[3 k + b] = band b of the
// group's k-th level (finest first); forward: in = A_{l-1}, out = A_L; inverse: in = A_L, out = A_{l-1}
bool dwt2_tail_supported(int hlen, int R0, int C0, int K);
int dwt2_tail_max_levels(int hlen, int R0, int C0, int Kmax);  // the most levels (<= Kmax) one launch can take from (R0, C0) on
hipError_t launch_dwt2_tail(const real_t* in, real_t* const* det, real_t* out, int R0, int C0, int K, int hlen, bool inverse,
                            const FilterBank& fb, int batch, hipStream_t s);
// the WHOLE 2D SWT of tiny images (at most 4096 samples, any sizes), one workgroup per image (launch_swt_tail.hip):
// det[3 (l - 1) + b] = band b of level l; forward: in = images, out = A_L; inverse: in = A_L, out = images, beta[l - 1] = the soft
// threshold applied to level l's details as they are read (nullptr: none)
bool swt2_tail_supported(int hlen, int Nr, int Nc, int L);
hipError_t launch_swt2_tail(const real_t* in, real_t* const* det, real_t* out, int Nr, int Nc, int L, int hlen, bool inverse,
                            const FilterBank& fb, const real_t* beta, int batch, hipStream_t s);
hipError_t launch_dwt1_fwd(const Fwd1DArgs& a, hipStream_t s);
hipError_t launch_dwt1_inv(const Inv1DArgs& a, hipStream_t s);
// K consecutive 1D levels in one launch (2^K must divide N0, even hlen); hipErrorNotSupported otherwise
int dwt1_fused_max_levels(int hlen);
bool dwt1_fused_supported(int hlen, int N0, int K, bool strict);  // strict: rows of 2^(K+2) samples (the several-rows-per-wavefront kernels), else 2^(K+1)
// batches of short rows: several rows per one-wavefront workgroup, any number of levels from one on (dwt1_rows_kernels.hpp)
bool dwt1_rows_tail_applies(int rows, int N0, int K, int hlen);
// up to three levels per launch in registers (dwt1_reg_kernels.hpp): even hlen <= 20, rows of >= 2048 samples
bool dwt1_reg_supported(int hlen, int N0, int K);
hipError_t launch_dwt1_fwd_reg(const real_t* in, real_t* const* det, real_t* app, int rows, int N0, int K, int hlen,
                               const FilterBank& fb, hipStream_t s);
hipError_t launch_dwt1_inv_reg(const real_t* app, const real_t* const* det, real_t* out, int rows, int N0, int K, int hlen,
                               const FilterBank& fb, hipStream_t s);
hipError_t launch_dwt1_fwd_fused(const real_t* in, real_t* const* det, real_t* app, int rows, int N0, int K, int hlen,
                                 const FilterBank& fb, hipStream_t s);
hipError_t launch_dwt1_inv_fused(const real_t* app, const real_t* const* det, real_t* out, int rows, int N0, int K,
                                 int hlen, const FilterBank& fb, hipStream_t s);
// fused a-trous level; the host guarantees a.f divides a.Nr
hipError_t launch_swt2_fwd(const Swt2DArgs& a, int batch, hipStream_t s);
hipError_t launch_swt2_inv(const Swt2DArgs& a, int batch, hipStream_t s);
bool swt2_fwd_stream_takes(const Swt2DArgs& a, int batch);  // the one-launch forward level of swt_fwdstream_kernels.hpp would take this level
hipError_t try_launch_swt2_fwd_stream(const Swt2DArgs& a, int batch, hipStream_t s);  // hipErrorNotSupported: declined
bool swt2_inv_stream_takes(const Swt2DArgs& a, int batch);
hipError_t try_launch_swt2_inv_stream(const Swt2DArgs& a, int batch, hipStream_t s);
// one a-trous level as a row launch + a column launch through scratch (2 Nr Nc batch elements): swt_split_kernels.hpp
bool swt2_split_supported(int hlen, int Nr, int Nc, int f, bool inverse, long long samples_per_launch);
int set_swt_split_min(int inverse, int taps);  // shortest filter on the split path (0: never); returns the previous value
hipError_t launch_swt2_split(const Swt2DArgs& a, real_t* scratch, bool inverse, int batch, hipStream_t s);
// one DECIMATED 2D level as a row launch + a column launch through scratch (Nr Nc batch elements): dwt2_split_kernels.hpp;
// (Nr, Nc) = the level's image side (forward: its input, inverse: its output), both even, Nc a multiple of 8
bool dwt2_split_supported(int hlen, int Nr, int Nc, bool inverse, long long samples_per_launch);
int set_dwt_split_min(int inverse, int taps);  // shortest filter on the split path (0: never; 100 + n: n taps at every size); returns the previous value
hipError_t launch_dwt2_split_fwd(const Fwd2DArgs& a, real_t* scratch, int batch, hipStream_t s);
hipError_t launch_dwt2_split_inv(const Inv2DArgs& a, real_t* scratch, int batch, hipStream_t s);
// levels l0 .. l0+K-1 (K = 2, 3; l0 = 1 or 4) of a 2-tap 2D SWT in one launch (swt2_fused_kernels.hpp)
bool swt2_fused_supported(int hlen, int Nr, int Nc, int l0, int K, bool inverse = false);
hipError_t launch_swt2_fused(const real_t* in, real_t* out, real_t* const* det, int Nr, int Nc, int l0, int K, bool inverse,
                             int hlen, const FilterBank& fb, const real_t* beta, int batch, hipStream_t s);
hipError_t launch_swt_pass_fwd(const SwtPassArgs& a, hipStream_t s);
hipError_t launch_swt_pass_inv(const SwtPassArgs& a, hipStream_t s);

struct NonsepArgs;
hipError_t launch_nonsep_fwd(const NonsepArgs& a, int batch, hipStream_t s);
hipError_t launch_nonsep_inv(const NonsepArgs& a, int batch, hipStream_t s);

// streaming operators over a 16-B aligned range of n floats (n % 4 == 0)
hipError_t launch_ew(int op, real_t* p, long long n, real_t b, hipStream_t s);
hipError_t launch_group_soft(real_t* d0, real_t* d1, real_t* d2, real_t* ap, long long n, real_t beta, int nb,
                             hipStream_t s);
hipError_t launch_axpy(real_t* dst, const real_t* src, long long n, real_t alpha, hipStream_t s);
hipError_t launch_norms(const real_t* p, long long n, double* scratch, double* out, hipStream_t s);  // scratch: norms_scratch_doubles() doubles; out: 2 doubles (device)
int norms_scratch_doubles();
// soft threshold + norms of the result in one sweep (ops_kernels.hpp); several sweeps share the scratch, then one final launch
hipError_t launch_soft_norms(real_t* p, long long n, long long split, real_t b_lo, real_t b_hi, bool keep_lo, bool store,
                             double* scratch, int first_block, int max_blocks, int* blocks, hipStream_t s);
hipError_t launch_norms_final(const double* scratch, int nblocks, double* out, hipStream_t s);
int norms_max_blocks();
hipError_t launch_circshift(const real_t* in, real_t* out, int batch, int Nr, int Nc, int sr, int sc, hipStream_t s);
hipError_t launch_copy(const real_t* src, real_t* dst, long long n, hipStream_t s);  // 16-B grid-stride copy (n % 4 == 0)
hipError_t launch_fill_hash(real_t* x, long long n, uint32_t seed, real_t scale, long long index_offset,
                            hipStream_t s);

}  // namespace pdwt
