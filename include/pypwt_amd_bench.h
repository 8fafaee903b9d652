/*
 * pypwt_amd_bench.h -- measurement and test hooks of libpypwt_amd.so.
 *
 * NOT part of the drop-in contract (pypwt_amd.h, which mirrors the reference's class member for member, reference
 * src/pypwt.pyx:8-61, pdwt/src/wt.h:20-76): nothing here has a counterpart in the reference, and a binding of the reference's
 * Python class needs none of it.  bench.py, tools/ and tests/ use these entry points: the deterministic on-device input, the
 * per-launch HIP-event timing, the one-level and flat-copy micro-benchmarks the roofline figures come from, the plan's
 * launch list as text, and the dispatch knobs of the A/B measurements.
 */
#ifndef PYPWT_AMD_BENCH_H
#define PYPWT_AMD_BENCH_H

#include "pypwt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fill the plan image on the device with the deterministic test input
 * x[i] = (lowbias32((i + index_offset) ^ seed) >> 8) * 2^-24 * scale (tests/golden, oracle, bench) */
int pdwt_fill_image_hash(pdwt_handle h, uint32_t seed, pdwt_real scale, long long index_offset);
/* per-launch HIP-event timing: when enabled every kernel launch is bracketed by
 * events on the plan's stream; pdwt_kernel_times returns (synchronising) the
 * number of recorded launches and copies names/milliseconds of the first `cap`. */
int pdwt_enable_kernel_timing(pdwt_handle h, int enable);
int pdwt_kernel_times(pdwt_handle h, float* ms, char (*names)[48], int cap);
int pdwt_reset_kernel_times(pdwt_handle h);
/* which of a step's alternative kernels served each recorded launch, same order and count as pdwt_kernel_times: "tile" (LDS
 * tiles), "wave" (registers + DPP), "ring" (register ring + LDS row halo), "generic" for the level launches of the 2D DWT, ""
 * for steps that have one kernel.  The launch NAMES stay what they were ("dwt2_fwd_level", ...): bench.py labels by them. */
int pdwt_kernel_families(pdwt_handle h, char (*families)[16], int cap);
/* micro-benchmark of ONE level: the launch(es) of level `level` (1 = finest; for fused multi-level
 * 1D launches: the first level of the group) of the forward (inverse = 0) or inverse transform are
 * enqueued `reps` times back to back between two HIP events on the plan's stream; returns the mean
 * milliseconds per repetition.  The data the level reads is whatever the buffers hold (run a
 * forward first); nothing else of the plan's state changes. */
int pdwt_time_level(pdwt_handle h, int level, int inverse, int reps, float* ms_per_launch);

/* NEW: the plan's launch lists as text, one line per direction: "fwd: LEVEL[1] LEVEL[2] PYR2[3-4]" (kind[levels]); what the
 * reference decides with if/else at every call (wt.cu:236-305) is decided once per plan here (plan.cpp: build_schedule).
 * Returns the length written (excluding the terminator) or a negative status. */
int pdwt_schedule_string(pdwt_handle h, char* buf, size_t n);

/* the measured ceiling beside it: a plain 16-B-per-lane grid-stride copy of `elems` values (clamped to pdwt_copy_capacity,
 * rounded down to a multiple of 4) from the plan's image buffer (its coefficient region when elems exceeds the image) into scratch, `reps` launches back to back between two
 * HIP events on the plan's stream; mean milliseconds per launch.  It moves 2 * elems * sizeof(pdwt_real) bytes: a level
 * kernel of the same footprint cannot be expected to run faster than this on the same GPU in the same cache state. */
int pdwt_time_copy(pdwt_handle h, long long elems, int reps, float* ms_per_launch);
long long pdwt_copy_capacity(pdwt_handle h); /* the largest `elems` pdwt_time_copy does not clamp: max(image, coefficient region) */
/* NEW: process-wide dispatch knobs (tests and A/B measurements; no counterpart in the reference, whose
 * kernel choice is fixed at compile time, pdwt/src/wt.cu:236-305).  Returns the previous value, or
 * PDWT_ERR_ARG for an unknown key.  Every key is read ONCE PER PLAN, when the plan is created (pdwt_create*, pdwt_clone
 * copies its source's): a plan keeps the values it was built with, so threads driving plans with different settings -- or a
 * thread that moves a knob -- cannot change another plan's kernel choice in mid-transform.  Keys:
 *   "wave_min_log2"  a 2D DWT level runs on the wave-per-tile kernels when at least 2^value samples
 *                    enter it (default 22, fp64 library 16; 0 = always when eligible; 63 = never)
 *   "lds_max_log2"   a 2D DWT level of at most 2^value samples prefers the LDS tiles to the wave-per-tile kernels
 *                    (default 25 since round 4, 24 before: one cache-resident image and the first doubling of it; at the
 *                    default, forward levels of images below 2^24 samples stay on the tiles up to 2^26; fp64 library 0 = never)
 *   "ring_min_log2"  a 2D DWT level of 12 or 16 taps runs on the register-ring kernels (dwt2_ring_kernels.hpp: one wavefront per
 *                    tile, row halo through LDS, column filter as running sums in registers) when at least 2^value samples
 *                    enter it and its rows have at least 1024 columns (default 25: batches from two 4096^2 images on, 6-15 %
 *                    faster than the LDS tiles there; 63 = never; below 25: every level of 10-20 taps of that size on, any
 *                    width -- tests and measurements)
 *   "long_fwd" / "long_inv"   the shortest (even) filter whose large 2D DWT levels run on the strip-streaming kernels
 *                    (dwt2_long_kernels.hpp: a workgroup walks down a strip of 64 coefficient columns, the row-filtered
 *                    history in LDS, both passes register-blocked -- no halo is filtered twice).  Default 18 taps: 18 taps
 *                    from 2^26 samples per launch; from 20 taps the inverse from 2^24 and the forward from 2^25; the forward
 *                    of 28 taps and more from 2^24, the inverse of 32 taps and more from 2^22; 0 = never;
 *                    100 + n = n taps (10-40) at every size the kernels take (tests).  Part of the plan's snapshot.  The fp64
 *                    library: forward 18-36 taps from 2^24 samples (up to 32 taps from 2^22), inverse 18-26 taps from 2^22.
 *   "reg1d"          bit 0 / bit 1: the forward / inverse 1D DWT levels run three at a time in registers
 *                    (dwt1_reg_kernels.hpp) where the rows qualify (even hlen <= 20, rows of >= 2048 samples that
 *                    are multiples of 32); default 3; 0 = the workgroup-wide LDS pyramids (57.6 vs 69.5 us per
 *                    forward+inverse on 2^24 samples); the forward uses them up to 2^25 samples per plan (a batch of
 *                    long rows is faster through the LDS pyramid), bit 2 lifts that limit.  fp32: only where at least 2^23
 *                    samples in rows of at least 16384 enter the launch -- shorter rows and smaller transforms are faster
 *                    through the one-launch LDS pyramid (4096 rows of 4096, db4 L5: 79.6 vs 70.6 us; 2^20 samples: 16.2 vs
 *                    14.2) --, bit 3 lifts those limits (tests).  
 *   "swt_fused"      1 (default): 2D SWT plans whose 3L+2 planes about fit the Infinity Cache (<= 320 MiB) run several
 *                    levels per launch in registers: 2-tap banks (haar) levels 1-3 and 4-6 (swt2_fused_kernels.hpp: 11 / 8
 *                    instead of 15 / 10 planes of traffic), 4-tap banks (db2, sym2, custom) levels (1, 2) and (3, 4)
 *                    (swt2_fused4_kernels.hpp: 8 instead of 10 planes per pair); 2: at any size; 0: a launch per level.
 *                    
 *   "swt_split_fwd" / "swt_split_inv"   the shortest (even) filter whose 2D SWT levels run as a register-blocked row
 *                    launch + column launch through scratch (swt_split_kernels.hpp) instead of one LDS-tiled launch:
 *                    defaults 18 / 10 taps (where the two launches are faster on MI355X: 40 taps 127-158 -> 46 us per 2048^2
 *                    forward level, 16 taps 93 -> 40 us per inverse level); the inverse of images below 1024^2 / 2048^2 starts at 24 / 12 taps
 *                    (two one-round launches cost more than they save there); 0 = never; 100 + n = n taps at every size (tests).
 *                    Part of the plan's snapshot.
 *   "swt_fwdstream" / "swt_invstream"   (round 6) the shortest (even) filter whose 2D SWT levels run in ONE launch each, row and
 *                    column pass streamed down column strips (swt_fwdstream_kernels.hpp: 6-40 taps, dilations 1-16;
 *                    swt_invstream_kernels.hpp: 6-28 taps, dilations 1-8; rows of whole 16-B groups, from 2^14 samples per
 *                    launch): default 6; 0 = never (the levels then run on the tiles / the two launches above); 100 + n = n
 *                    taps at every size they take (tests).  Part of the plan's snapshot.
 *   "swt_colstream"  (round 6) the shortest (even) filter whose two-launch SWT levels run their COLUMN pass as a strip walk with the
 *                    filter's history in LDS (swt_colstream_kernels.hpp) instead of the register kernels: default 10 (the inverse
 *                    at every size, the forward from 2^23 samples per launch); 0 = never; 100 + n = n taps, both directions, every size.
 *   "dwt_split_fwd" / "dwt_split_inv"   the shortest (even) filter whose DECIMATED 2D levels run as a register-blocked row
 *                    launch + column launch through scratch (dwt2_split_kernels.hpp; rows and columns even, columns a
 *                    multiple of 8) instead of one LDS-tiled launch.  Default 0 (never): measured no faster than LDS tiles
 *                    of the right shape on MI355X (40 taps 2048^2 L5 forward 91 against 53 us), so the kernels are compiled
 *                    into the test library libpypwt_amd_lab.so only and the product accepts the keys and does nothing;
 *                    100 + n = n taps at every size (tests).  Part of the plan's snapshot.
 *   "chain"          levels 1..K of a 2D DWT in ONE launch with in-launch hand-offs between the levels
 *                    (dwt2_chain_kernels.hpp; even filters of at most 8 taps, whole 16 x 128 tiles at every level):
 *                    0 (default): never -- measured break-even for two levels and slower beyond on MI355X; 1: one
 *                    cache-resident image (2^22 < samples <= 2^24) in both directions and the inverse of batches of
 *                    >= 2^26 samples; 2: wherever the kernel applies; 3: 2 and the forward of such batches too.  Read when
 *                    a plan is created.
 *   "chain_timeout"  ticks of the 100 MHz s_memrealtime counter a chained tile waits for a producer tile before it
 *                    computes that producer itself (default 3000 = 30 us; 0 makes nearly every wait take that path: tests)
 *   "wave2"          1: eligible forward level pairs run as ONE two-level wave launch (default 0: measured
 *                    slower than two launches on MI355X, kept for tests and re-measurement) */
int pdwt_set_tuning(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* PYPWT_AMD_BENCH_H */
