// kernels_common.hpp -- shared device helpers for the gfx950 wavelet kernels.
//
// Every kernel body is written as a "tile function"
//     template<...> PDWT_DEVICE void xxx_tile(const Args&, int bx, int by, int bz, float* smem)
// called from a thin __global__ wrapper with blockIdx and the dynamic-LDS base.
// Work inside a tile is organised in barrier-separated phases; inside a phase a
// thread only touches its own registers and LDS/global memory.  That discipline
// lets tests/cpu_emu compile the SAME tile functions with g++ (PDWT_CPU_EMU: a
// phase becomes a loop over thread ids, a barrier becomes nothing) and fuzz the
// index math against the oracle in this GPU-less container; it is a sanitizer
// build only and is never part of the shipped library.
#pragma once

#include <stddef.h>
#include <stdint.h>

// Arithmetic / storage type of the library.  The default build is fp32 (the reference's default and
// the only type its Python class accepts); -DPDWT_DOUBLE builds the fp64 variant (the reference's
// DOUBLEPRECISION compile-time switch, pdwt/src/filters.h:16-30): same sources, the generic kernels
// only -- the packed-fp32 fast paths are compiled out.
#ifdef PDWT_DOUBLE
typedef double real_t;
#else
typedef float real_t;
#endif

#ifdef PDWT_CPU_EMU
#include <math.h>
#define PDWT_DEVICE inline
#define PDWT_FOR_THREADS(tid, NT) for (int tid = 0; tid < (NT); ++tid)
#define PDWT_FOR_SUBTHREADS(tid, NT) for (int tid = 0; tid < (NT); ++tid)
#define PDWT_SYNC() ((void)0)
// registers that live across phases: one copy per emulated thread
#define PDWT_PER_THREAD(type, name, count, NT) type name##_store[(NT)][(count)]
#define PDWT_MINE(name, tid) name##_store[tid]
#define PDWT_RESTRICT
struct pdwt_float2 { float x, y; };
struct pdwt_float4 { float x, y, z, w; };
typedef pdwt_float2 f32x2;
typedef pdwt_float4 f32x4;
static inline float pdwt_fma(float a, float b, float c) { return a * b + c; }
static inline double pdwt_fma(double a, double b, double c) { return a * b + c; }
#ifdef PDWT_DOUBLE
struct pdwt_real2 { real_t x, y; };
struct pdwt_real4 { real_t x, y, z, w; };
typedef pdwt_real2 real2_t;
typedef pdwt_real4 real4_t;
#else
typedef pdwt_float2 real2_t;  // the same structs as f32x2 / f32x4: the fp32 tile functions mix both spellings
typedef pdwt_float4 real4_t;
#endif
#else
#include <hip/hip_runtime.h>
#define PDWT_DEVICE __device__ __forceinline__
// one trip: the executing thread
#define PDWT_FOR_THREADS(tid, NT) for (int tid = threadIdx.x, pdwt_once_ = 1; pdwt_once_; pdwt_once_ = 0)
// a tile function that may run as one of SEVERAL sub-groups of NT threads inside a larger workgroup (dwt1_fused_kernels.hpp: short
// rows, four rows per workgroup): the thread index inside the sub-group.  With blockDim.x == NT it is PDWT_FOR_THREADS.
#define PDWT_FOR_SUBTHREADS(tid, NT) for (int tid = (int)(threadIdx.x % (NT)), pdwt_once_ = 1; pdwt_once_; pdwt_once_ = 0)
#define PDWT_SYNC() __syncthreads()
#define PDWT_PER_THREAD(type, name, count, NT) type name##_store[(count)]
#define PDWT_MINE(name, tid) name##_store
#define PDWT_RESTRICT __restrict__
typedef float2 f32x2;
typedef float4 f32x4;
static __device__ __forceinline__ float pdwt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static __device__ __forceinline__ double pdwt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#ifdef PDWT_DOUBLE
typedef double2 real2_t;
typedef double4 real4_t;
#else
typedef float2 real2_t;
typedef float4 real4_t;
#endif
#endif

namespace pdwt {

constexpr int kMaxTaps = 40;  // same limit as the reference (pdwt/src/common.h:15)

// Per-launch filter taps, passed BY VALUE in the kernel-argument segment: with a
// compile-time filter length every tap index is a constant after unrolling and
// the taps are fetched with scalar loads (s_load) straight into SGPRs -- no
// constant-memory bank shared by all plans (the reference's global __constant__
// arrays, pdwt/src/common.h:28-36, make two live plans overwrite each other).
struct FilterBank {
    real_t lo[kMaxTaps];
    real_t hi[kMaxTaps];
};

// ---- index helpers -------------------------------------------------------

PDWT_DEVICE int true_mod(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}

// periodic index into [0, n): one conditional add/sub in the common case
PDWT_DEVICE int wrap_periodic(int i, int n) {
    int m = i;
    if (m < 0) m += n;
    else if (m >= n) m -= n;
    if ((unsigned)m >= (unsigned)n) m = true_mod(i, n);  // tile larger than the signal
    return m;
}

// analysis source index: the signal is extended to n + (n odd) samples by
// repeating the last one, and that is periodized
// (semantics of pdwt/src/separable.cu:114-121, restated as modulo + clamp)
PDWT_DEVICE int wrap_analysis(int i, int n) {
    const int np = n + (n & 1);
    int m = wrap_periodic(i, np);
    return m >= n ? n - 1 : m;
}

PDWT_DEVICE int analysis_centre(int hlen) { return (hlen & 1) ? hlen / 2 : hlen / 2 - 1; }

// soft threshold sign(x) max(|x|-b, 0) written as x - clamp(x, -b, b): the same value for every
// finite x (b >= 0), the identity for b == 0.  fp32 on the GPU: v_med3_f32 + v_sub_f32 (the two-comparison
// form compiled to two v_cmp + two v_cndmask + v_sub: 30 % of the fused SWT inverse's vector instructions).
PDWT_DEVICE real_t soft_shrink(real_t x, real_t b) {
#if !defined(PDWT_CPU_EMU) && !defined(PDWT_DOUBLE)
    return x - __builtin_amdgcn_fmed3f(x, -b, b);
#else
    const real_t c = x < -b ? -b : (x > b ? b : x);
    return x - c;
#endif
}

// ---- agent-coherent 16-B accesses ----------------------------------------
// Data handed from one workgroup to another INSIDE a launch (dwt2_chain_kernels.hpp) is written with `sc1`
// (write-through) stores and read with `sc1` loads, which bypass the CU's L1 and do not rely on another XCD's
// L2 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", valid forms).
// Raw-buffer intrinsics carry the cache-policy bits (aux bit 4 = sc1) and are tracked by the compiler's
// s_waitcnt insertion, unlike inline asm.  The emulation build has one memory: plain accesses.
#if !defined(PDWT_CPU_EMU) && !defined(PDWT_DOUBLE)
typedef unsigned pdwt_u32x4 __attribute__((ext_vector_type(4)));
struct CohPlane {
    __amdgpu_buffer_rsrc_t rsrc;
};
PDWT_DEVICE CohPlane coh_plane(const void* base) {
    CohPlane c;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)0xffffffffu, 0x00020000);
    return c;
}
PDWT_DEVICE f32x4 coh_load16(const CohPlane& c, long long elem) {
    // whole-vector bit cast: with the four components extracted one by one hipcc 7.2 narrows the load to
    // buffer_load_dword and leaves three of the four registers undefined
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(elem * 4), 0, 16));
}
PDWT_DEVICE void coh_store16(const CohPlane& c, long long elem, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pdwt_u32x4, v), c.rsrc, (int)(elem * 4), 0, 16);
}
#else
#ifdef PDWT_DOUBLE
typedef real4_t coh4_t;
#else
typedef f32x4 coh4_t;
#endif
struct CohPlane {
    real_t* base;
};
PDWT_DEVICE CohPlane coh_plane(const void* base) { return CohPlane{const_cast<real_t*>(static_cast<const real_t*>(base))}; }
PDWT_DEVICE coh4_t coh_load16(const CohPlane& c, long long elem) { return *reinterpret_cast<const coh4_t*>(c.base + elem); }
PDWT_DEVICE void coh_store16(const CohPlane& c, long long elem, const coh4_t& v) { *reinterpret_cast<coh4_t*>(c.base + elem) = v; }
#endif

// ---- argument blocks -----------------------------------------------------

// one decimated 2D analysis level: in (Nr,Nc) -> A,H,V,D (Nr2,Nc2)
struct Fwd2DArgs {
    const real_t* in;
    real_t *A, *H, *V, *D;
    int Nr, Nc, Nr2, Nc2;
    long long in_bstride, out_bstride;  // elements between consecutive images of a batch
    int hlen;
    FilterBank fb;  // dec_lo, dec_hi
};

// one decimated 2D synthesis level: A,H,V,D (Nrc,Ncc) -> out (Nr,Nc), Nr <= 2 Nrc
struct Inv2DArgs {
    const real_t *A, *H, *V, *D;
    real_t* out;
    int Nrc, Ncc, Nr, Nc;
    long long in_bstride, out_bstride;
    int hlen;
    FilterBank fb;  // rec_lo, rec_hi
};

// one decimated 1D analysis level on `rows` independent rows: in (rows,Nc) -> L,H (rows,Nc2)
struct Fwd1DArgs {
    const real_t* in;
    real_t *L, *H;
    int rows, Nc, Nc2;
    int hlen;
    FilterBank fb;
};

struct Inv1DArgs {
    const real_t *L, *H;
    real_t* out;
    int rows, Ncc, Nc;
    int hlen;
    FilterBank fb;
};

// one undecimated (a-trous) 2D level, dilation f = 2^(level-1)
struct Swt2DArgs {
    const real_t* in;          // forward: input plane; inverse: unused
    real_t *A, *H, *V, *D;     // forward: outputs; inverse: inputs
    real_t* out;               // inverse: output plane
    int Nr, Nc, f;
    long long bstride;
    int hlen;
    real_t soft_beta;          // inverse: soft-threshold applied to H,V,D as they are loaded (0 = none)
    FilterBank fb;
};

struct Swt1DArgs {
    const real_t* in;   // forward input / inverse approximation
    const real_t* det;  // inverse: detail band
    real_t *L, *H;      // forward outputs
    real_t* out;        // inverse output
    int rows, Nc, f;
    int hlen;
    FilterBank fb;
};

}  // namespace pdwt
