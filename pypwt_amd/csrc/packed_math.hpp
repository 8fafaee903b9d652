// packed_math.hpp -- 2-wide helpers shared by the tuned kernels.
//
// gfx950 executes v_pk_fma_f32 on (x,y) register pairs; keeping the low/high filter outputs (or the
// (A,V)/(H,D) band pairs) interleaved makes every multiply-add of the transform one packed
// instruction with the (lo,hi) tap pair coming from SGPRs.  The element type is real_t: the fp64 build
// (-DPDWT_DOUBLE) compiles the register kernels of dwt2_wave_kernels.hpp with the same code, where a "pair"
// is two v_fma_f64 (there is no packed fp64 FMA) and a 4-vector two 16-B memory instructions.
#pragma once

#include "kernels_common.hpp"

namespace pdwt {

#ifdef PDWT_CPU_EMU
struct v2f {
    real_t x, y;
};
static inline v2f mk2(real_t a, real_t b) { return v2f{a, b}; }
static inline v2f fma2(v2f a, v2f b, v2f c) { return v2f{a.x * b.x + c.x, a.y * b.y + c.y}; }
#else
typedef real_t v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2f mk2(real_t a, real_t b) {
    v2f r;
    r.x = a;
    r.y = b;
    return r;
}
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif
static PDWT_DEVICE v2f bc(real_t a) { return mk2(a, a); }

// Packed multiply-adds with ONE HALF of an operand pair broadcast, spelled out: hipcc broadcasts the low half of a
// register pair through op_sel_hi but copies the high half into a fresh pair first (two v_mov_b32 per use, and the copies
// stay live: the 16-tap split SWT column kernel went to 294 VGPRs with 409 moves beside its 512 packed FMAs).
//   fma2_bx / fma2_by (p, t, acc):  acc + (p.x, p.x) * t   /   acc + (p.y, p.y) * t      t: a (uniform) tap pair in SGPRs
//   fma2_tx / fma2_ty (p, t, acc):  acc + p * (t.x, t.x)   /   acc + p * (t.y, t.y)
#if !defined(PDWT_CPU_EMU) && !defined(PDWT_DOUBLE)
static __device__ __forceinline__ v2f fma2_bx(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(p), "s"(t));
    return acc;
}
static __device__ __forceinline__ v2f fma2_by(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "s"(t));
    return acc;
}
static __device__ __forceinline__ v2f fma2_tx(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(p), "s"(t));
    return acc;
}
static __device__ __forceinline__ v2f fma2_ty(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "s"(t));
    return acc;
}
// the same with the tap pair in VECTOR registers (dwt2_long_kernels.hpp: 84 tap SGPRs beside the kernel's pointers and sizes
// spill; a wavefront that may use 256 VGPRs keeps the table there instead)
static __device__ __forceinline__ v2f fma2_bx_v(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(p), "v"(t));
    return acc;
}
static __device__ __forceinline__ v2f fma2_by_v(v2f p, v2f t, v2f acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "v"(t));
    return acc;
}
static __device__ __forceinline__ v2f in_vgprs(v2f t) {  // a uniform value the compiler must hold in vector registers from here on
    asm volatile("" : "+v"(t));
    return t;
}
static __device__ __forceinline__ v2f fma2_s(v2f p, v2f t, v2f acc) {  // element-wise, the tap pair from SGPRs
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "s"(t));
    return acc;
}
#else
static PDWT_DEVICE v2f fma2_bx(v2f p, v2f t, v2f acc) { return fma2(bc(p.x), t, acc); }
static PDWT_DEVICE v2f fma2_by(v2f p, v2f t, v2f acc) { return fma2(bc(p.y), t, acc); }
static PDWT_DEVICE v2f fma2_tx(v2f p, v2f t, v2f acc) { return fma2(p, bc(t.x), acc); }
static PDWT_DEVICE v2f fma2_ty(v2f p, v2f t, v2f acc) { return fma2(p, bc(t.y), acc); }
static PDWT_DEVICE v2f fma2_s(v2f p, v2f t, v2f acc) { return fma2(p, t, acc); }
static PDWT_DEVICE v2f fma2_bx_v(v2f p, v2f t, v2f acc) { return fma2(bc(p.x), t, acc); }
static PDWT_DEVICE v2f fma2_by_v(v2f p, v2f t, v2f acc) { return fma2(bc(p.y), t, acc); }
#ifdef PDWT_CPU_EMU
static PDWT_DEVICE v2f in_vgprs(v2f t) { return t; }
#else
static __device__ __forceinline__ v2f in_vgprs(v2f t) {  // fp64: a pair is four vector registers
    asm volatile("" : "+v"(t));
    return t;
}
#endif
#endif

// Reads 16 B from LDS as ONE ds_read_b128 (256 B/clk) even when only some components are used
// afterwards; without the barrier hipcc narrows it to ds_read2_b32 / ds_read2_b64 pairs, which
// run at half the LDS rate (MI355X_MICROARCH.md, LDS table).
#ifdef PDWT_CPU_EMU
typedef f32x4 v4f;
static inline v4f lds_load16(const void* p) { return *reinterpret_cast<const v4f*>(p); }
static inline void lds_pin(v4f&) {}
#else
typedef real_t v4f __attribute__((ext_vector_type(4)));
// An empty asm barrier right after the load would also make the wave WAIT for it before issuing the next one.
// Where several loads feed one computation, issue them all with lds_load16 and pin them afterwards
// (LDS returns in order, so the compiler waits with a counting s_waitcnt and the loads pipeline).
static __device__ __forceinline__ v4f lds_load16(const void* p) { return *reinterpret_cast<const v4f*>(p); }
static __device__ __forceinline__ void lds_pin(v4f& w) { asm volatile("" : "+v"(w)); }
#endif

// taps interleaved as (lo[j], hi[j]) pairs
struct FilterBankI {
    v2f t[kMaxTaps];
};

}  // namespace pdwt
