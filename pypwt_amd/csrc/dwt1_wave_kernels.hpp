// dwt1_wave_kernels.hpp -- ALL levels of a (batched) 1D DWT in one launch, one WAVEFRONT per segment
// (gfx950): the 1D counterpart of dwt2_wave_kernels.hpp.
//
// The workgroup-wide fused kernels of dwt1_fused_kernels.hpp run the pyramid level by level between
// barriers: at level k only 2^-k of the threads have work, every level is a barrier-separated phase, and
// rocprofv3 shows them issue-bound (about 42 lane-instructions per input sample, as many scalar as vector
// instructions; profiles/r01h_rocprofv3_summary_cfg3.txt) at 0.46-0.51 of the HBM peak.  Here ONE wavefront
// owns a contiguous piece of the row and runs the whole cascade by itself, a "pass" at a time:
//
//   pass (k, c): the 64 lanes produce the 128 outputs [128 c, 128 c + 128) of level k -- two per lane, low- and
//   high-pass together as packed (L,H) pairs -- from the 256 + hlen - 2 approximation samples of level k-1
//   around [256 c, 256 c + 256), read with 16-B LDS loads from that level's RING (512 floats, private to the
//   wavefront); the details go to HBM (8-B stores, 512 B per wavefront), the approximations into the next
//   level's ring (level K: to HBM).  Level 0's ring is filled from global memory, one 16-B load per lane and
//   chunk, two chunks ahead.
//
// A tiny wave-uniform scheduler runs the deepest pass whose inputs are complete (level k pass c needs level k-1
// up to pass 2c+2), otherwise fetches the next input chunk: every pass keeps all 64 lanes busy, no workgroup
// barrier exists, and LDS traffic is wavefront-private (LDS executes a wavefront's instructions in order, so a
// pass reads what earlier passes of the same wavefront wrote).  The filter support a segment shares with its
// neighbours is recomputed: (2^k - 1)(hlen - 2) samples at level K-k, about 2 extra passes per level and side.
//
// Exactness: as in dwt1_fused_kernels.hpp the periodic extension is applied to the INPUT only, which equals the
// reference's per-level periodization (pdwt/src/separable.cu:114-121) when every level length is even; the host
// requires 128 * 2^K | N0 (whole passes at every level).
#pragma once

#include "dwt1_fused_kernels.hpp"  // kMaxFusedLevels
#include "kernels_common.hpp"
#include "packed_math.hpp"

namespace pdwt {

// floats per level ring: pass c reads [256c - 8, 256c + 272) while the level below has written up to 256c + 383;
// ring 0 (the input) is filled four 256-sample chunks at a time (c+1 .. c+4 while pass c still reads the tail of
// chunk c-1): eight chunks long
constexpr int kWave1Ring = 512;
constexpr int kWave1Ring0 = 2048;
constexpr int kWave1Fetch = 4;  // input chunks requested together: 4 KB in flight per wavefront
constexpr int kWave1MaxSched = 640, kWave1Bias = 512, kWave1MaxLevels = 8;

struct Fwd1DWaveArgs {
    const float* in;                // (rows, N0)
    float* det[kMaxFusedLevels];    // det[k-1] = D_k, (rows, N0 >> k)
    float* app;                     // A_K, (rows, N0 >> K)
    int rows, N0, K;
    int unitK;                      // level-K passes (of 128 outputs) per wavefront
    int units;                      // wavefronts per row = (N0 >> K) / 128 / unitK
    FilterBankI fb;                 // (dec_lo, dec_hi)
    // the pass list every wavefront runs, built once on the host (dwt1_wave_build_schedule): entry =
    // level | (relative pass index + kWave1Bias) << 4, level 0 = "move the fetched input chunks into ring 0"
    int nsched;
    unsigned sched[kWave1MaxSched];  // 32-bit entries: read with s_load (a 16-bit entry became a VECTOR load + s_waitcnt vmcnt(0) per pass)
};

// per wavefront: ring of level 0, then the rings of levels 1 .. K-1
constexpr int dwt1_wave_lds_floats(int K) { return kWave1Ring0 + (K - 1) * kWave1Ring; }
PDWT_DEVICE float* dwt1_wave_ring(float* lds, int level) { return level == 0 ? lds : lds + kWave1Ring0 + (level - 1) * kWave1Ring; }

PDWT_DEVICE int floor_div(int a, int b) {  // b > 0
    const int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

#ifdef PDWT_CPU_EMU
#define PDWT_WAVE1_LANES(lane) for (int lane = 0; lane < 64; ++lane)
#define PDWT_WAVE1_FENCE() ((void)0)
#else
#define PDWT_WAVE1_LANES(lane) for (int lane = threadIdx.x & 63, pdwt_once_ = 1; pdwt_once_; pdwt_once_ = 0)
// compiler-level ordering of the wavefront-private LDS traffic between passes.  NOT __builtin_amdgcn_fence(...,
// "wavefront"): that lowers to s_waitcnt vmcnt(0) lgkmcnt(0), i.e. every pass waited for its detail STORES to
// be acknowledged (2700 cycles per pass, 97 us for cfg3).  LDS executes a wavefront's instructions in order, so
// keeping the compiler from reordering is all that is needed.
#define PDWT_WAVE1_FENCE() asm volatile("" ::: "memory")
#endif

// One pass: level k (1-based), outputs [128 c, 128 c + 128) from ring `src` (level k-1) -> details to `det_row`,
// approximations into ring `dst` (or to `app_row` when dst is null).  `store` = this wavefront owns the outputs;
// `oc` = pass index inside the row (periodic), i.e. where the 128 outputs go.
template <int HLEN, int SRC_MASK>
PDWT_DEVICE void dwt1_wave_pass(const float* src, float* dst, float* det_row, float* app_row, int c, int oc,
                                bool store, const FilterBankI& fb) {
    constexpr int src_mask = SRC_MASK;
    constexpr int C = HLEN / 2 - 1;
    constexpr int PADL = (4 - (C & 3)) & 3;            // window origin 4 l - C - PADL is a multiple of 4
    constexpr int NV = (PADL + HLEN + 2 + 3) & ~3;     // floats read per lane
    PDWT_WAVE1_LANES(lane) {
        float v[NV];
        v4f w[NV / 4];
        const int base = 256 * c + 4 * lane - C - PADL;  // multiple of 4 (may be negative: rings are indexed mod 512)
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) w[q] = lds_load16(src + ((base + 4 * q) & src_mask));
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            lds_pin(w[q]);
            v[4 * q + 0] = w[q].x; v[4 * q + 1] = w[q].y; v[4 * q + 2] = w[q].z; v[4 * q + 3] = w[q].w;
        }
        // four accumulation chains (even / odd taps of both outputs): a wavefront runs passes one after the other,
        // so the depth of the dependent FMA chain is part of every pass's latency
        v2f e0 = mk2(0.f, 0.f), e1 = e0, o0 = e0, o1 = e0;
#pragma unroll
        for (int j = 0; j < HLEN; j += 2) {
            const v2f te = fb.t[HLEN - 1 - j], to = fb.t[HLEN - 2 - j];
            e0 = fma2(bc(v[PADL + j]), te, e0);
            e1 = fma2(bc(v[PADL + 2 + j]), te, e1);
            o0 = fma2(bc(v[PADL + j + 1]), to, o0);
            o1 = fma2(bc(v[PADL + 3 + j]), to, o1);
        }
        const v2f lh0 = mk2(e0.x + o0.x, e0.y + o0.y), lh1 = mk2(e1.x + o1.x, e1.y + o1.y);
        if (dst) {
            f32x2 a;
            a.x = lh0.x; a.y = lh1.x;
            *reinterpret_cast<f32x2*>(dst + ((128 * c + 2 * lane) & (kWave1Ring - 1))) = a;
        }
        if (store) {
            f32x2 d;
            d.x = lh0.y; d.y = lh1.y;
            *reinterpret_cast<f32x2*>(det_row + 128 * (long long)oc + 2 * lane) = d;
            if (!dst) {
                f32x2 a;
                a.x = lh0.x; a.y = lh1.x;
                *reinterpret_cast<f32x2*>(app_row + 128 * (long long)oc + 2 * lane) = a;
            }
        }
    }
    PDWT_WAVE1_FENCE();
}

// The pass list of ONE wavefront, in coordinates relative to its first owned pass of every level (identical for
// all wavefronts).  The greedy rule: run the deepest level whose next pass has its inputs -- level k pass c needs
// level k-1 up to pass 2c+2 (level 1: input chunks up to c+1), or everything that level will ever produce --
// otherwise move the next group of kWave1Fetch input chunks into ring 0.  The ranges are the exact sample ranges
// the owned level-K outputs depend on, cascaded down.  Deciding this on the device cost ~400 scalar
// instructions per pass (the kernel ran at 1.7 TB/s); as a table it costs a dozen.  Host code.
inline int dwt1_wave_build_schedule(int K, int hlen, int unitK, unsigned* sched, int cap) {
    const int C = hlen / 2 - 1;
    int next[kMaxFusedLevels + 1], last[kMaxFusedLevels + 1];
    int lo = 0, hi = 128 * unitK - 1;
    auto fdiv = [](int a, int b) { int q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; };
    for (int k = K; k >= 1; --k) {
        next[k] = fdiv(lo, 128);
        last[k] = fdiv(hi, 128);
        lo = 2 * lo - C;
        hi = 2 * hi - C + hlen - 1;
    }
    next[0] = next[1] - 1;  // level-1 pass c reads the input chunks c-1 .. c+1 (16-B aligned window)
    last[0] = last[1] + 1;
    int n = 0;
    while (next[K] <= last[K]) {
        int run = 0;
        for (int k = 1; k <= K; ++k) {
            if (next[k] > last[k]) continue;
            const int need = (k == 1) ? next[1] + 2 : 2 * next[k] + 3;
            if (next[k - 1] >= need || next[k - 1] > last[k - 1]) run = k;
        }
        // relative index: level k's owned passes start at 0 in these coordinates (unit = 0)
        const int rel = (run == 0 ? next[0] : next[run]) + kWave1Bias;
        if (n >= cap || rel < 0 || rel >= 4096) return -1;
        sched[n++] = (unsigned)(run | (rel << 4));
        if (run == 0) next[0] += kWave1Fetch;
        else ++next[run];
    }
    return n;
}

// One wavefront: row `row`, level-K passes [unit * unitK, (unit + 1) * unitK); `lds` = its rings.
template <int HLEN>
PDWT_DEVICE void dwt1_fwd_wave(const Fwd1DWaveArgs& a, int row, int unit, float* lds) {
    const int K = a.K;
    const float* PDWT_RESTRICT in = a.in + (long long)row * a.N0;
    const int nchunk0 = a.N0 >> 8;  // input chunks per row (periodic)
    const int base_unit = unit * a.unitK;
    float* det_row[kWave1MaxLevels];  // this row of every detail band (static indices below: scalar registers)
#pragma unroll
    for (int k = 1; k <= kWave1MaxLevels; ++k) det_row[k - 1] = a.det[k - 1] + (long long)row * (a.N0 >> k);
    float* app_row = a.app + (long long)row * (a.N0 >> K);

    // input chunks travel in groups of four: four 16-B loads per lane are in flight while the previous group is
    // being transformed; a fetch step moves a landed group into ring 0 and requests the next one (no register
    // copies, no branch around a load)
    PDWT_PER_THREAD(v4f, pend, kWave1Fetch, 64);
    const int first_chunk = (base_unit << (K - 1)) + ((int)(a.sched[0] >> 4) - kWave1Bias);  // entry 0 is a fetch step
    int cn = true_mod(first_chunk, nchunk0);  // next chunk to request (position in the periodic row)
    auto request_group = [&]() {
#pragma unroll
        for (int q = 0; q < kWave1Fetch; ++q) {
            PDWT_WAVE1_LANES(lane) { PDWT_MINE(pend, lane)[q] = *reinterpret_cast<const v4f*>(in + 256 * cn + 4 * lane); }
            cn = (cn + 1 == nchunk0) ? 0 : cn + 1;
        }
    };
    request_group();

    unsigned e_next = a.sched[0];
    for (int s = 0; s < a.nsched; ++s) {
        const unsigned e = e_next;
        e_next = a.sched[s + 1 < a.nsched ? s + 1 : s];  // scalar load in flight while this entry runs
        const int k = (int)(e & 15u), rel = (int)(e >> 4) - kWave1Bias;
        if (k == 0) {
            // move the fetched group of chunks from its registers into ring 0, request the next
            const int ch = (base_unit << (K - 1)) + rel;
            PDWT_WAVE1_LANES(lane) {
#pragma unroll
                for (int q = 0; q < kWave1Fetch; ++q)
                    *reinterpret_cast<v4f*>(lds + ((256 * (ch + q) + 4 * lane) & (kWave1Ring0 - 1))) = PDWT_MINE(pend, lane)[q];
            }
            PDWT_WAVE1_FENCE();
            request_group();
            continue;
        }
        const int c = (base_unit << (K - k)) + rel;
        const bool store = rel >= 0 && rel < (a.unitK << (K - k));  // owned passes lie inside the row: no wrap of c
        // one specialisation per level: ring addresses and the band pointer are compile-time / loop-invariant
        // (a run-time level cost an s_load + wait for the band pointer and ~15 address instructions per pass)
#define PDWT_W1_CASE(KK)                                                                                              \
    case KK:                                                                                                          \
        dwt1_wave_pass<HLEN, (KK == 1 ? kWave1Ring0 : kWave1Ring) - 1>(                                               \
            dwt1_wave_ring(lds, KK - 1), (KK < K) ? dwt1_wave_ring(lds, KK) : nullptr, det_row[KK - 1], app_row, c, c, \
            store, a.fb);                                                                                             \
        break;
        switch (k) {
            PDWT_W1_CASE(1) PDWT_W1_CASE(2) PDWT_W1_CASE(3) PDWT_W1_CASE(4)
            PDWT_W1_CASE(5) PDWT_W1_CASE(6) PDWT_W1_CASE(7) PDWT_W1_CASE(8)
        }
#undef PDWT_W1_CASE
    }
}

#ifndef PDWT_CPU_EMU
template <int HLEN, int NT>
__global__ void __launch_bounds__(NT) dwt1_fwd_wave_kernel(const Fwd1DWaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = (long long)blockIdx.x * (NT / 64) + wave;
    const long long total = (long long)a.rows * a.units;
    if (w >= total) return;
    const int row = (int)(w / a.units), unit = (int)(w - (long long)row * a.units);
    dwt1_fwd_wave<HLEN>(a, row, unit, pdwt_smem + wave * dwt1_wave_lds_floats(a.K));
}
#endif

}  // namespace pdwt
