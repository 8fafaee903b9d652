// dwt2_chain_kernels.hpp -- SEVERAL consecutive 2D DWT levels in ONE launch, level by level, for gfx950.
//
// Why: a level-per-launch forward of one cache-resident 4096^2 image spends 13 of its 35 us in levels 2-4 (25 % of the
// bytes): every level is its own dependent launch -- drain, 1.7-1.9 us of boundary, ramp-up -- and the small levels never
// fill the chip.  Fusing levels by recomputing halos (tile pyramids, two-level wavefronts, streaming strips) was measured
// slower for one image.  This kernel keeps the level-per-launch DATA FLOW (every A_l is written once and
// read once, no halo recomputation, the tuned one-tile-per-workgroup LDS tiles of dwt2_fast_kernels.hpp unchanged) and
// removes the launch boundaries: the grid is the tiles of ALL levels, in level order; a tile of level l+1 starts when the
// tiles of level l it reads have published themselves.  tools/queuebench.hip is the measured proxy (profiles/r03*_queuebench*):
// four levels 41.3 us as four launches, 37.8 us as one.
//
// Hand-off (MI355X_MICROARCH.md, "inter-workgroup visibility", valid forms; checked word by word in tools/queuebench.hip
// and by the parity tests): the producer writes the handed-off plane (A_l) with 16-B `sc1` write-through stores, every
// storing wave drains them (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane stores the tile's flag
// (= the launch's epoch) with an `sc1` store.  The consumer polls the flags of its (at most 4 x 4) producer tiles with
// `sc1` loads from one wavefront, the other waves wait at a workgroup barrier, then EVERY load of the handed-off plane is
// an `sc1` load.  Flags are per (image, level, tile) words and hold the epoch of the last launch that completed the tile,
// so nothing is reset between launches.
//
// Placement independence: HIP promises no dispatch order.  In practice workgroups start in block-id order, so a tile's
// producers (smaller ids) are running or done when it starts, and the wait is short.  Correctness does not depend on
// that: a wait is bounded (`timeout` ticks of the 100 MHz s_memrealtime), and a tile whose producer has not published
// in time computes that producer ITSELF (recursively down to level 1, whose tiles depend on nothing) -- the values are
// identical, so the duplicate stores are benign -- and carries on.  Every wait therefore ends; no schedule can deadlock.
//
// Batches: the grid is (batch + levels - 1) "steps" of one image's tiles; step s runs level index k of image s - k, so a
// producer level precedes its consumer by a whole image of work (the waits hit published flags) and A_l is read back
// from the Infinity Cache a few tens of microseconds after it was written instead of from HBM a whole batch later.
#pragma once

#include "dwt2_fast_kernels.hpp"

namespace pdwt {

constexpr int kChainMaxLevels = 6;

struct ChainLevel {
    // forward: in -> A,H,V,D ; inverse: A,H,V,D -> out.  (Nr, Nc) are the full-resolution dims of the level,
    // (Nr2, Nc2) the coefficient dims.
    const float* in;
    float *A, *H, *V, *D;
    float* out;
    int Nr, Nc, Nr2, Nc2;
    long long hi_bstride, lo_bstride;  // elements between images: full-resolution plane / coefficient planes
    int tiles_x, tiles_y;
    int first;  // position of the level's first tile inside a step
    int row0;   // tile row the level starts with (rows are walked periodically from here, see launch_dwt2_chain.hip)
};

struct ChainArgs {
    ChainLevel lv[kChainMaxLevels];  // in EXECUTION order: forward finest level first, inverse coarsest first
    int nl, batch, tiles_per_image;
    int stagger;       // 1: step s runs level index k of image s - k (batches); 0: image after image
    int xcd_bands;     // 1: within a level, XCD x (block id mod 8) walks its own contiguous eighth of the tiles
    unsigned* flags;   // [batch][tiles_per_image]
    unsigned epoch;
    unsigned timeout;  // s_memrealtime ticks (100 MHz) a tile waits for its producers before it helps itself
    FilterBankI fb;
};

#ifndef PDWT_CPU_EMU

PDWT_DEVICE int chain_floordiv(int a, int b) {  // b > 0
    const int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

// Wave 0 polls the flags of the producer tiles rows [rlo, rlo + nr) x columns [clo, clo + nc) (periodic) of one image's
// level; returns -1 when all carry this launch's epoch, else (row << 16 | column) of one that did not within `timeout`.
PDWT_DEVICE int chain_wait(const unsigned* flags, int tiles_y, int tiles_x, int rlo, int nr, int clo, int nc, unsigned epoch,
                           unsigned timeout, int* s_word) {
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        const bool need = i < nr * nc;
        const int ir = i / nc, ic = i - ir * nc;
        const int pr = true_mod(rlo + ir, tiles_y), pc = true_mod(clo + ic, tiles_x);
        const unsigned* f = flags + (need ? pr * tiles_x + pc : 0);
        bool ok = !need;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (!ok) ok = (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) break;
            __builtin_amdgcn_s_sleep(4);
        }
        const unsigned long long bad = __ballot(!ok);
        if (i == 0) {
            int w = -1;
            if (bad) {
                const int j = __builtin_ctzll(bad);
                const int jr = j / nc, jc = j - jr * nc;
                w = true_mod(rlo + jr, tiles_y) << 16 | true_mod(clo + jc, tiles_x);
            }
            *s_word = w;
        }
    }
    __syncthreads();
    return *s_word;
}

// every storing wave drains its stores, the workgroup meets, ONE lane publishes the tile
PDWT_DEVICE void chain_publish(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block id -> (level index, tile, image); false for the idle blocks of the first / last steps of a batch
PDWT_DEVICE bool chain_decode(const ChainArgs& q, int block, int& li, int& by, int& bx, int& bz) {
    const int n = q.tiles_per_image;
    const int step = block / n, pos = block - step * n;
    li = 0;
    for (int k = 1; k < q.nl; ++k)
        if (pos >= q.lv[k].first) li = k;
    bz = q.stagger ? step - li : step;
    if (bz < 0 || bz >= q.batch) return false;
    const ChainLevel& L = q.lv[li];
    int t = pos - L.first;
    const int nt = L.tiles_x * L.tiles_y;
    // blocks b and b + 8 share an XCD: give each XCD a contiguous band of the level (vertical AND horizontal neighbours,
    // which share halo lines, then meet in one L2), as the single-level kernels do (xcd_tile)
    if (q.xcd_bands && !(nt & 7) && !(L.first & 7)) t = (t & 7) * (nt >> 3) + (t >> 3);
    const int r = t / L.tiles_x;
    bx = t - r * L.tiles_x;
    by = r + L.row0;
    if (by >= L.tiles_y) by -= L.tiles_y;
    return true;
}

// The work loop shared by both directions: run `cur`; if a producer of `cur` has not published in time, run that
// producer first (and its producers ...).  Everything here is workgroup-uniform.
template <class Deps, class Run>
PDWT_DEVICE void chain_run(const ChainArgs& q, int li, int by, int bx, int bz, Deps deps, Run run) {
    __shared__ int s_word;
    __shared__ int s_stack[3 * kChainMaxLevels];
    int sp = 0;
    unsigned* flags = q.flags + (long long)bz * q.tiles_per_image;
    for (;;) {
        int miss = -1;
        if (li > 0) {
            int rlo, nr, clo, nc;
            deps(li, by, bx, rlo, nr, clo, nc);
            const ChainLevel& P = q.lv[li - 1];
            miss = chain_wait(flags + P.first, P.tiles_y, P.tiles_x, rlo, nr, clo, nc, q.epoch, q.timeout, &s_word);
        }
        if (miss >= 0) {  // help ourselves: the producer first, then this tile again
            if (threadIdx.x == 0) {
                s_stack[3 * sp] = li; s_stack[3 * sp + 1] = by; s_stack[3 * sp + 2] = bx;
            }
            ++sp;
            li -= 1; by = miss >> 16; bx = miss & 0xffff;
            __syncthreads();
            continue;
        }
        run(li, by, bx, bz);
        if (li + 1 < q.nl) chain_publish(flags + q.lv[li].first + by * q.lv[li].tiles_x + bx, q.epoch);
        if (sp == 0) break;
        --sp;
        __syncthreads();  // LDS tile and s_stack are reused
        li = s_stack[3 * sp]; by = s_stack[3 * sp + 1]; bx = s_stack[3 * sp + 2];
    }
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_fwd_chain_kernel(const ChainArgs q) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    using G = FwdFastGeom<HLEN, TX>;
    int li, by, bx, bz;
    if (!chain_decode(q, blockIdx.x, li, by, bx, bz)) return;
    chain_run(
        q, li, by, bx, bz,
        [&](int l, int y, int x, int& rlo, int& nr, int& clo, int& nc) {
            // level l stages rows [2 TY y - C, + 2 TY + HLEN - 2) x columns [2 TX x - C - PADL, + RXA) of A_(l-1), whose
            // tiles are TY x TX
            const int y0 = 2 * y * TY - G::C, xa = 2 * x * TX - G::C - G::PADL;
            rlo = chain_floordiv(y0, TY);
            nr = chain_floordiv(y0 + 2 * TY + HLEN - 3, TY) - rlo + 1;
            clo = chain_floordiv(xa, TX);
            nc = chain_floordiv(xa + G::RXA - 1, TX) - clo + 1;
        },
        [&](int l, int y, int x, int z) {
            const ChainLevel& L = q.lv[l];
            Fwd2DFastArgs a;
            a.in = L.in; a.A = L.A; a.H = L.H; a.V = L.V; a.D = L.D;
            a.Nr = L.Nr; a.Nc = L.Nc; a.Nr2 = L.Nr2; a.Nc2 = L.Nc2;
            a.in_bstride = L.hi_bstride; a.out_bstride = L.lo_bstride;
            a.tiles_x = L.tiles_x; a.tiles_y = L.tiles_y;
            a.fb = q.fb;
            dwt2_fwd_fast_tile<HLEN, TX, TY, NT>(a, x, y, z, pdwt_smem, l > 0, l + 1 < q.nl);
        });
}

template <int HLEN, int TX, int TY, int NT>
__global__ void __launch_bounds__(NT) dwt2_inv_chain_kernel(const ChainArgs q) {
    extern __shared__ __attribute__((aligned(16))) float pdwt_smem[];
    using G = InvFastGeom<HLEN, TX>;
    int li, by, bx, bz;
    if (!chain_decode(q, blockIdx.x, li, by, bx, bz)) return;
    chain_run(
        q, li, by, bx, bz,
        [&](int l, int y, int x, int& rlo, int& nr, int& clo, int& nc) {
            // level index l stages coefficient rows [TY y - C, + TY + H2 + 1) x columns [TX x - C - PADL, + CXA) of its A
            // plane = the output of level index l-1, whose tiles are 2 TY x 2 TX
            const int cy0 = y * TY - G::C, cxa = x * TX - G::C - G::PADL;
            rlo = chain_floordiv(cy0, 2 * TY);
            nr = chain_floordiv(cy0 + TY + G::H2, 2 * TY) - rlo + 1;
            clo = chain_floordiv(cxa, 2 * TX);
            nc = chain_floordiv(cxa + G::CXA - 1, 2 * TX) - clo + 1;
        },
        [&](int l, int y, int x, int z) {
            const ChainLevel& L = q.lv[l];
            Inv2DFastArgs a;
            a.A = L.A; a.H = L.H; a.V = L.V; a.D = L.D; a.out = L.out;
            a.Nrc = L.Nr2; a.Ncc = L.Nc2; a.Nr = L.Nr; a.Nc = L.Nc;
            a.in_bstride = L.lo_bstride; a.out_bstride = L.hi_bstride;
            a.tiles_x = L.tiles_x; a.tiles_y = L.tiles_y;
            a.fb = q.fb;
            dwt2_inv_fast_tile<HLEN, TX, TY, NT>(a, x, y, z, pdwt_smem, l > 0, l + 1 < q.nl);
        });
}

#endif  // !PDWT_CPU_EMU

}  // namespace pdwt
