// CPU emulation build of the gfx950 tile functions (sanitizer / index-math fuzz only).
// Compiles pypwt_amd/csrc/*_kernels.hpp with PDWT_CPU_EMU: phases run as loops over
// thread ids, blocks as loops over the grid.  Test infrastructure; never shipped.
#define PDWT_CPU_EMU 1
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pypwt_amd/csrc/dwt1_fused_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt1_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt1_reg_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt1_rows_kernels.hpp"
#include "../../pypwt_amd/csrc/swt2_fused_kernels.hpp"
#include "../../pypwt_amd/csrc/swt2_fused4_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_fast_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_pyramid_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_pyr3_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_tail_kernels.hpp"
#include "../../pypwt_amd/csrc/swt2_tail_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_strip_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_wave_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_ring_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_long_kernels.hpp"
#include "../../pypwt_amd/csrc/nonsep_kernels.hpp"
#include "../../pypwt_amd/csrc/swt_kernels.hpp"
#include "../../pypwt_amd/csrc/swt_split_kernels.hpp"
#include "../../pypwt_amd/csrc/swt_colstream_kernels.hpp"
#include "../../pypwt_amd/csrc/swt_fwdstream_kernels.hpp"
#include "../../pypwt_amd/csrc/swt_invstream_kernels.hpp"
#include "../../pypwt_amd/csrc/strip_walk.hpp"
#include "../../pypwt_amd/csrc/swt_stream_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_stream_kernels.hpp"
#include "../../pypwt_amd/csrc/dwt2_split_kernels.hpp"

using namespace pdwt;

#define EMU_API extern "C" __attribute__((visibility("default")))

static void set_bank(FilterBank& fb, const float* lo, const float* hi, int hlen) {
    std::memset(&fb, 0, sizeof(fb));
    for (int i = 0; i < hlen; i++) { fb.lo[i] = lo[i]; fb.hi[i] = hi[i]; }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#define EMU_EVEN_HLENS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) \
    X(28) X(30) X(32) X(34) X(36) X(38) X(40)

template <int HLEN, int TX, int TY, int NT>
static void run_fwd2d(const Fwd2DArgs& a, int batch) {
    std::vector<float> smem(fwd2d_lds_floats<TX, TY>(a.hlen) + 64, -12345.f);
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < cdiv(a.Nr2, TY); by++)
            for (int bx = 0; bx < cdiv(a.Nc2, TX); bx++)
                dwt2_fwd_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
}

template <int HLEN, int TX, int TY, int NT>
static void run_inv2d(const Inv2DArgs& a, int batch) {
    std::vector<float> smem(inv2d_lds_floats<TX, TY>(a.hlen) + 64, -12345.f);
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < cdiv(a.Nr, 2 * TY); by++)
            for (int bx = 0; bx < cdiv(a.Nc, 2 * TX); bx++)
                dwt2_inv_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
}

// generic != 0 forces the runtime-length (HLEN = 0) instantiation
EMU_API int emu_dwt2_fwd(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen,
                         int generic, int tile, float* A, float* H, float* V, float* D) {
    Fwd2DArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = (Nr + 1) / 2; a.Nc2 = (Nc + 1) / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    if (generic || (hlen & 1)) {
        if (tile == 0) run_fwd2d<0, 64, 16, 256>(a, batch); else run_fwd2d<0, 64, 32, 256>(a, batch);
        return 0;
    }
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_fwd2d<h, 64, 16, 256>(a, batch); else run_fwd2d<h, 64, 32, 256>(a, batch); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

EMU_API int emu_dwt2_inv(const float* A, const float* H, const float* V, const float* D, int batch, int Nrc, int Ncc,
                         int Nr, int Nc, const float* lo, const float* hi, int hlen, int generic, int tile, float* out) {
    Inv2DArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out;
    a.Nrc = Nrc; a.Ncc = Ncc; a.Nr = Nr; a.Nc = Nc;
    a.in_bstride = (long long)Nrc * Ncc; a.out_bstride = (long long)Nr * Nc;
    a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    if (generic || (hlen & 1)) {
        if (tile == 0) run_inv2d<0, 64, 16, 256>(a, batch); else run_inv2d<0, 64, 32, 256>(a, batch);
        return 0;
    }
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_inv2d<h, 64, 16, 256>(a, batch); else run_inv2d<h, 64, 32, 256>(a, batch); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ 1D DWT
template <int HLEN, int TXO, int NT>
static void run_fwd1d(const Fwd1DArgs& a) {
    std::vector<float> smem(fwd1d_lds_floats<TXO>(a.hlen) + 64, -12345.f);
    for (int row = 0; row < a.rows; row++)
        for (int bx = 0; bx < cdiv(a.Nc2, TXO); bx++) dwt1_fwd_tile<HLEN, TXO, NT>(a, bx, row, smem.data());
}
template <int HLEN, int TXO, int NT>
static void run_inv1d(const Inv1DArgs& a) {
    std::vector<float> smem(inv1d_lds_floats<TXO>(a.hlen) + 64, -12345.f);
    for (int row = 0; row < a.rows; row++)
        for (int bx = 0; bx < cdiv(a.Nc, 2 * TXO); bx++) dwt1_inv_tile<HLEN, TXO, NT>(a, bx, row, smem.data());
}

EMU_API int emu_dwt1_fwd(const float* in, int rows, int Nc, const float* lo, const float* hi, int hlen, int generic,
                         int wide, float* L, float* H) {
    Fwd1DArgs a;
    a.in = in; a.L = L; a.H = H; a.rows = rows; a.Nc = Nc; a.Nc2 = (Nc + 1) / 2; a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    if (generic || (hlen & 1)) { if (wide) run_fwd1d<0, 1024, 256>(a); else run_fwd1d<0, 256, 256>(a); return 0; }
    switch (hlen) {
#define X(h) case h: if (wide) run_fwd1d<h, 1024, 256>(a); else run_fwd1d<h, 256, 256>(a); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

EMU_API int emu_dwt1_inv(const float* L, const float* H, int rows, int Ncc, int Nc, const float* lo, const float* hi,
                         int hlen, int generic, int wide, float* out) {
    Inv1DArgs a;
    a.L = L; a.H = H; a.out = out; a.rows = rows; a.Ncc = Ncc; a.Nc = Nc; a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    if (generic || (hlen & 1)) { if (wide) run_inv1d<0, 1024, 256>(a); else run_inv1d<0, 256, 256>(a); return 0; }
    switch (hlen) {
#define X(h) case h: if (wide) run_inv1d<h, 1024, 256>(a); else run_inv1d<h, 256, 256>(a); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ SWT
template <int HLEN, bool INV>
static void run_swt2(const Swt2DArgs& a, int batch) {
    constexpr int TX = 64, TY = 16, NT = 256;
    std::vector<float> smem(swt2d_lds_floats<TX, TY>(a.hlen) + 64, -12345.f);
    const int M = cdiv(a.Nr, a.f);  // the longest dilation phase (f need not divide Nr)
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < cdiv(M, TY) * a.f; by++)
            for (int bx = 0; bx < cdiv(a.Nc, TX); bx++) {
                if (INV) swt2_inv_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
                else swt2_fwd_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
            }
}

template <int HLEN, bool INV, int TX = 128>
static void run_swt2_vec(const Swt2DArgs& a, int batch) {
    constexpr int TY = 16, NT = 256;
    std::vector<float> smem(swt2d_inv_vec_lds_floats<TX, TY, NT>(HLEN, true) + 64, NAN);
    const int M = cdiv(a.Nr, a.f);  // the longest dilation phase (f need not divide Nr)
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < cdiv(M, TY) * a.f; by++)
            for (int bx = 0; bx < cdiv(a.Nc, TX); bx++) {
                if (INV) swt2_inv_vec_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
                else swt2_fwd_vec_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
            }
}

// inverse != 0: A,H,V,D are inputs and io is the output plane; else io is the input plane
// generic: 0 = compile-time filter length, 1 = run-time filter length, 2 = vectorised (4 columns per thread)
EMU_API int emu_swt2(int inverse, float* io, int batch, int Nr, int Nc, int level, const float* lo, const float* hi,
                     int hlen, int generic, float* A, float* H, float* V, float* D) {
    Swt2DArgs a;
    a.in = io; a.out = io; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.f = 1 << (level - 1); a.bstride = (long long)Nr * Nc; a.hlen = hlen; a.soft_beta = 0.f;
    set_bank(a.fb, lo, hi, hlen);
    if (generic == 3) {  // the 256-column tiles the host uses for hlen 2 and 4
        if (Nc < 4 || (hlen != 2 && hlen != 4)) return -3;
        if (hlen == 2) { if (inverse) run_swt2_vec<2, true, 256>(a, batch); else run_swt2_vec<2, false, 256>(a, batch); }
        else { if (inverse) run_swt2_vec<4, true, 256>(a, batch); else run_swt2_vec<4, false, 256>(a, batch); }
        return 0;
    }
    if (generic == 2) {
        if ((hlen & 1) || Nc < 4) return -3;  // rows of any length (round 5)
        switch (hlen) {
#define X(h) case h: if (inverse) run_swt2_vec<h, true>(a, batch); else run_swt2_vec<h, false>(a, batch); return 0;
            EMU_EVEN_HLENS(X)
#undef X
        }
        return -1;
    }
    if (generic || (hlen & 1)) { if (inverse) run_swt2<0, true>(a, batch); else run_swt2<0, false>(a, batch); return 0; }
    switch (hlen) {
#define X(h) case h: if (inverse) run_swt2<h, true>(a, batch); else run_swt2<h, false>(a, batch); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

EMU_API int emu_swt_pass(int inverse, const float* in0, const float* in1, int Nr, int Nc, int level, int along_y,
                         const float* lo, const float* hi, int hlen, float* out0, float* out1) {
    SwtPassArgs a;
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1;
    a.Nr = Nr; a.Nc = Nc; a.f = 1 << (level - 1); a.along_y = along_y; a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    const long long total = (long long)Nr * Nc;
    for (long long b = 0; b < (total + 255) / 256; b++) {
        if (inverse) swt_pass_inv_tile<256>(a, b, nullptr);
        else swt_pass_fwd_tile<256>(a, b, nullptr);
    }
    return 0;
}

// ------------------------------------------------------------------ tuned 2D kernels
static void set_bank_i(FilterBankI& fb, const float* lo, const float* hi, int hlen) {
    std::memset(&fb, 0, sizeof(fb));
    for (int i = 0; i < hlen; i++) { fb.t[i].x = lo[i]; fb.t[i].y = hi[i]; }
}

template <int HLEN, int TX, int TY, int NT>
static void run_fwd2d_fast(Fwd2DFastArgs a, int batch) {
    std::vector<float> smem(fwd2d_fast_lds_floats<HLEN, TX, TY>() + 64, -12345.f);
    a.tiles_x = cdiv(a.Nc2, TX); a.tiles_y = cdiv(a.Nr2, TY);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    std::vector<int> seen(a.tiles_x * a.tiles_y, 0);
    for (int bz = 0; bz < batch; bz++)
        for (int b = 0; b < 8 * chunk; b++) {
            int bx, by;
            if (!xcd_tile(b, a.tiles_x, a.tiles_y, bx, by)) continue;
            if (bz == 0) seen[by * a.tiles_x + bx]++;
            dwt2_fwd_fast_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
        }
    for (int v : seen) if (v != 1) std::abort();  // the XCD renumbering must be a bijection
}

template <int HLEN, int TX, int TY, int NT>
static void run_inv2d_fast(Inv2DFastArgs a, int batch) {
    std::vector<float> smem(inv2d_fast_lds_floats<HLEN, TX, TY>() + 64, -12345.f);
    a.tiles_x = cdiv(a.Nc, 2 * TX); a.tiles_y = cdiv(a.Nr, 2 * TY);
    const int chunk = (a.tiles_x * a.tiles_y + 7) / 8;
    for (int bz = 0; bz < batch; bz++)
        for (int b = 0; b < 8 * chunk; b++) {
            int bx, by;
            if (!xcd_tile(b, a.tiles_x, a.tiles_y, bx, by)) continue;
            dwt2_inv_fast_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
        }
}

template <int HLEN, int TX, int TY, int NT>
static void run_fwd2d_stream(Fwd2DFastArgs a, int batch, int nwg) {
    std::vector<float> smem(fwd2d_fast_lds_floats<HLEN, TX, TY>() + 64, -12345.f);
    a.tiles_x = cdiv(a.Nc2, TX); a.tiles_y = cdiv(a.Nr2, TY);
    for (int wg = 0; wg < nwg; wg++) dwt2_fwd_fast_stream<HLEN, TX, TY, NT>(a, wg, nwg, batch, smem.data());
}

// nwg: number of persistent workgroups (multiple of 8)
EMU_API int emu_dwt2_fwd_stream(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen,
                                int nwg, float* A, float* H, float* V, float* D) {
    if ((hlen & 1) || (Nc & 3) || (nwg & 7)) return -2;
    Fwd2DFastArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = (Nr + 1) / 2; a.Nc2 = Nc / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: run_fwd2d_stream<h, 64, 8, 256>(a, batch, nwg); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

EMU_API int emu_dwt2_fwd_fast(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen,
                              int tile, float* A, float* H, float* V, float* D) {
    if (hlen & 1) return -2;  // any width: rows that are not whole quads take the unaligned staging / element stores
    Fwd2DFastArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = (Nr + 1) / 2; a.Nc2 = (Nc + 1) / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_fwd2d_fast<h, 64, 16, 256>(a, batch); else if (tile == 1) run_fwd2d_fast<h, 64, 32, 512>(a, batch); else run_fwd2d_fast<h, 64, 32, 256>(a, batch); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

EMU_API int emu_dwt2_inv_fast(const float* A, const float* H, const float* V, const float* D, int batch, int Nrc,
                              int Ncc, int Nr, int Nc, const float* lo, const float* hi, int hlen, int tile, float* out) {
    if ((hlen & 1) || Nc > 2 * Ncc || Nc < 2 * Ncc - 1) return -2;
    Inv2DFastArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out;
    a.Nrc = Nrc; a.Ncc = Ncc; a.Nr = Nr; a.Nc = Nc;
    a.in_bstride = (long long)Nrc * Ncc; a.out_bstride = (long long)Nr * Nc;
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_inv2d_fast<h, 64, 16, 256>(a, batch); else if (tile == 1) run_inv2d_fast<h, 64, 8, 256>(a, batch); else run_inv2d_fast<h, 64, 32, 512>(a, batch); return 0;
        EMU_EVEN_HLENS(X)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ non-separable
// filt: 4 banks of hlen*hlen ; inverse != 0: A..D inputs, io output
EMU_API int emu_nonsep(int inverse, float* io, int batch, int Nr, int Nc, int do_swt, int level, const float* filt,
                       int hlen, float* A, float* H, float* V, float* D) {
    NonsepArgs a;
    a.in = io; a.out = io; a.A = A; a.H = H; a.V = V; a.D = D; a.filt = filt;
    a.Nr = Nr; a.Nc = Nc; a.Nrc = do_swt ? Nr : (Nr + 1) / 2; a.Ncc = do_swt ? Nc : (Nc + 1) / 2;
    a.f = 1 << (level - 1); a.do_swt = do_swt;
    a.img_bstride = (long long)Nr * Nc; a.coef_bstride = (long long)a.Nrc * a.Ncc; a.hlen = hlen;
    std::vector<float> smem(nonsep_lds_floats(hlen) + 16, -1.f);
    const long long total = inverse ? (long long)Nr * Nc : (long long)a.Nrc * a.Ncc;
    for (int bz = 0; bz < batch; bz++)
        for (long long b = 0; b < (total + 255) / 256; b++) {
            if (inverse) nonsep_inv_tile<256>(a, b, bz, smem.data());
            else nonsep_fwd_tile<256>(a, b, bz, smem.data());
        }
    return 0;
}

// ------------------------------------------------------------------ fused multi-level 1D
// outputs: det = K row-major planes concatenated in level order (each rows x (N0>>k)), app = rows x (N0>>K)
template <int HLEN, int TF, int NT>
static void run_fwd1d_fused(Fwd1DFusedArgs a) {
    std::vector<float> smem(fwd1d_fused_lds_floats(TF, HLEN, a.K) + 64, NAN);
    const int tiles = cdiv(a.N0 >> a.K, TF);
    for (int row = 0; row < a.rows; row++)
        for (int bx = 0; bx < tiles; bx++) dwt1_fwd_fused_tile<HLEN, TF, NT>(a, bx, row, smem.data());
}
template <int HLEN, int T0, int NT>
static void run_inv1d_fused(Inv1DFusedArgs a) {
    std::vector<float> smem(inv1d_fused_lds_floats(T0, HLEN, a.K) + 64, NAN);
    const int tiles = cdiv(a.N0, T0);
    for (int row = 0; row < a.rows; row++)
        for (int bx = 0; bx < tiles; bx++) dwt1_inv_fused_tile<HLEN, T0, NT>(a, bx, row, smem.data());
}

EMU_API int emu_dwt1_fused(int inverse, float* io, int rows, int N0, int K, const float* lo, const float* hi, int hlen,
                           int small, float* det, float* app) {
    // (rows of N0 % 2^(K+1) == 0: the forward stores its deepest level in pairs, the inverse stages it in pairs where it must)
    if ((hlen & 1) || (N0 & 3) || (N0 % (1 << (K + 1))) || K > kMaxFusedLevels) return -2;
    float* dptr[kMaxFusedLevels] = {};
    size_t off = 0;
    for (int k = 1; k <= K; k++) { dptr[k - 1] = det + off; off += (size_t)rows * (N0 >> k); }
    if (!inverse) {
        Fwd1DFusedArgs a;
        a.in = io; a.app = app; a.rows = rows; a.N0 = N0; a.K = K;
        for (int k = 0; k < kMaxFusedLevels; k++) a.det[k] = dptr[k];
        set_bank_i(a.fb, lo, hi, hlen);
        switch (hlen) {
#define X(h) case h: if (small) run_fwd1d_fused<h, 16, 256>(a); else run_fwd1d_fused<h, 128, 256>(a); return 0;
            EMU_EVEN_HLENS(X)
#undef X
        }
    } else {
        Inv1DFusedArgs a;
        a.out = io; a.app = app; a.rows = rows; a.N0 = N0; a.K = K;
        for (int k = 0; k < kMaxFusedLevels; k++) a.det[k] = dptr[k];
        set_bank_i(a.fb, lo, hi, hlen);
        switch (hlen) {
#define X(h) case h: if (small) run_inv1d_fused<h, 1024, 256>(a); else run_inv1d_fused<h, 8192, 256>(a); return 0;
            EMU_EVEN_HLENS(X)
#undef X
        }
    }
    return -1;
}

// ------------------------------------------------------------------ two-level pyramid (forward)
template <int HLEN, int TX2, int TY2, int NT>
static void run_fwd_pyr2(FwdPyr2Args a, int batch) {
    std::vector<float> smem(Pyr2Geom<HLEN, TX2, TY2>::LDS_FLOATS + 64, NAN);
    a.tiles_x = cdiv(a.N0c / 4, TX2); a.tiles_y = cdiv(a.N0r / 4, TY2);
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < a.tiles_y; by++)
            for (int bx = 0; bx < a.tiles_x; bx++) dwt2_fwd_pyr2_tile<HLEN, TX2, TY2, NT>(a, bx, by, bz, smem.data());
}

// l1: H,V,D planes of the first level (3 x batch x N0r/2 x N0c/2), l2: A,H,V,D of the second (4 x ...)
EMU_API int emu_dwt2_fwd_pyr2(const float* in, int batch, int N0r, int N0c, const float* lo, const float* hi, int hlen,
                              int tile, float* l1, float* l2) {
    if ((hlen & 1) || hlen > 16 || (N0c & 7) || (N0r & 3)) return -2;
    FwdPyr2Args a;
    const long long n1 = (long long)batch * (N0r / 2) * (N0c / 2), n2 = (long long)batch * (N0r / 4) * (N0c / 4);
    a.in = in; a.H1 = l1; a.V1 = l1 + n1; a.D1 = l1 + 2 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.N0r = N0r; a.N0c = N0c;
    a.in_bstride = (long long)N0r * N0c; a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_fwd_pyr2<h, 32, 4, 256>(a, batch); else run_fwd_pyr2<h, 32, 8, 256>(a, batch); return 0;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ two-level pyramid (inverse)
template <int HLEN, int TX, int TY, int NT>
static void run_inv_pyr2(InvPyr2Args a, int batch) {
    std::vector<float> smem(InvPyr2Geom<HLEN, TX, TY>::LDS_FLOATS + 64, NAN);
    a.tiles_x = cdiv(a.N0c, 2 * TX); a.tiles_y = cdiv(a.N0r, 2 * TY);
    for (int bz = 0; bz < batch; bz++)
        for (int by = 0; by < a.tiles_y; by++)
            for (int bx = 0; bx < a.tiles_x; bx++) dwt2_inv_pyr2_tile<HLEN, TX, TY, NT>(a, bx, by, bz, smem.data());
}

EMU_API int emu_dwt2_inv_pyr2(const float* l1, const float* l2, int batch, int N0r, int N0c, const float* lo,
                              const float* hi, int hlen, int tile, float* out) {
    if ((hlen & 1) || hlen > 16 || (N0c & 7) || (N0r & 3)) return -2;
    InvPyr2Args a;
    const long long n1 = (long long)batch * (N0r / 2) * (N0c / 2), n2 = (long long)batch * (N0r / 4) * (N0c / 4);
    a.H1 = l1; a.V1 = l1 + n1; a.D1 = l1 + 2 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.out = out; a.N0r = N0r; a.N0c = N0c;
    a.out_bstride = (long long)N0r * N0c; a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (tile == 0) run_inv_pyr2<h, 64, 8, 256>(a, batch); else run_inv_pyr2<h, 64, 16, 256>(a, batch); return 0;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ three-level pyramid (small images)
// det: H,V,D of level 1 (3 x batch x N0r/2 x N0c/2), then of level 2, then of level 3; app: A3 (batch x N0r/8 x N0c/8)
template <int HLEN, int T>
static void run_pyr3(Pyr3Args a, int batch, bool inverse) {
    constexpr int T0 = 8 * T, NT = 256;
    if (!inverse) {
        std::vector<float> smem(Pyr3FwdGeom<HLEN, T>::LDS + 64, NAN);
        a.tiles_x = cdiv(a.N0c / 8, T); a.tiles_y = cdiv(a.N0r / 8, T);
        for (int bz = 0; bz < batch; bz++)
            for (int by = 0; by < a.tiles_y; by++)
                for (int bx = 0; bx < a.tiles_x; bx++) dwt2_fwd_pyr3_tile<HLEN, T, NT>(a, bx, by, bz, smem.data());
    } else {
        std::vector<float> smem(Pyr3InvGeom<HLEN, T0>::LDS + 64, NAN);
        a.tiles_x = cdiv(a.N0c, T0); a.tiles_y = cdiv(a.N0r, T0);
        for (int bz = 0; bz < batch; bz++)
            for (int by = 0; by < a.tiles_y; by++)
                for (int bx = 0; bx < a.tiles_x; bx++) dwt2_inv_pyr3_tile<HLEN, T0, NT>(a, bx, by, bz, smem.data());
    }
}

EMU_API int emu_dwt2_pyr3(int inverse, float* image, int batch, int N0r, int N0c, const float* lo, const float* hi, int hlen,
                          int tile, float* det, float* app) {
    if ((hlen & 1) || hlen > 16 || (N0c & 7) || (N0r & 7)) return -2;
    Pyr3Args a;
    long long off = 0;
    for (int k = 0; k < 3; k++) {
        const long long n = (long long)batch * (N0r >> (k + 1)) * (N0c >> (k + 1));
        for (int b = 0; b < 3; b++) { a.det[k][b] = det + off; off += n; }
    }
    a.in = inverse ? app : image;
    a.out = inverse ? image : app;
    a.N0r = N0r; a.N0c = N0c;
    set_bank(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (tile == 2) run_pyr3<h, 2>(a, batch, inverse != 0); else if (tile == 4) run_pyr3<h, 4>(a, batch, inverse != 0); else run_pyr3<h, 8>(a, batch, inverse != 0); return 0;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ all remaining levels of a small approximation in one launch
// det: H,V,D of the group's level 1 (3 x batch x ceil(R0/2) x ceil(C0/2)), then of level 2, ...; app: A_K (sizes by ceil-halving)
template <int HLEN>
static void run_tail_emu(const TailArgs& a, int batch, bool inverse, int threads, float* smem) {
    for (int bz = 0; bz < batch; bz++) {
        if (inverse) { if (threads == 1024) dwt2_inv_tail_image<HLEN, 1024>(a, bz, smem); else if (threads == 64) dwt2_inv_tail_image<HLEN, 64>(a, bz, smem); else dwt2_inv_tail_image<HLEN, 256>(a, bz, smem); }
        else { if (threads == 1024) dwt2_fwd_tail_image<HLEN, 1024>(a, bz, smem); else if (threads == 64) dwt2_fwd_tail_image<HLEN, 64>(a, bz, smem); else dwt2_fwd_tail_image<HLEN, 256>(a, bz, smem); }
    }
}
template <int HLEN>
static void run_tail_emu_p2(const TailArgs& a, int batch, bool inverse, int threads, float* smem) {
    for (int bz = 0; bz < batch; bz++) {
        if (inverse) { if (threads == 1024) dwt2_inv_tail_image_p2<HLEN, 1024>(a, bz, smem); else if (threads == 64) dwt2_inv_tail_image_p2<HLEN, 64>(a, bz, smem); else dwt2_inv_tail_image_p2<HLEN, 256>(a, bz, smem); }
        else { if (threads == 1024) dwt2_fwd_tail_image_p2<HLEN, 1024>(a, bz, smem); else if (threads == 64) dwt2_fwd_tail_image_p2<HLEN, 64>(a, bz, smem); else dwt2_fwd_tail_image_p2<HLEN, 256>(a, bz, smem); }
    }
}
template <int HLEN>
static void run_tail_emu2(const TailArgs& a, int batch, bool inverse, int threads, float* smem) {
    if (a.lgR >= 0 && a.lgC >= 0) run_tail_emu_p2<HLEN>(a, batch, inverse, threads, smem);  // the mask / shift kernels
    else run_tail_emu<HLEN>(a, batch, inverse, threads, smem);                       // the general ones (unrolled == 2: also for powers of two)
}
EMU_API int emu_dwt2_tail(int inverse, float* image, int batch, int R0, int C0, int K, const float* lo, const float* hi, int hlen,
                          int threads, int unrolled, float* det, float* app) {
    auto lg2 = [](int v) { int lg = 0; while ((1 << lg) < v) lg++; return (1 << lg) == v ? lg : -1; };
    if ((hlen & 1) || K < 1 || K > kTailMaxLevels) return -2;
    if ((long long)R0 * C0 > kTailTrips * threads || (long long)R0 * C0 > kTailMaxSamples) return -2;
    TailArgs a;
    a.R0 = R0; a.C0 = C0; a.lgR = lg2(R0); a.lgC = lg2(C0); a.K = K; a.hlen = hlen;
    if (unrolled == 2) a.lgR = a.lgC = -1;  // power-of-two sizes through the general kernels too
    const size_t lds = tail_geometry(a, inverse != 0);
    if (lds * sizeof(float) > 160 * 1024) return -2;
    long long off = 0;
    for (int k = 0; k < kTailMaxLevels; k++) {
        const long long n = k < K ? (long long)batch * a.r[k + 1] * a.c[k + 1] : 0;
        for (int b = 0; b < 3; b++) { a.det[k][b] = k < K ? det + off : nullptr; off += n; }
    }
    a.in = inverse ? app : image;
    a.out = inverse ? image : app;
    set_bank(a.fb, lo, hi, hlen);
    std::vector<float> smem((lds > tail_lds_elems(R0 * C0) ? lds : tail_lds_elems(R0 * C0)) + 64, NAN);
    if (unrolled && hlen <= 8) {
        switch (hlen) {
            case 2: run_tail_emu2<2>(a, batch, inverse != 0, threads, smem.data()); return 0;
            case 4: run_tail_emu2<4>(a, batch, inverse != 0, threads, smem.data()); return 0;
            case 6: run_tail_emu2<6>(a, batch, inverse != 0, threads, smem.data()); return 0;
            case 8: run_tail_emu2<8>(a, batch, inverse != 0, threads, smem.data()); return 0;
        }
    }
    run_tail_emu2<0>(a, batch, inverse != 0, threads, smem.data());
    return 0;
}

// ------------------------------------------------------------------ all levels of G short rows per one-wavefront workgroup
// data: forward input / inverse output (rows x N0); det: D_1 (rows x N0/2), D_2, ...; app: A_K
template <int HLEN>
static void run_rows_tail_emu(const RowsTailArgs& a, bool inverse, float* smem) {
    const int blocks = (a.rows + a.G - 1) / a.G;
    for (int b = 0; b < blocks; b++) {
        if (inverse) dwt1_rows_tail_inv<HLEN, 64>(a, b, smem);
        else dwt1_rows_tail_fwd<HLEN, 64>(a, b, smem);
    }
}
EMU_API int emu_dwt1_rows_tail(int inverse, float* data, int rows, int N0, int K, int G, const float* lo, const float* hi, int hlen,
                               int unrolled, float* det, float* app) {
    if ((hlen & 1) || K < 1 || K > kRowsTailMaxLevels || G < 1 || G * N0 > kRowsTailSamples || (N0 % (1 << K))) return -2;
    RowsTailArgs a;
    long long off = 0;
    for (int k = 0; k < kRowsTailMaxLevels; k++) {
        a.det[k] = k < K ? det + off : nullptr;
        if (k < K) off += (long long)rows * (N0 >> (k + 1));
    }
    a.in = inverse ? app : data;
    a.out = inverse ? data : app;
    a.rows = rows; a.N0 = N0; a.K = K; a.G = G; a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    std::vector<float> smem(rows_tail_lds_elems(G * N0) + 64, NAN);
    if (unrolled && hlen <= 20) {
        switch (hlen) {
#define X(h) case h: run_rows_tail_emu<h>(a, inverse != 0, smem.data()); return 0;
            X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
        }
    }
    run_rows_tail_emu<0>(a, inverse != 0, smem.data());
    return 0;
}

// ------------------------------------------------------------------ the whole SWT of a tiny image in one launch
// det: H,V,D of level 1 (3 x batch x n), then of level 2, ...; app: A_L (batch x n)
EMU_API int emu_swt2_tail(int inverse, float* image, int batch, int Nr, int Nc, int L, const float* lo, const float* hi, int hlen,
                          const float* beta, float* det, float* app) {
    if (L < 1 || L > kSwtTailMaxLevels || Nr < 2 || Nc < 2 || (long long)Nr * Nc > kSwtTailMaxSamples) return -2;
    const long long n = (long long)Nr * Nc;
    SwtTailArgs a;
    for (int l = 0; l < kSwtTailMaxLevels; l++) {
        for (int b = 0; b < 3; b++) a.det[l][b] = l < L ? det + (3LL * l + b) * batch * n : nullptr;
        a.beta[l] = (beta && l < L) ? beta[l] : 0.f;
    }
    a.in = inverse ? app : image;
    a.out = inverse ? image : app;
    a.R = Nr; a.C = Nc; a.L = L; a.hlen = hlen;
    set_bank(a.fb, lo, hi, hlen);
    std::vector<float> smem(swt_tail_lds_elems((int)n, inverse != 0) + 64, NAN);
    int lgC = 0, lgR = 0;
    while ((1 << lgC) < Nc) lgC++;
    while ((1 << lgR) < Nr) lgR++;
    const bool pow2 = (1 << lgC) == Nc && (1 << lgR) == Nr;
    a.lgC = pow2 ? lgC : -1;
    a.lgR = pow2 ? lgR : -1;
    const bool general = !pow2 || (beta && beta[0] < 0);  // (a negative first threshold: powers of two through the general kernels too)
    if (beta && beta[0] < 0) a.beta[0] = 0.f;
    // the launcher's three shapes: 256 threads x 16 staging trips; 256 x 4 (images of at most 1024 samples); one wavefront x 4 (at most 256)
    const int shape = n <= kSwtTailWaveSamples ? (batch & 1 ? 2 : 1) : (n <= 1024 ? (batch & 1) : 0);
    for (int bz = 0; bz < batch; bz++) {
        if (shape == 2) {
            if (inverse) { if (!general) swt2_inv_tail_image_p2<64, 4>(a, bz, smem.data()); else swt2_inv_tail_image<64, false, 4>(a, bz, smem.data()); }
            else { if (!general) swt2_fwd_tail_image_p2<64, 4>(a, bz, smem.data()); else swt2_fwd_tail_image<64, false, 4>(a, bz, smem.data()); }
        } else if (shape == 1) {
            if (inverse) { if (!general) swt2_inv_tail_image_p2<256, 4>(a, bz, smem.data()); else swt2_inv_tail_image<256, false, 4>(a, bz, smem.data()); }
            else { if (!general) swt2_fwd_tail_image_p2<256, 4>(a, bz, smem.data()); else swt2_fwd_tail_image<256, false, 4>(a, bz, smem.data()); }
        } else {
            if (inverse) { if (!general) swt2_inv_tail_image_p2<256>(a, bz, smem.data()); else swt2_inv_tail_image<256, false>(a, bz, smem.data()); }
            else { if (!general) swt2_fwd_tail_image_p2<256>(a, bz, smem.data()); else swt2_fwd_tail_image<256, false>(a, bz, smem.data()); }
        }
    }
    return 0;
}

// ------------------------------------------------------------------ two-level streaming strips (forward)
template <int HLEN, int TX2, int NT>
static void run_fwd_strip2(FwdStrip2Args a, int batch) {
    std::vector<float> smem(Strip2Geom<HLEN, TX2>::LDS_FLOATS + 64, NAN);
    a.strips = cdiv(a.N0c / 4, TX2); a.segs = cdiv(a.N0r / 4, a.seg2);
    for (int bz = 0; bz < batch; bz++)
        for (int sg = 0; sg < a.segs; sg++)
            for (int st = 0; st < a.strips; st++) dwt2_fwd_strip2_wg<HLEN, TX2, NT>(a, st, sg, bz, smem.data());
}

EMU_API int emu_dwt2_fwd_strip2(const float* in, int batch, int N0r, int N0c, const float* lo, const float* hi, int hlen,
                                int seg2, float* l1, float* l2) {
    if ((hlen & 1) || hlen > 8 || (N0c & 7) || (N0r & 3)) return -2;
    FwdStrip2Args a;
    const long long n1 = (long long)batch * (N0r / 2) * (N0c / 2), n2 = (long long)batch * (N0r / 4) * (N0c / 4);
    a.in = in; a.H1 = l1; a.V1 = l1 + n1; a.D1 = l1 + 2 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.N0r = N0r; a.N0c = N0c; a.seg2 = seg2;
    a.in_bstride = (long long)N0r * N0c; a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: run_fwd_strip2<h, 32, 256>(a, batch); return 0;
        X(2) X(4) X(6) X(8)
#undef X
    }
    return -1;
}

template <int HLEN, int TX, int NT>
static void run_inv_strip2(InvStrip2Args a, int batch) {
    std::vector<float> smem(InvStrip2Geom<HLEN, TX>::LDS_FLOATS + 64, NAN);
    a.strips = cdiv(a.N0c / 2, TX); a.segs = cdiv(a.N0r, a.seg_rows);
    for (int bz = 0; bz < batch; bz++)
        for (int sg = 0; sg < a.segs; sg++)
            for (int st = 0; st < a.strips; st++) dwt2_inv_strip2_wg<HLEN, TX, NT>(a, st, sg, bz, smem.data());
}

EMU_API int emu_dwt2_inv_strip2(const float* l1, const float* l2, int batch, int N0r, int N0c, const float* lo,
                                const float* hi, int hlen, int seg_rows, float* out) {
    if ((hlen & 1) || hlen > 8 || (N0c & 15) || (N0r & 3)) return -2;
    InvStrip2Args a;
    const long long n1 = (long long)batch * (N0r / 2) * (N0c / 2), n2 = (long long)batch * (N0r / 4) * (N0c / 4);
    a.H1 = l1; a.V1 = l1 + n1; a.D1 = l1 + 2 * n1;
    a.A2 = l2; a.H2 = l2 + n2; a.V2 = l2 + 2 * n2; a.D2 = l2 + 3 * n2;
    a.out = out; a.N0r = N0r; a.N0c = N0c; a.seg_rows = seg_rows;
    a.out_bstride = (long long)N0r * N0c; a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    set_bank_i(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: run_inv_strip2<h, 64, 256>(a, batch); return 0;
        X(2) X(4) X(6) X(8)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ wave-per-tile 2D level kernels
static void interleave_bank(FilterBankI& o, const float* lo, const float* hi, int hlen) {
    std::memset(&o, 0, sizeof(o));
    for (int i = 0; i < hlen; i++) { o.t[i].x = lo[i]; o.t[i].y = hi[i]; }
}

template <int HLEN>
static void run_fwd_wave(const FwdWaveArgs& a, int batch, int guard) {
    for (int bz = 0; bz < batch; bz++)
        for (int seg = 0; seg < a.segs; seg++)
            for (int strip = 0; strip < a.strips; strip++) {
                if (guard) dwt2_fwd_wave<HLEN, true>(a, strip, seg, bz);
                else dwt2_fwd_wave<HLEN, false>(a, strip, seg, bz);
            }
}

// guard = 0 requires Nc % 256 == 0 and seg_out, Nr2 multiples of GR/2 (checked: returns -2 otherwise)
EMU_API int emu_dwt2_fwd_wave(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen,
                              int seg_out, int guard, float* A, float* H, float* V, float* D) {
    if ((hlen & 1) || hlen < 2 || hlen > 8 || (Nc & 3)) return -1;
    FwdWaveArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = (Nr + 1) / 2; a.Nc2 = Nc / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    a.strips = cdiv(Nc, 256); a.seg_out = seg_out; a.segs = cdiv(a.Nr2, seg_out);
    interleave_bank(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: { constexpr int G2 = FwdWaveGeom<h>::GR / 2; \
        if (!guard && ((Nc % 256) || (seg_out % G2) || (a.Nr2 % G2) || (a.Nr2 % seg_out))) return -2; \
        run_fwd_wave<h>(a, batch, guard); return 0; }
        X(2) X(4) X(6) X(8)
#undef X
    }
    return -1;
}

template <int HLEN>
static void run_inv_wave(const InvWaveArgs& a, int batch, int guard) {
    for (int bz = 0; bz < batch; bz++)
        for (int seg = 0; seg < a.segs; seg++)
            for (int strip = 0; strip < a.strips; strip++) {
                if (guard) dwt2_inv_wave<HLEN, true>(a, strip, seg, bz);
                else dwt2_inv_wave<HLEN, false>(a, strip, seg, bz);
            }
}

EMU_API int emu_dwt2_inv_wave(const float* A, const float* H, const float* V, const float* D, int batch, int Nrc, int Ncc,
                              int Nr, int Nc, const float* lo, const float* hi, int hlen, int seg_pairs, int guard,
                              float* out) {
    if ((hlen & 1) || hlen < 2 || hlen > 8 || (Ncc & 1) || Nc != 2 * Ncc) return -1;
    InvWaveArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out;
    a.Nrc = Nrc; a.Ncc = Ncc; a.Nr = Nr; a.Nc = Nc;
    a.in_bstride = (long long)Nrc * Ncc; a.out_bstride = (long long)Nr * Nc;
    a.strips = cdiv(Ncc, 128); a.seg_pairs = seg_pairs; a.segs = cdiv(Nrc, seg_pairs);
    interleave_bank(a.fb, lo, hi, hlen);
    for (int d = 0; d < hlen / 2; d++) {
        a.pl[d].x = lo[hlen - 2 - 2 * d]; a.pl[d].y = lo[hlen - 1 - 2 * d];
        a.ph[d].x = hi[hlen - 2 - 2 * d]; a.ph[d].y = hi[hlen - 1 - 2 * d];
    }
    switch (hlen) {
#define X(h) case h: { constexpr int GRI = InvWaveGeom<h>::GR; \
        if (!guard && ((Ncc % 128) || (seg_pairs % GRI) || (Nrc % GRI) || (Nrc % seg_pairs) || Nr != 2 * Nrc)) return -2; \
        run_inv_wave<h>(a, batch, guard); return 0; }
        X(2) X(4) X(6) X(8)
#undef X
    }
    return -1;
}

// ------------------------------------------------------------------ register-ring level kernels for long filters (dwt2_ring_kernels.hpp)
template <int HLEN, int CPL>
static void run_fwd_ring(const FwdWaveArgs& a, int batch) {
    std::vector<float> lds(FwdRingGeom<HLEN, CPL>::LDS_REALS, -12345.f);
    for (int bz = 0; bz < batch; bz++)
        for (int seg = 0; seg < a.segs; seg++)
            for (int strip = 0; strip < a.strips; strip++) {
                std::fill(lds.begin(), lds.end(), -12345.f);  // a wavefront never reads what another one staged
                dwt2_fwd_ring<HLEN, CPL>(a, strip, seg, bz, lds.data());
            }
}

// cpl = image columns per lane (2 or 4); any Nr, Nc % 4 == 0, any seg_out >= 1
EMU_API int emu_dwt2_fwd_ring(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen,
                              int seg_out, int cpl, float* A, float* H, float* V, float* D) {
    if ((hlen & 1) || hlen < kRingMinHlen || hlen > kRingMaxHlen || (Nc & 3) || (cpl != 2 && cpl != 4) || seg_out < 1) return -1;
    FwdWaveArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = (Nr + 1) / 2; a.Nc2 = Nc / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    a.strips = cdiv(Nc, 64 * cpl); a.seg_out = seg_out; a.segs = cdiv(a.Nr2, seg_out);
    interleave_bank(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: if (cpl == 2) run_fwd_ring<h, 2>(a, batch); else run_fwd_ring<h, 4>(a, batch); return 0;
        X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
    }
    return -1;
}

template <int HLEN, int CPL>
static void run_inv_ring(const InvRingArgs& a, int batch) {
    std::vector<float> lds(InvRingGeom<HLEN, CPL>::LDS_REALS, -12345.f);
    for (int bz = 0; bz < batch; bz++)
        for (int seg = 0; seg < a.segs; seg++)
            for (int strip = 0; strip < a.strips; strip++) {
                std::fill(lds.begin(), lds.end(), -12345.f);
                dwt2_inv_ring<HLEN, CPL>(a, strip, seg, bz, lds.data());
            }
}

// cpl = image columns per lane (2 or 4); Ncc even, Nc == 2 Ncc, Nr = 2 Nrc or 2 Nrc - 1, any seg_pairs >= 1
EMU_API int emu_dwt2_inv_ring(const float* A, const float* H, const float* V, const float* D, int batch, int Nrc, int Ncc,
                              int Nr, int Nc, const float* lo, const float* hi, int hlen, int seg_pairs, int cpl, float* out) {
    if ((hlen & 1) || hlen < kRingMinHlen || hlen > kRingMaxHlen || (Ncc & 1) || Nc != 2 * Ncc || (cpl != 2 && cpl != 4) || seg_pairs < 1)
        return -1;
    InvRingArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out;
    a.Nrc = Nrc; a.Ncc = Ncc; a.Nr = Nr; a.Nc = Nc;
    a.in_bstride = (long long)Nrc * Ncc; a.out_bstride = (long long)Nr * Nc;
    a.strips = cdiv(Ncc, 32 * cpl); a.seg_pairs = seg_pairs; a.segs = cdiv(Nrc, seg_pairs);
    interleave_bank(a.fb, lo, hi, hlen);
    for (int d = 0; d < hlen / 2; d++) {
        a.pl[d].x = lo[hlen - 2 - 2 * d]; a.pl[d].y = lo[hlen - 1 - 2 * d];
        a.ph[d].x = hi[hlen - 2 - 2 * d]; a.ph[d].y = hi[hlen - 1 - 2 * d];
    }
    switch (hlen) {
#define X(h) case h: if (cpl == 2) run_inv_ring<h, 2>(a, batch); else run_inv_ring<h, 4>(a, batch); return 0;
        X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
    }
    return -1;
}

// two forward levels per wavefront: in (N0r, N0c) -> det1 = H1|V1|D1 planes, band2 = A2|H2|V2|D2 planes
EMU_API int emu_dwt2_fwd2_wave(const float* in, int batch, int N0r, int N0c, const float* lo, const float* hi, int hlen,
                               int seg2_out, float* det1, float* band2) {
    if ((hlen & 1) || hlen < 2 || hlen > 8 || (N0r & 3) || (N0c & 15)) return -1;
    FwdWave2Args a;
    const long long q1 = (long long)batch * (N0r / 2) * (N0c / 2), q2 = (long long)batch * (N0r / 4) * (N0c / 4);
    a.in = in; a.H1 = det1; a.V1 = det1 + q1; a.D1 = det1 + 2 * q1;
    a.A2 = band2; a.H2 = band2 + q2; a.V2 = band2 + 2 * q2; a.D2 = band2 + 3 * q2;
    a.N0r = N0r; a.N0c = N0c;
    a.in_bstride = (long long)N0r * N0c; a.l1_bstride = (long long)(N0r / 2) * (N0c / 2);
    a.l2_bstride = (long long)(N0r / 4) * (N0c / 4);
    a.strips = cdiv(N0c, 240); a.seg2_out = seg2_out; a.segs = cdiv(N0r / 4, seg2_out);
    interleave_bank(a.fb, lo, hi, hlen);
    for (int bz = 0; bz < batch; bz++)
        for (int seg = 0; seg < a.segs; seg++)
            for (int strip = 0; strip < a.strips; strip++) switch (hlen) {
                case 2: dwt2_fwd2_wave<2>(a, strip, seg, bz); break;
                case 4: dwt2_fwd2_wave<4>(a, strip, seg, bz); break;
                case 6: dwt2_fwd2_wave<6>(a, strip, seg, bz); break;
                case 8: dwt2_fwd2_wave<8>(a, strip, seg, bz); break;
            }
    return 0;
}

// ------------------------------------------------------------------ 1D, up to three levels in registers
template <int HLEN, int K>
static void run_fwd1d_reg(const Fwd1DRegArgs& a) {
    const long long waves = (long long)a.rows * a.wpr;
    for (long long w = 0; w < waves; w++) dwt1_fwd_reg<HLEN, K>(a, w);
}

EMU_API int emu_dwt1_fwd_reg(const float* in, int rows, int N0, int K, const float* lo, const float* hi, int hlen,
                             int bpw, float* det, float* app) {
    if ((hlen & 1) || hlen > kReg1MaxHlen || K < 1 || K > kReg1MaxLevels || (N0 % 16) || N0 < 2048) return -2;
    Fwd1DRegArgs a;
    a.in = in; a.app = app; a.rows = rows; a.N0 = N0;
    size_t off = 0;
    for (int k = 0; k < kReg1MaxLevels; k++) a.det[k] = nullptr;
    for (int k = 1; k <= K; k++) { a.det[k - 1] = det + off; off += (size_t)rows * (N0 >> k); }
    reg1_fwd_blocks(hlen, K, N0, &a.nblk, &a.nplain);
    a.bpw = bpw;
    a.wpr = (a.nblk + bpw - 1) / bpw;
    set_bank_i(a.fb, lo, hi, hlen);
#define Y(h, k) if (hlen == h && K == k) { run_fwd1d_reg<h, k>(a); return 0; }
#define X(h) Y(h, 1) Y(h, 2) Y(h, 3)
    X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
#undef Y
    return -1;
}

template <int HLEN, int K>
static void run_inv1d_reg(const Inv1DRegArgs& a) {
    const long long waves = (long long)a.rows * a.wpr;
    std::vector<float> lds(kReg1LdsFloats, NAN);
    for (long long w = 0; w < waves; w++) dwt1_inv_reg<HLEN, K>(a, w, lds.data());
}

EMU_API int emu_dwt1_inv_reg(const float* app, const float* det, int rows, int N0, int K, const float* lo, const float* hi,
                             int hlen, int bpw, float* out) {
    if ((hlen & 1) || hlen > kReg1MaxHlen || K < 1 || K > kReg1MaxLevels || (N0 % 16) || N0 < 2048) return -2;
    Inv1DRegArgs a;
    a.app = app; a.out = out; a.rows = rows; a.N0 = N0;
    size_t off = 0;
    for (int k = 0; k < kReg1MaxLevels; k++) a.det[k] = nullptr;
    for (int k = 1; k <= K; k++) { a.det[k - 1] = det + off; off += (size_t)rows * (N0 >> k); }
    reg1_inv_blocks(hlen, K, N0, &a.nblk, &a.nplain);
    a.bpw = bpw;
    a.wpr = (a.nblk + bpw - 1) / bpw;
    set_bank_i(a.fb, lo, hi, hlen);
#define Y(h, k) if (hlen == h && K == k) { run_inv1d_reg<h, k>(a); return 0; }
#define X(h) Y(h, 1) Y(h, 2) Y(h, 3)
    X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#undef X
#undef Y
    return -1;
}

// ------------------------------------------------------------------ 2D SWT, 2-tap filters, two or three levels per launch
template <int K, int F0>
static void run_swt_fused(SwtFusedArgs& a, int batch, bool inverse, int cpl) {
    // sizes the aligned instantiations cannot take run the GEN ones (SwtWalk); `cpl` + 8 forces them on every size
    const bool gen = swt_walk_general(a.Nr, a.Nc, F0) || cpl >= 8;
    cpl &= 7;
    a.wk = swt_walk(a.Nr, a.Nc, F0, inverse ? cpl : 4);
    if (inverse) a.strips = cpl == 2 ? swt_walk_strips(a.wk, a.Nc, 2 * SwtInvGeom<K, F0, 2>::V) : swt_walk_strips(a.wk, a.Nc, 4 * SwtInvGeom<K, F0, 4>::V);
    else a.strips = swt_walk_strips(a.wk, a.Nc, 4 * SwtFusedGeom<K, F0>::V);
    a.segs = (a.wk.rows_phase + a.seg_rows - 1) / a.seg_rows;
    const long long waves = (long long)batch * a.wk.phases * a.segs * a.strips;
    for (long long w = 0; w < waves; w++) {
        if (!inverse) { if (gen) swt2_fwd_fused<K, F0, true>(a, w); else swt2_fwd_fused<K, F0, false>(a, w); }
        else if (cpl == 2) { if (gen) swt2_inv_fused<K, F0, 4, 2, true>(a, w); else swt2_inv_fused<K, F0, 4, 2, false>(a, w); }
        else { if (gen) swt2_inv_fused<K, F0, 4, 4, true>(a, w); else swt2_inv_fused<K, F0, 4, 4, false>(a, w); }
    }
}

// planes: forward  in -> det (K x [H, V, D] planes, level l0 first) and out;  inverse  in (A) + det -> out
EMU_API int emu_swt2_fused(const float* in, float* det, float* out, int batch, int Nr, int Nc, int K, int f0, int seg_rows,
                           const float* lo, const float* hi, const float* beta, int inverse, int cpl) {
    if (K < 2 || K > 3 || (f0 != 1 && f0 != 8) || Nc < 256 || Nr / f0 < (1 << K) || seg_rows % (1 << K)) return -2;
    SwtFusedArgs a;
    const long long plane = (long long)Nr * Nc;
    a.in = in; a.out = out; a.Nr = Nr; a.Nc = Nc; a.bstride = plane;
    for (int k = 0; k < kSwtFusedMaxLevels; k++) {
        a.H[k] = a.V[k] = a.D[k] = nullptr;
        a.beta[k] = (beta && k < K) ? beta[k] : 0.f;
    }
    for (int k = 0; k < K; k++) {
        a.H[k] = det + (3 * k + 0) * batch * plane;
        a.V[k] = det + (3 * k + 1) * batch * plane;
        a.D[k] = det + (3 * k + 2) * batch * plane;
    }
    a.lo[0] = lo[0]; a.lo[1] = lo[1]; a.hi[0] = hi[0]; a.hi[1] = hi[1];
    a.seg_rows = seg_rows;
    a.segs = (Nr / f0 + seg_rows - 1) / seg_rows;
#define Y(k, f) if (K == k && f0 == f) { run_swt_fused<k, f>(a, batch, inverse != 0, cpl); return 0; }
    Y(2, 1) Y(3, 1) Y(2, 8) Y(3, 8)
#undef Y
    return -1;
}

// ---- 4-tap fused SWT pairs (swt2_fused4_kernels.hpp): planes as in emu_swt2_fused with K = 2
template <int F0>
static void run_swt4(Swt4Args& a, int batch, bool inverse, bool gen) {
    using G = Swt4Geom<F0>;
    const int V = inverse ? G::Vi : G::Vf;
    a.wk = swt_walk(a.Nr, a.Nc, F0, 4);
    a.strips = swt_walk_strips(a.wk, a.Nc, 4 * V);
    a.segs = (a.wk.rows_phase + a.seg_rows - 1) / a.seg_rows;
    const long long waves = (long long)batch * a.wk.phases * a.segs * a.strips;
    for (long long w = 0; w < waves; w++) {
        if (!inverse) { if (gen) swt4_fwd_fused<F0, true>(a, w); else swt4_fwd_fused<F0, false>(a, w); }
        else { if (gen) swt4_inv_fused<F0, 4, true>(a, w); else swt4_inv_fused<F0, 4, false>(a, w); }
    }
}

EMU_API int emu_swt4_fused(const float* in, float* det, float* out, int batch, int Nr, int Nc, int f0, int seg_rows,
                           const float* lo, const float* hi, const float* beta, int inverse) {
    const bool gen = swt_walk_general(Nr, Nc, f0) || (inverse & 2);  // inverse + 2: the GEN instantiations on every size
    if ((f0 != 1 && f0 != 4) || Nc < (gen ? 256 : 64) || Nr / f0 < 8 || seg_rows % 8) return -2;
    Swt4Args a;
    const long long plane = (long long)Nr * Nc;
    a.in = in; a.out = out; a.Nr = Nr; a.Nc = Nc; a.bstride = plane;
    for (int k = 0; k < 2; k++) {
        a.H[k] = det + (3 * k + 0) * batch * plane;
        a.V[k] = det + (3 * k + 1) * batch * plane;
        a.D[k] = det + (3 * k + 2) * batch * plane;
        a.beta[k] = beta ? beta[k] : 0.f;
    }
    for (int j = 0; j < 4; j++) { a.lo[j] = lo[j]; a.hi[j] = hi[j]; }
    a.seg_rows = seg_rows;
    a.segs = (Nr / f0 + seg_rows - 1) / seg_rows;
    if (f0 == 1) run_swt4<1>(a, batch, (inverse & 1) != 0, gen);
    else run_swt4<4>(a, batch, (inverse & 1) != 0, gen);
    return 0;
}

// the column pass streamed down strips (swt_colstream_kernels.hpp) on the arguments of the register column kernels
static int g_colstream_runs = 0;
EMU_API int emu_colstream_runs() { return g_colstream_runs; }  // launches that took the strip kernels so far
template <int HLEN, bool INV>
static bool run_swt_colstream(const SwtSplitArgs& c) {
    if constexpr (HLEN < 10) {
        return false;
    } else {
        constexpr int TXC = 64, TY = 32, NT = 256, M = 8;
        using G = SwtColStreamGeom<HLEN, INV, TXC, TY>;
        SwtColStreamArgs a;
        for (int k = 0; k < 4; ++k) { a.in[k] = c.in[k]; a.out[k] = c.out[k]; }
        a.Nr = c.Nr; a.Nc = c.Nc; a.f = c.f; a.in_bstride = c.in_bstride; a.out_bstride = c.out_bstride;
        a.soft_beta = c.soft_beta; a.t = c.t;
        a.wk = swt_walk(c.Nr, c.Nc, c.f, 4);
        if ((c.Nc & 3) || a.wk.rows_phase < TY) return false;
        a.strips = (c.Nc + TXC - 1) / TXC;
        const char* e = getenv("EMU_COLSTREAM_SEG");  // rows of a chain per segment (default: two segments)
        int seg = e ? atoi(e) : (a.wk.rows_phase + 1) / 2;
        a.seg = (seg + TY - 1) / TY * TY;
        a.segs = (a.wk.rows_phase + a.seg - 1) / a.seg;
        std::vector<float> smem(G::LDS_REALS, NAN);
        ++g_colstream_runs;
        for (int bz = 0; bz < c.batch; ++bz)
            for (int py = 0; py < a.wk.phases; ++py)
                for (int sg = 0; sg < a.segs; ++sg)
                    for (int st = 0; st < a.strips; ++st) swt_colstream_wg<HLEN, INV, TXC, TY, NT, M>(a, st, py, sg, bz, smem.data());
        return true;
    }
}

// ---- host-side geometry of the strip walks and of the any-size fused groups (pure functions)
EMU_API int emu_strip_walk_seg(int rows, long long units, int ty, int warm, int slots) { return strip_walk_seg(rows, units, ty, warm, slots); }
EMU_API int emu_swt_stage_pad(int x0, int xs, int Nc) { return swt_stage_pad(x0, xs, Nc); }
EMU_API void emu_swt_walk(int Nr, int Nc, int f0, int cols_per_lane, int* out4) {
    const SwtWalk w = swt_walk(Nr, Nc, f0, cols_per_lane);
    out4[0] = w.phases; out4[1] = w.rows_phase; out4[2] = (int)w.magic; out4[3] = w.pad;
}
EMU_API int emu_swt_walk_row(int Nr, int Nc, int f0, int py, int idx) {
    const SwtWalk w = swt_walk(Nr, Nc, f0, 4);
    return swt_walk_row<true, 1>(w, Nr, py, idx * f0);
}

// ---- one forward a-trous level in ONE launch, row and column pass streamed down strips (swt_fwdstream_kernels.hpp)
template <int HLEN, int F>
static int run_swt_fwdstream(const float* in, float* A, float* H, float* V, float* D, int batch, int Nr, int Nc, const float* lo, const float* hi, int seg_rows) {
    constexpr int TXC = 64, TY = F >= 8 ? 16 : 32, NT = 256, KB = F >= 8 ? 4 : 8, M = F >= 8 ? 4 : 8;
    using G = SwtFwdStreamGeom<HLEN, F, TXC, TY>;
    SwtFwdStreamArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D; a.Nr = Nr; a.Nc = Nc; a.bstride = (long long)Nr * Nc;
    a.wk = swt_walk(Nr, Nc, F, 4);
    if (((Nc & 3) && Nc < TXC + (HLEN - 1) * F + 4) || a.wk.rows_phase < TY) return -2;
    for (int j = 0; j < HLEN; ++j) a.t.t[j] = mk2(lo[HLEN - 1 - j], hi[HLEN - 1 - j]);
    a.strips = (Nc + TXC - 1) / TXC;
    const int seg = seg_rows > 0 ? seg_rows : (a.wk.rows_phase + 1) / 2;
    a.seg = (seg + TY - 1) / TY * TY;
    a.segs = (a.wk.rows_phase + a.seg - 1) / a.seg;
    std::vector<float> smem(G::LDS_REALS, NAN);
    for (int bz = 0; bz < batch; ++bz)
        for (int py = 0; py < a.wk.phases; ++py)
            for (int sg = 0; sg < a.segs; ++sg)
                for (int st = 0; st < a.strips; ++st) {
                    std::fill(smem.begin(), smem.end(), NAN);
                    swt_fwdstream_wg<HLEN, F, TXC, TY, NT, KB, M>(a, st, py, sg, bz, smem.data());
                }
    return 0;
}

EMU_API int emu_swt2_fwdstream(const float* in, int batch, int Nr, int Nc, int level, const float* lo, const float* hi, int hlen,
                               int seg_rows, float* A, float* H, float* V, float* D) {
    const int f = 1 << (level - 1);
#define Y(h, ff) if (hlen == h && f == ff) return run_swt_fwdstream<h, ff>(in, A, H, V, D, batch, Nr, Nc, lo, hi, seg_rows);
#define X(h) Y(h, 1) Y(h, 2) Y(h, 4) Y(h, 8) Y(h, 16)
    X(6) X(8) X(10) X(12) X(16) X(20) X(26) X(40)
#undef X
#undef Y
    return -1;
}

// ---- one inverse a-trous level in ONE launch (swt_invstream_kernels.hpp)
template <int HLEN, int F>
static int run_swt_invstream(const float* A, const float* H, const float* V, const float* D, float* out, int batch, int Nr, int Nc, const float* lo,
                             const float* hi, float beta, int seg_rows) {
    constexpr bool kShort = F >= 4 || (F == 2 && HLEN > 36);  // launch_swt_invstream.hip
    constexpr int TXC = 64, TY = kShort ? 16 : 32, NT = 256, KB = kShort ? 4 : 8, M = kShort ? 4 : 8;
    using G = SwtInvStreamGeom<HLEN, F, TXC, TY>;
    SwtInvStreamArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out; a.Nr = Nr; a.Nc = Nc; a.bstride = (long long)Nr * Nc; a.soft_beta = beta;
    a.wk = swt_walk(Nr, Nc, F, 4);
    if (((Nc & 3) && Nc < TXC + (HLEN - 1) * F + 4) || a.wk.rows_phase < TY) return -2;
    for (int j = 0; j < HLEN; ++j) a.t.t[j] = mk2(lo[HLEN - 1 - j], hi[HLEN - 1 - j]);
    a.strips = (Nc + TXC - 1) / TXC;
    const int seg = seg_rows > 0 ? seg_rows : (a.wk.rows_phase + 1) / 2;
    a.seg = (seg + TY - 1) / TY * TY;
    a.segs = (a.wk.rows_phase + a.seg - 1) / a.seg;
    std::vector<float> smem(G::LDS_REALS, NAN);
    for (int bz = 0; bz < batch; ++bz)
        for (int py = 0; py < a.wk.phases; ++py)
            for (int sg = 0; sg < a.segs; ++sg)
                for (int st = 0; st < a.strips; ++st) {
                    std::fill(smem.begin(), smem.end(), NAN);
                    swt_invstream_wg<HLEN, F, TXC, TY, NT, KB, M>(a, st, py, sg, bz, smem.data());
                }
    return 0;
}

EMU_API int emu_swt2_invstream(const float* A, const float* H, const float* V, const float* D, int batch, int Nr, int Nc, int level, const float* lo,
                               const float* hi, int hlen, float beta, int seg_rows, float* out) {
    const int f = 1 << (level - 1);
#define Y(h, ff) if (hlen == h && f == ff) return run_swt_invstream<h, ff>(A, H, V, D, out, batch, Nr, Nc, lo, hi, beta, seg_rows);
#define X(h) Y(h, 1) Y(h, 2) Y(h, 4) Y(h, 8)
    X(6) X(8) X(10) X(12) X(16) X(18) X(20) X(26)  // (the product builds 6-28 taps)
#undef X
#undef Y
    return -1;
}

// ---- one a-trous level as a row pass + a column pass through scratch (swt_split_kernels.hpp); planes as in emu_swt2
template <int HLEN>
static void run_swt_split(const Swt2DArgs& a, int batch, bool inverse, float* tmp) {
    constexpr int NT = 256, R = 4;
    const long long plane = (long long)a.Nr * a.Nc;
    SwtSplitArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = batch; k.soft_beta = a.soft_beta;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2(a.fb.lo[HLEN - 1 - j], a.fb.hi[HLEN - 1 - j]);
    const int f = a.f;
    const long long col_items = split_col_waves(batch, a.Nr, a.Nc, f, R);
    const long long row_items4 = f >= 4 ? split_row_waves(batch, a.Nr, split_row_items4(a.Nc, f, R)) : 0;
    const long long row_lds = split_row_lds_waves(batch, a.Nr, a.Nc);
    std::vector<float> smem(16384, NAN);  // one workgroup's LDS
    const bool direct = getenv("EMU_SPLIT_DIRECT") != nullptr;  // dilation 4 through the kernel of the dilations >= 8 (quads f apart, no LDS)
    const bool colstream = getenv("EMU_SPLIT_COLSTREAM") != nullptr;  // the column passes through swt_colstream_kernels.hpp
    auto blocks = [](long long waves) { return (waves + NT / 64 - 1) / (NT / 64); };
    if (!inverse) {
        SwtSplitArgs r = k;
        r.in[0] = a.in; r.in_bstride = a.bstride; r.out[0] = tmp; r.out[1] = tmp + plane; r.out_bstride = 2 * plane;
        if (f == 1) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 1, NT>(r, b, smem.data());
        else if (f == 2) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 2, NT>(r, b, smem.data());
        else if (f == 4 && !direct) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 4, NT>(r, b, smem.data());
        else for (long long b = 0; b < blocks(row_items4); ++b) swt_row_fwd4_tile<HLEN, R, NT>(r, b);
        SwtSplitArgs c = k;
        c.in[0] = tmp; c.in[1] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0] = a.A; c.out[1] = a.H; c.out[2] = a.V; c.out[3] = a.D; c.out_bstride = a.bstride;
        if (colstream && run_swt_colstream<HLEN, false>(c)) return;
        for (long long b = 0; b < blocks(col_items); ++b) swt_col_fwd_tile<HLEN, R, NT>(c, b);
        return;
    }
    SwtSplitArgs c = k;
    c.in[0] = a.A; c.in[1] = a.H; c.in[2] = a.V; c.in[3] = a.D; c.in_bstride = a.bstride; c.out[0] = tmp; c.out_bstride = 2 * plane;
    if (!(colstream && run_swt_colstream<HLEN, true>(c)))
        for (long long b = 0; b < blocks(col_items); ++b) swt_col_inv_tile<HLEN, R, NT>(c, b);
    SwtSplitArgs r = k;
    r.in[0] = tmp; r.in_bstride = 2 * plane; r.out[0] = a.out; r.out_bstride = a.bstride;
    if (f == 1) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 1, 1, NT>(r, b, smem.data());
    else if (f == 2) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 2, 1, NT>(r, b, smem.data());
    else if (f == 4 && !direct) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 4, 1, NT>(r, b, smem.data());
    else for (long long b = 0; b < blocks(row_items4); ++b) swt_row_inv4_tile<HLEN, R, NT>(r, b);
}

EMU_API int emu_swt2_split(int inverse, float* io, int batch, int Nr, int Nc, int level, const float* lo, const float* hi,
                           int hlen, float soft_beta, float* A, float* H, float* V, float* D) {
    Swt2DArgs a;
    a.in = io; a.out = io; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.f = 1 << (level - 1); a.bstride = (long long)Nr * Nc; a.hlen = hlen; a.soft_beta = soft_beta;
    if ((Nc & 3) || Nc < 16 || a.f >= Nr || a.f >= Nc) return -2;
    set_bank(a.fb, lo, hi, hlen);
    std::vector<float> tmp((size_t)2 * Nr * Nc * batch + 16, NAN);
    switch (hlen) {
#define X(h) case h: run_swt_split<h>(a, batch, inverse != 0, tmp.data()); return 0;
        X(4) X(6) X(8) X(10) X(12) X(16) X(20) X(26) X(40)
#undef X
    }
    return -1;
}

// ---- one a-trous level as two launches of the any-length stream kernels (swt_stream_kernels.hpp: the fp64 library's long filters);
// planes as in emu_swt2.  nc = columns per work item (2: pairs where rows are even), R = outputs per work item
template <int NC, int R>
static void run_swt_stream(const Swt2DArgs& a, int batch, bool inverse, float* tmp) {
    constexpr int NT = 256;
    const long long plane = (long long)a.Nr * a.Nc;
    SwtStreamArgs k{};
    k.Nr = a.Nr; k.Nc = a.Nc; k.f = a.f; k.batch = batch; k.hlen = a.hlen; k.scale = 0.5f;
    for (int j = 0; j < a.hlen; ++j) {
        k.tl[kStreamPadL + j] = a.fb.lo[a.hlen - 1 - j];
        k.th[kStreamPadL + j] = a.fb.hi[a.hlen - 1 - j];
    }
    auto blocks = [](long long waves) { return (waves + NT / 64 - 1) / (NT / 64); };
    auto rows = [&](const SwtStreamArgs& r, bool syn) {
        const bool two = NC == 2 && !(a.f & 1);
        const long long n = blocks(stream_waves_x(r.problems, batch, a.Nr, a.Nc, a.f, R, two ? 2 : 1));
        for (long long b = 0; b < n; ++b) {
            if (syn) { if (two) swt_stream_tile<true, false, NC, R, NT>(r, b); else swt_stream_tile<true, false, 1, R, NT>(r, b); }
            else { if (two) swt_stream_tile<false, false, NC, R, NT>(r, b); else swt_stream_tile<false, false, 1, R, NT>(r, b); }
        }
    };
    auto cols = [&](const SwtStreamArgs& c, bool syn) {
        const long long n = blocks(stream_waves_y(c.problems, batch, a.Nr, a.Nc, a.f, R, NC));
        for (long long b = 0; b < n; ++b) {
            if (syn) swt_stream_tile<true, true, NC, R, NT>(c, b);
            else swt_stream_tile<false, true, NC, R, NT>(c, b);
        }
    };
    if (!inverse) {
        SwtStreamArgs r = k;
        r.problems = 1; r.in[0][0] = a.in; r.in_bstride = a.bstride;
        r.out[0][0] = tmp; r.out[0][1] = tmp + plane; r.out_bstride = 2 * plane;
        rows(r, false);
        SwtStreamArgs c = k;
        c.problems = 2; c.in[0][0] = tmp; c.in[1][0] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0][0] = a.A; c.out[0][1] = a.H; c.out[1][0] = a.V; c.out[1][1] = a.D; c.out_bstride = a.bstride;
        cols(c, false);
        return;
    }
    SwtStreamArgs c = k;
    c.problems = 2; c.in[0][0] = a.A; c.in[0][1] = a.H; c.in[1][0] = a.V; c.in[1][1] = a.D; c.in_bstride = a.bstride;
    c.soft[0][1] = c.soft[1][0] = c.soft[1][1] = a.soft_beta;
    c.out[0][0] = tmp; c.out[1][0] = tmp + plane; c.out_bstride = 2 * plane;
    cols(c, true);
    SwtStreamArgs r = k;
    r.problems = 1; r.in[0][0] = tmp; r.in[0][1] = tmp + plane; r.in_bstride = 2 * plane;
    r.out[0][0] = a.out; r.out_bstride = a.bstride;
    rows(r, true);
}

EMU_API int emu_swt2_stream(int inverse, float* io, int batch, int Nr, int Nc, int level, const float* lo, const float* hi,
                            int hlen, float soft_beta, float* A, float* H, float* V, float* D, int R) {
    Swt2DArgs a;
    a.in = io; a.out = io; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.f = 1 << (level - 1); a.bstride = (long long)Nr * Nc; a.hlen = hlen; a.soft_beta = soft_beta;
    if (a.f >= Nr || a.f >= Nc || hlen > kMaxTaps) return -2;
    set_bank(a.fb, lo, hi, hlen);
    std::vector<float> tmp((size_t)2 * Nr * Nc * batch + 16, NAN);
    const bool two = !(Nc & 1);  // (the emulation's planes are 8-B aligned floats: pairs need even rows only)
    if (R == 4) { if (two) run_swt_stream<2, 4>(a, batch, inverse != 0, tmp.data()); else run_swt_stream<1, 4>(a, batch, inverse != 0, tmp.data()); }
    else if (R == 8) { if (two) run_swt_stream<2, 8>(a, batch, inverse != 0, tmp.data()); else run_swt_stream<1, 8>(a, batch, inverse != 0, tmp.data()); }
    else if (R == 2) { if (two) run_swt_stream<2, 2>(a, batch, inverse != 0, tmp.data()); else run_swt_stream<1, 2>(a, batch, inverse != 0, tmp.data()); }
    else return -1;
    return 0;
}

// ---- the row kernels as the (batched) 1D transform: separate approximation / detail planes
template <int HLEN>
static void run_swt1_split(const float* in0, const float* in1, int Nr, int Nc, int f, const FilterBank& fb, bool inverse, float* out0, float* out1) {
    constexpr int NT = 256, R = 4;
    SwtSplitArgs k{};
    k.Nr = Nr; k.Nc = Nc; k.f = f; k.batch = 1;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2(fb.lo[HLEN - 1 - j], fb.hi[HLEN - 1 - j]);
    k.in[0] = in0; k.in[1] = in1; k.out[0] = out0; k.out[1] = out1;
    const long long row_items4 = f >= 4 ? split_row_waves(1, Nr, split_row_items4(Nc, f, R)) : 0;
    const long long row_lds = split_row_lds_waves(1, Nr, Nc);
    std::vector<float> smem(16384, NAN);
    auto blocks = [](long long waves) { return (waves + NT / 64 - 1) / (NT / 64); };
    if (!inverse) {
        if (f == 1) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 1, NT>(k, b, smem.data());
        else if (f == 2) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 2, NT>(k, b, smem.data());
        else if (f == 4) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_fwd_lds_tile<HLEN, 4, NT>(k, b, smem.data());
        else for (long long b = 0; b < blocks(row_items4); ++b) swt_row_fwd4_tile<HLEN, R, NT>(k, b);
        return;
    }
    if (f == 1) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 1, 2, NT>(k, b, smem.data());
    else if (f == 2) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 2, 2, NT>(k, b, smem.data());
    else if (f == 4) for (long long b = 0; b < blocks(row_lds); ++b) swt_row_inv_lds_tile<HLEN, 4, 2, NT>(k, b, smem.data());
    else for (long long b = 0; b < blocks(row_items4); ++b) swt_row_inv4p_tile<HLEN, R, NT>(k, b);
}

EMU_API int emu_swt1_split(int inverse, const float* in0, const float* in1, int Nr, int Nc, int level, const float* lo, const float* hi,
                           int hlen, float* out0, float* out1) {
    const int f = 1 << (level - 1);
    if ((Nc & 3) || Nc < 16 || f >= Nc) return -2;
    FilterBank fb;
    set_bank(fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h: run_swt1_split<h>(in0, in1, Nr, Nc, f, fb, inverse != 0, out0, out1); return 0;
        X(4) X(6) X(8) X(10) X(12) X(16) X(20) X(26) X(40)
#undef X
    }
    return -1;
}


// ---- one DECIMATED level as a row pass + a column pass through scratch (dwt2_split_kernels.hpp).  forward: in (Nr, Nc) ->
// A, H, V, D (Nr/2, Nc/2); inverse: A, H, V, D -> io (Nr, Nc).  R = output rows per work item of the column kernels.
template <int HLEN, int R>
static void run_dwt_split(bool inverse, float* io, int batch, int Nr, int Nc, const FilterBank& fb, float* A, float* H, float* V,
                          float* D, float* tmp) {
    constexpr int NT = 256;
    const int Nr2 = Nr / 2, Nc2 = Nc / 2;
    DwtSplitArgs k{};
    k.batch = batch;
    for (int j = 0; j < HLEN; ++j) k.t.t[j] = mk2(fb.lo[HLEN - 1 - j], fb.hi[HLEN - 1 - j]);
    std::vector<float> smem(16384, NAN);  // one workgroup's LDS
    auto blocks = [](long long waves) { return (waves + NT / 64 - 1) / (NT / 64); };
    const long long img = (long long)Nr * Nc, band = (long long)Nr2 * Nc2;
    if (!inverse) {
        const long long plane = (long long)Nr * Nc2;
        DwtSplitArgs r = k;
        r.rows = Nr; r.cols = Nc; r.in[0] = io; r.in_bstride = img;
        r.out[0] = tmp; r.out[1] = tmp + plane; r.out_bstride = 2 * plane;
        for (long long b = 0; b < blocks(dwt_row_waves(batch, Nr, Nc)); ++b) dwt_row_fwd_tile<HLEN, NT>(r, b, smem.data());
        DwtSplitArgs c = k;
        c.rows = Nr; c.cols = Nc2; c.in[0] = tmp; c.in[1] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0] = A; c.out[1] = H; c.out[2] = V; c.out[3] = D; c.out_bstride = band;
        for (long long b = 0; b < blocks(dwt_col_waves(batch, Nr2, Nc2, R)); ++b) dwt_col_fwd_tile<HLEN, R, NT, (R == 4 ? 8 : 2)>(c, b);
        return;
    }
    DwtSplitArgs c = k;
    c.rows = Nr2; c.cols = Nc2; c.in[0] = A; c.in[1] = H; c.in[2] = V; c.in[3] = D; c.in_bstride = band;
    c.out[0] = tmp; c.out_bstride = img;
    for (long long b = 0; b < blocks(dwt_col_waves(batch, Nr, Nc2, R)); ++b) dwt_col_inv_tile<HLEN, R, NT, (R == 4 ? 4 : (R == 2 ? 3 : 2))>(c, b);
    DwtSplitArgs r = k;
    r.rows = Nr; r.cols = Nc2; r.in[0] = tmp; r.in_bstride = img; r.out[0] = io; r.out_bstride = img;
    for (long long b = 0; b < blocks(dwt_row_waves(batch, Nr, Nc)); ++b) dwt_row_inv_tile<HLEN, NT>(r, b, smem.data());
}

EMU_API int emu_dwt2_split(int inverse, float* io, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen, int R,
                           float* A, float* H, float* V, float* D) {
    if ((Nr & 1) || (Nc & 7) || Nc < 16) return -2;
    FilterBank fb;
    set_bank(fb, lo, hi, hlen);
    std::vector<float> tmp((size_t)Nr * Nc * batch + 16, NAN);
    switch (hlen) {
#define X(h)                                                                                              \
    case h:                                                                                               \
        if (R == 2) run_dwt_split<h, 2>(inverse != 0, io, batch, Nr, Nc, fb, A, H, V, D, tmp.data());      \
        else if (R == 4) run_dwt_split<h, 4>(inverse != 0, io, batch, Nr, Nc, fb, A, H, V, D, tmp.data()); \
        else run_dwt_split<h, 8>(inverse != 0, io, batch, Nr, Nc, fb, A, H, V, D, tmp.data());             \
        return 0;
        X(10) X(12) X(14) X(16) X(20) X(22) X(26) X(38) X(40)
#undef X
    }
    return -1;
}


// ---- one DECIMATED level as two launches of the any-length stream kernels (dwt2_stream_kernels.hpp: the fp64 library's long
// filters); planes as in emu_dwt2_split.  Even sizes, even filter lengths.
template <int NC, int R>
static void run_dwt_stream(bool inverse, float* io, int batch, int Nr, int Nc, const FilterBank& fb, int hlen, float* A, float* H, float* V,
                           float* D, float* tmp) {
    constexpr int NT = 256;
    const int Nr2 = Nr / 2, Nc2 = Nc / 2;
    const long long plane = (long long)Nr * Nc2, img = (long long)Nr * Nc, band = (long long)Nr2 * Nc2;
    DwtStreamArgs k{};
    k.batch = batch; k.hlen = hlen;
    if (!inverse) {
        for (int j = 0; j < hlen; ++j) { k.t[0][kStreamPadL + j] = fb.lo[hlen - 1 - j]; k.t[1][kStreamPadL + j] = fb.hi[hlen - 1 - j]; }
    } else {
        for (int j = 0; j < hlen / 2; ++j)
            for (int par = 0; par < 2; ++par) {
                k.t[par][kStreamPadL + j] = fb.lo[hlen - 1 - (2 * j + 1 - par)];
                k.t[2 + par][kStreamPadL + j] = fb.hi[hlen - 1 - (2 * j + 1 - par)];
            }
    }
    auto blocks = [](long long waves) { return (waves + NT / 64 - 1) / (NT / 64); };
    if (!inverse) {
        DwtStreamArgs r = k;
        r.problems = 1; r.in_rows = Nr; r.in_cols = Nc; r.out_rows = Nr; r.out_cols = Nc2;
        r.in[0][0] = io; r.in_bstride = img; r.out[0][0] = tmp; r.out[0][1] = tmp + plane; r.out_bstride = 2 * plane;
        for (long long b = 0; b < blocks(dwt_stream_waves(false, 1, batch, Nr, Nc, Nc2, R, 1)); ++b) dwt_stream_tile<false, false, 1, R, NT>(r, b);
        DwtStreamArgs c = k;
        c.problems = 2; c.in_rows = Nr; c.in_cols = Nc2; c.out_rows = Nr2; c.out_cols = Nc2;
        c.in[0][0] = tmp; c.in[1][0] = tmp + plane; c.in_bstride = 2 * plane;
        c.out[0][0] = A; c.out[0][1] = H; c.out[1][0] = V; c.out[1][1] = D; c.out_bstride = band;
        for (long long b = 0; b < blocks(dwt_stream_waves(true, 2, batch, Nr, Nc2, Nr2, R, NC)); ++b) dwt_stream_tile<false, true, NC, R, NT>(c, b);
        return;
    }
    DwtStreamArgs c = k;
    c.problems = 2; c.in_rows = Nr2; c.in_cols = Nc2; c.out_rows = Nr; c.out_cols = Nc2;
    c.in[0][0] = A; c.in[0][1] = H; c.in[1][0] = V; c.in[1][1] = D; c.in_bstride = band;
    c.out[0][0] = tmp; c.out[1][0] = tmp + plane; c.out_bstride = 2 * plane;
    for (long long b = 0; b < blocks(dwt_stream_waves(true, 2, batch, Nr2, Nc2, Nr2, R, NC)); ++b) dwt_stream_tile<true, true, NC, R, NT>(c, b);
    DwtStreamArgs r = k;
    r.problems = 1; r.in_rows = Nr; r.in_cols = Nc2; r.out_rows = Nr; r.out_cols = Nc;
    r.in[0][0] = tmp; r.in[0][1] = tmp + plane; r.in_bstride = 2 * plane; r.out[0][0] = io; r.out_bstride = img;
    for (long long b = 0; b < blocks(dwt_stream_waves(false, 1, batch, Nr, Nc2, Nc2, R, 1)); ++b) dwt_stream_tile<true, false, 1, R, NT>(r, b);
}

EMU_API int emu_dwt2_stream(int inverse, float* io, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen, int R,
                            float* A, float* H, float* V, float* D) {
    if ((Nr & 1) || (Nc & 1) || (hlen & 1) || hlen > kMaxTaps) return -2;
    FilterBank fb;
    set_bank(fb, lo, hi, hlen);
    std::vector<float> tmp((size_t)Nr * Nc * batch + 16, NAN);
    const bool two = !((Nc / 2) & 1);
    if (R == 4) { if (two) run_dwt_stream<2, 4>(inverse != 0, io, batch, Nr, Nc, fb, hlen, A, H, V, D, tmp.data()); else run_dwt_stream<1, 4>(inverse != 0, io, batch, Nr, Nc, fb, hlen, A, H, V, D, tmp.data()); }
    else if (R == 2) { if (two) run_dwt_stream<2, 2>(inverse != 0, io, batch, Nr, Nc, fb, hlen, A, H, V, D, tmp.data()); else run_dwt_stream<1, 2>(inverse != 0, io, batch, Nr, Nc, fb, hlen, A, H, V, D, tmp.data()); }
    else return -1;
    return 0;
}

// ------------------------------------------------------------------ strip-streaming level kernels for long filters (dwt2_long_kernels.hpp)
// shape: 0 = <64, 16, 256, KB 2 (inverse 4), M 4, XB 1> (a launch shape), 1 = <32, 8, 128, 2 (4), 2, 2>, 2 = <16, 4, 64, 4, 4, 1>,
// 3 = <64, 16, 512, 4, 2, 2>
template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB>
static void run_fwd_long(FwdLongArgs a, int batch, int seg) {
    using G = FwdLongGeom<HLEN, TXC, TY>;
    a.strips = cdiv(a.Nc2, TXC);
    a.seg = cdiv(seg < 1 ? a.Nr2 : seg, TY) * TY;
    a.segs = cdiv(a.Nr2, a.seg);
    std::vector<float> lds(G::LDS_REALS, NAN);
    const int chunk = (a.strips * a.segs + 7) / 8;
    for (int bz = 0; bz < batch; bz++)
        for (int b = 0; b < 8 * chunk; b++) {
            int strip, sg;
            if (!xcd_tile(b, a.strips, a.segs, strip, sg)) continue;
            std::fill(lds.begin(), lds.end(), NAN);  // a workgroup never reads what another one left
            dwt2_fwd_long_wg<HLEN, TXC, TY, NT, KB, M, XB>(a, strip, sg, bz, lds.data());
        }
}

// Nr, Nc even, Nc % 4 == 0, Nr >= 2 TY of the shape; seg = output rows per segment (rounded up to whole steps; < 1: one segment)
EMU_API int emu_dwt2_fwd_long(const float* in, int batch, int Nr, int Nc, const float* lo, const float* hi, int hlen, int seg,
                              int shape, float* A, float* H, float* V, float* D) {
    if ((hlen & 1) || hlen < kLongMinHlen || hlen > kMaxTaps || (Nr & 1) || (Nc & 3)) return -1;
    const int ty = shape == 1 ? 8 : (shape == 2 ? 4 : 16);
    if (Nr < 2 * ty) return -2;
    FwdLongArgs a;
    a.in = in; a.A = A; a.H = H; a.V = V; a.D = D;
    a.Nr = Nr; a.Nc = Nc; a.Nr2 = Nr / 2; a.Nc2 = Nc / 2;
    a.in_bstride = (long long)Nr * Nc; a.out_bstride = (long long)a.Nr2 * a.Nc2;
    interleave_bank(a.fb, lo, hi, hlen);
    switch (hlen) {
#define X(h) case h:                                                            \
        if (shape == 0) run_fwd_long<h, 64, 16, 256, 2, 4, 1>(a, batch, seg);    \
        else if (shape == 1) run_fwd_long<h, 32, 8, 128, 2, 2, 2>(a, batch, seg); \
        else if (shape == 2) run_fwd_long<h, 16, 4, 64, 4, 4, 1>(a, batch, seg);  \
        else run_fwd_long<h, 64, 16, 512, 4, 2, 2>(a, batch, seg);               \
        return 0;
        X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
#undef X
    }
    return -1;
}

template <int HLEN, int TXC, int TY, int NT, int KB, int M, int XB>
static void run_inv_long(InvLongArgs a, int batch, int seg, const float* lo, const float* hi) {
    using G = InvLongGeom<HLEN, TXC, TY>;
    a.strips = cdiv(a.Ncc, TXC);
    a.seg = cdiv(seg < 1 ? a.Nrc : seg, TY) * TY;
    a.segs = cdiv(a.Nrc, a.seg);
    long_syn_tables<HLEN>(a, lo, hi);
    std::vector<float> lds(G::LDS_REALS, NAN);
    const int chunk = (a.strips * a.segs + 7) / 8;
    for (int bz = 0; bz < batch; bz++)
        for (int b = 0; b < 8 * chunk; b++) {
            int strip, sg;
            if (!xcd_tile(b, a.strips, a.segs, strip, sg)) continue;
            std::fill(lds.begin(), lds.end(), NAN);
            dwt2_inv_long_wg<HLEN, TXC, TY, NT, KB, M, XB>(a, strip, sg, bz, lds.data());
        }
}

// Ncc % 4 == 0, Nc == 2 Ncc, Nr == 2 Nrc, Nrc >= TY of the shape; seg = coefficient rows per segment
EMU_API int emu_dwt2_inv_long(const float* A, const float* H, const float* V, const float* D, int batch, int Nrc, int Ncc,
                              int Nr, int Nc, const float* lo, const float* hi, int hlen, int seg, int shape, float* out) {
    if ((hlen & 1) || hlen < kLongMinHlen || hlen > kMaxTaps || (Ncc & 3) || Nc != 2 * Ncc || Nr != 2 * Nrc) return -1;
    const int ty = shape == 1 ? 8 : (shape == 2 ? 4 : 16);
    if (Nrc < ty) return -2;
    InvLongArgs a;
    a.A = A; a.H = H; a.V = V; a.D = D; a.out = out;
    a.Nrc = Nrc; a.Ncc = Ncc; a.Nr = Nr; a.Nc = Nc;
    a.in_bstride = (long long)Nrc * Ncc; a.out_bstride = (long long)Nr * Nc;
    std::vector<float> plo(kMaxTaps, 0.f), phi(kMaxTaps, 0.f);
    for (int i = 0; i < hlen; i++) { plo[i] = lo[i]; phi[i] = hi[i]; }
    switch (hlen) {
#define X(h) case h:                                                            \
        if (shape == 0) run_inv_long<h, 64, 16, 256, 4, 4, 1>(a, batch, seg, plo.data(), phi.data());    \
        else if (shape == 1) run_inv_long<h, 32, 8, 128, 4, 2, 2>(a, batch, seg, plo.data(), phi.data()); \
        else if (shape == 2) run_inv_long<h, 16, 4, 64, 4, 4, 1>(a, batch, seg, plo.data(), phi.data());  \
        else run_inv_long<h, 64, 16, 512, 8, 2, 2>(a, batch, seg, plo.data(), phi.data());               \
        return 0;
        X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
#undef X
    }
    return -1;
}
