// launch_dwt2_chain.hip -- K consecutive 2D DWT levels in ONE launch (dwt2_chain_kernels.hpp): the level-per-launch data flow
// without the launch boundaries.  Preconditions (dwt2_chain_supported): even filters of at most 8 taps (the 64 x 8 LDS
// tiles), every level of the group made of WHOLE tiles (rows % 16 == 0, columns % 128 == 0 at each level) and at least
// as large as a tile's staged region, 16-B aligned planes.
#include "dwt2_chain_kernels.hpp"
#include "launch.hpp"
#include "launch_util.hpp"

#include <atomic>
#include <cstdlib>

namespace pdwt {

namespace {
constexpr int TX = 64, TY = 8, NT = 256;

void interleave(FilterBankI& o, const FilterBank& fb) {
    for (int i = 0; i < kMaxTaps; i++) {
        o.t[i].x = fb.lo[i];
        o.t[i].y = fb.hi[i];
    }
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 30 us: a producer that started before its consumer has long published by then (a tile lives 4-5 us).
// pdwt_set_tuning("chain_timeout", ticks): tests set 0, which turns nearly every wait into the self-help path.
std::atomic<int>& timeout_ticks() {
    static std::atomic<int> v{lab_env("PDWT_CHAIN_TIMEOUT") ? atoi(lab_env("PDWT_CHAIN_TIMEOUT")) : 3000};
    return v;
}

template <int HLEN>
hipError_t run(ChainArgs& q, bool inverse, bool stagger, hipStream_t s) {
    q.stagger = stagger ? 1 : 0;
    static const int xcd = lab_env("PDWT_CHAIN_XCD") ? atoi(lab_env("PDWT_CHAIN_XCD")) : 1;
    q.xcd_bands = xcd;
    const int steps = stagger ? q.batch + q.nl - 1 : q.batch;
    const long long blocks = (long long)steps * q.tiles_per_image;
    if (blocks > 0x7fffffffLL) return hipErrorNotSupported;
    if (inverse) {
        constexpr size_t lds = (size_t)inv2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(float);
        hipLaunchKernelGGL((dwt2_inv_chain_kernel<HLEN, TX, TY, NT>), dim3((unsigned)blocks), dim3(NT), lds, s, q);
    } else {
        constexpr size_t lds = (size_t)fwd2d_fast_lds_floats<HLEN, TX, TY>() * sizeof(float);
        hipLaunchKernelGGL((dwt2_fwd_chain_kernel<HLEN, TX, TY, NT>), dim3((unsigned)blocks), dim3(NT), lds, s, q);
    }
    return hipGetLastError();
}
}  // namespace

int set_chain_timeout(int ticks) { return timeout_ticks().exchange(ticks < 0 ? 0 : ticks); }

// (Nr, Nc): dims entering the FINEST level of the group
bool dwt2_chain_supported(int hlen, int Nr, int Nc, int K) {
    if ((hlen & 1) || hlen < 2 || hlen > 8 || K < 2 || K > kChainMaxLevels) return false;
    if ((long long)Nr * Nc >= (1LL << 29)) return false;  // 32-bit byte offsets inside one plane (raw-buffer accesses)
    for (int k = 0; k < K; k++) {
        const int r = Nr >> k, c = Nc >> k;
        if ((r << k) != Nr || (c << k) != Nc) return false;
        if (r % (2 * TY) || c % (2 * TX)) return false;   // whole tiles
        if (r < 2 * TY + hlen - 2 || c < 2 * TX + 16) return false;  // the branch-free staging wraps at most once
    }
    return true;
}

// tiles of one image over the K levels = words of flag memory per image
int dwt2_chain_tiles(int Nr, int Nc, int K) {
    int n = 0;
    for (int k = 0; k < K; k++) n += ((Nr >> (k + 1)) / TY) * ((Nc >> (k + 1)) / TX);
    return n;
}

// Forward: in (Nr, Nc) -> level k (k = 0 finest): app[k] (A), det[3 k + {0,1,2}] (H, V, D).
hipError_t launch_dwt2_fwd_chain(const real_t* in, real_t* const* det, real_t* const* app, int Nr, int Nc, int K, int hlen,
                                 const FilterBank& fb, int batch, unsigned* flags, unsigned epoch, hipStream_t s) {
    if (!dwt2_chain_supported(hlen, Nr, Nc, K) || !flags) return hipErrorNotSupported;
    ChainArgs q{};
    q.nl = K; q.batch = batch; q.flags = flags; q.epoch = epoch; q.timeout = (unsigned)timeout_ticks().load(std::memory_order_relaxed);
    interleave(q.fb, fb);
    int first = 0, row0 = -8;
    for (int k = 0; k < K; k++) {
        ChainLevel& L = q.lv[k];
        L.in = k == 0 ? in : app[k - 1];
        L.A = app[k]; L.H = det[3 * k]; L.V = det[3 * k + 1]; L.D = det[3 * k + 2];
        L.out = nullptr;
        L.Nr = Nr >> k; L.Nc = Nc >> k; L.Nr2 = L.Nr / 2; L.Nc2 = L.Nc / 2;
        L.hi_bstride = (long long)L.Nr * L.Nc; L.lo_bstride = (long long)L.Nr2 * L.Nc2;
        L.tiles_x = L.Nc2 / TX; L.tiles_y = L.Nr2 / TY;
        L.first = first;
        first += L.tiles_x * L.tiles_y;
        // Rows are walked periodically from row0: the first rows of level k+1 read (through the periodic extension) the
        // LAST rows of level k, so level k starts a few rows before row 0 and level k+1 at the first row whose
        // producers are then all early: row0' = ceil((row0 + 1) / 2).
        L.row0 = ((row0 % L.tiles_y) + L.tiles_y) % L.tiles_y;
        row0 = row0 + 1 >= 0 ? (row0 + 2) / 2 : -((-(row0 + 1)) / 2);
        if (!al16(L.in) || !al16(L.A) || !al16(L.H) || !al16(L.V) || !al16(L.D)) return hipErrorNotSupported;
    }
    q.tiles_per_image = first;
    const bool stagger = batch >= 4;
    switch (hlen) {
        case 2: return run<2>(q, false, stagger, s);
        case 4: return run<4>(q, false, stagger, s);
        case 6: return run<6>(q, false, stagger, s);
        case 8: return run<8>(q, false, stagger, s);
    }
    return hipErrorNotSupported;
}

// Inverse of the same group: app[K-1] (the coarsest approximation, an INPUT) and det -> app[K-2] ... app[0] -> out (Nr, Nc).
hipError_t launch_dwt2_inv_chain(real_t* out, real_t* const* det, real_t* const* app, int Nr, int Nc, int K, int hlen,
                                 const FilterBank& fb, int batch, unsigned* flags, unsigned epoch, hipStream_t s) {
    if (!dwt2_chain_supported(hlen, Nr, Nc, K) || !flags) return hipErrorNotSupported;
    ChainArgs q{};
    q.nl = K; q.batch = batch; q.flags = flags; q.epoch = epoch; q.timeout = (unsigned)timeout_ticks().load(std::memory_order_relaxed);
    interleave(q.fb, fb);
    int first = 0, row0 = 0;
    for (int j = 0; j < K; j++) {  // execution order: coarsest level first
        const int k = K - 1 - j;
        ChainLevel& L = q.lv[j];
        L.in = nullptr;
        L.A = app[k]; L.H = det[3 * k]; L.V = det[3 * k + 1]; L.D = det[3 * k + 2];
        L.out = k == 0 ? out : app[k - 1];
        L.Nr = Nr >> k; L.Nc = Nc >> k; L.Nr2 = L.Nr / 2; L.Nc2 = L.Nc / 2;
        L.hi_bstride = (long long)L.Nr * L.Nc; L.lo_bstride = (long long)L.Nr2 * L.Nc2;
        L.tiles_x = L.Nc2 / TX; L.tiles_y = L.Nr2 / TY;
        L.first = first;
        first += L.tiles_x * L.tiles_y;
        // tile row y of the next (finer) level reads producer rows from floor((8 y - C) / 16): it starts at 2 row0 + 1
        L.row0 = row0 % L.tiles_y;
        row0 = 2 * row0 + 1;
        if (!al16(L.out) || !al16(L.A) || !al16(L.H) || !al16(L.V) || !al16(L.D)) return hipErrorNotSupported;
    }
    q.tiles_per_image = first;
    const bool stagger = batch >= 4;
    switch (hlen) {
        case 2: return run<2>(q, true, stagger, s);
        case 4: return run<4>(q, true, stagger, s);
        case 6: return run<6>(q, true, stagger, s);
        case 8: return run<8>(q, true, stagger, s);
    }
    return hipErrorNotSupported;
}

}  // namespace pdwt
