// plan.cpp -- C ABI (include/pypwt_amd.h) over the gfx950 kernels.
//
// Host-side counterpart of the reference's class Wavelets (pdwt/src/wt.cu) and its level-loop
// drivers (pdwt/src/separable.cu:179-236, :332-395, :496-537, :629-672).  Each function cites
// the reference member it replaces.  No CPU fallback exists: without a GPU every entry point
// that touches data fails with PDWT_ERR_HIP.
#include "plan.hpp"
#include "tuning.hpp"

#include <mutex>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>
#include <strings.h>

#include "launch.hpp"
#include "launch_util.hpp"
#include "nonsep_kernels.hpp"
#include "wavelet_table.hpp"

using namespace pdwt;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(PDWT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                  \
    } while (0)

#define CHECK_HANDLE(h) \
    if (!(h)) return fail(PDWT_ERR_ARG, "null plan handle")

int div2(int n) { return (n + (n & 1)) / 2; }  // pdwt/src/utils.cu:24-27

int ilog2(int i) {  // pdwt/src/utils.cu:14-20 (terminates for i <= 0 too)
    int l = 0;
    while (i > 1) {
        i >>= 1;
        ++l;
    }
    return l;
}

long long pad64(long long n) { return (n + 63) & ~63LL; }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------- device-memory / stream pool
// The reference's users often build one Wavelets object per image (its tests and tutorials do); hipMalloc + hipFree +
// hipStreamCreate/Destroy then cost 3.4 ms per object (tools/createtime.py) -- twenty times the transform of a 512^2
// image.  Destroyed plans therefore hand their device blocks and their stream to a small process-wide pool and new plans
// take a block of at least their size and at most 1.25x of it from there.  Only SMALL blocks are kept -- at most
// PDWT_POOL_BLOCK_MB (default 64 MiB: the arena of a 2048^2 plan is 37 MiB) each, PDWT_POOL_MB (default 256 MiB) and 64
// blocks in total, 8 streams; a plan has been synchronised before it gets here: the 3.4 ms matter next to a small
// transform only, and the arena of a large plan must go back to the driver when the plan is destroyed (another
// allocator in the process -- torch, the caller's own hipMalloc -- never sees this pool).  Every hipMalloc of the
// library goes through device_malloc(), which releases the pool and retries once when the driver is out of memory.
// pdwt_trim_pool() releases everything.
struct DevicePool {
    struct Block { int dev; size_t bytes; void* p; };
    std::mutex m;
    std::vector<Block> blocks;
    std::vector<std::pair<int, hipStream_t>> streams;
    size_t cached = 0;
    size_t limit = [] { const char* e = getenv("PDWT_POOL_MB"); return (size_t)(e ? atoll(e) : 256) << 20; }();
    size_t block_limit = [] { const char* e = getenv("PDWT_POOL_BLOCK_MB"); return (size_t)(e ? atoll(e) : 64) << 20; }();
};
DevicePool& device_pool() {
    static DevicePool* p = new DevicePool();  // intentionally leaked: no HIP calls from static destructors
    return *p;
}
// hipMalloc for every allocation of the library: out of memory -> give the cached blocks back and try once more
hipError_t device_malloc(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    DevicePool& P = device_pool();
    std::vector<DevicePool::Block> drop;
    {
        std::lock_guard<std::mutex> g(P.m);
        drop.swap(P.blocks);
        P.cached = 0;
    }
    if (drop.empty()) return e;
    for (auto& b : drop) (void)hipFree(b.p);
    return hipMalloc(out, bytes);
}
hipError_t pool_alloc(int dev, void** out, size_t bytes, size_t* got) {
    DevicePool& P = device_pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        int best = -1;
        for (int i = 0; i < (int)P.blocks.size(); i++) {
            const auto& b = P.blocks[i];
            if (b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
                (best < 0 || b.bytes < P.blocks[best].bytes))
                best = i;
        }
        if (best >= 0) {
            *out = P.blocks[best].p;
            *got = P.blocks[best].bytes;
            P.cached -= P.blocks[best].bytes;
            P.blocks.erase(P.blocks.begin() + best);
            return hipSuccess;
        }
    }
    *got = bytes;
    return device_malloc(out, bytes);
}
void pool_free(int dev, void* p, size_t bytes) {
    if (!p) return;
    DevicePool& P = device_pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        if (bytes > 0 && bytes <= P.block_limit && P.cached + bytes <= P.limit && P.blocks.size() < 64) {
            P.blocks.push_back({dev, bytes, p});
            P.cached += bytes;
            return;
        }
    }
    (void)hipFree(p);
}
hipError_t pool_stream(int dev, hipStream_t* s) {
    DevicePool& P = device_pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        for (size_t i = 0; i < P.streams.size(); i++)
            if (P.streams[i].first == dev) {
                *s = P.streams[i].second;
                P.streams.erase(P.streams.begin() + i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void pool_return_stream(int dev, hipStream_t s) {
    DevicePool& P = device_pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        if (P.streams.size() < 8) {
            P.streams.push_back({dev, s});
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

// Per-kernel timing: ONE event is recorded on the plan's stream in front of every launch; a
// launch's duration is the distance to the next event (the next launch's, or the closing event
// recorded by pdwt_kernel_times).  One event per launch instead of a start/stop pair halves the
// perturbation (an event costs about 1 us of stream time on MI355X).
struct Stamp {
    pdwt_plan* plan = nullptr;
    size_t index = 0;
    bool pushed = false;  // this object's entry is plan->stamps[index] (a failed hipEventCreate pushes nothing)
    const hipError_t* verdict = nullptr;  // the launcher's answer, when the caller has one: a decline takes the entry back
    Stamp(pdwt_plan* p, const char* name, const hipError_t* launch_result = nullptr) : verdict(launch_result) {
        if (!p->timing) return;
        KernelStamp s;
        s.stop = nullptr;
        s.family[0] = 0;
        if (hipEventCreate(&s.start) != hipSuccess) return;
        snprintf(s.name, sizeof(s.name), "%s", name);
        (void)hipEventRecord(s.start, p->stream);
        note_family("");
        plan = p;
        index = p->stamps.size();
        pushed = true;
        p->stamps.push_back(s);
    }
    // the launcher declined (hipErrorNotSupported) and the step falls back to other launches: take THIS entry back
    void drop() {
        if (!pushed || index + 1 != plan->stamps.size()) return;
        (void)hipEventDestroy(plan->stamps.back().start);
        plan->stamps.pop_back();
        pushed = false;
    }
    ~Stamp() {  // after the launch: which kernel family served it
        if (verdict && *verdict == hipErrorNotSupported) drop();  // the step falls back to level launches, which stamp themselves
        if (pushed && index < plan->stamps.size()) snprintf(plan->stamps[index].family, sizeof(plan->stamps[index].family), "%s", last_family());
    }
};

void clear_stamps(pdwt_plan* p) {
    for (auto& s : p->stamps) {
        (void)hipEventDestroy(s.start);
        if (s.stop) (void)hipEventDestroy(s.stop);
    }
    p->stamps.clear();
}

void set_bank(FilterBank& fb, const double* lo, const double* hi, int n) {
    memset(&fb, 0, sizeof(fb));
    for (int i = 0; i < n; i++) {
        fb.lo[i] = (real_t)lo[i];
        fb.hi[i] = (real_t)hi[i];
    }
}

int ensure_tmp(pdwt_plan* p, long long elems) {
    if (p->tmp_elems >= elems) return PDWT_OK;
    if (p->tmp) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipFree(p->tmp));
        p->tmp = nullptr;
        p->tmp_elems = 0;
    }
    HIP_TRY(device_malloc((void**)&p->tmp, (size_t)elems * sizeof(real_t)));
    p->tmp_elems = elems;
    return PDWT_OK;
}

// number of coefficient bands (pdwt/src/common.cu:400-445)
int num_bands(const pdwt_plan* p) { return p->info.ndims == 2 ? 3 * p->info.nlevels + 1 : p->info.nlevels + 1; }

// Builds level dims, band table and arena layout; allocates and zeroes the arena.
int build_layout(pdwt_plan* p) {
    const int L = p->info.nlevels, B = p->batch;
    const bool two_d = p->info.ndims == 2, swt = p->info.do_swt != 0;
    p->lr.assign(L + 1, p->info.Nr);
    p->lc.assign(L + 1, p->info.Nc);
    for (int l = 1; l <= L; l++) {
        p->lr[l] = (two_d && !swt) ? div2(p->lr[l - 1]) : p->lr[l - 1];
        p->lc[l] = swt ? p->lc[l - 1] : div2(p->lc[l - 1]);
    }
    long long off = 0;
    p->bands.clear();
    p->bands.push_back({off, p->lr[L], p->lc[L]});
    off += pad64((long long)B * p->lr[L] * p->lc[L]);
    const int per = two_d ? 3 : 1;
    for (int l = 1; l <= L; l++)
        for (int k = 0; k < per; k++) {
            p->bands.push_back({off, p->lr[l], p->lc[l]});
            off += pad64((long long)B * p->lr[l] * p->lc[l]);
        }
    p->coeff_elems = off;
    p->approx_off.clear();
    if (!swt) {
        p->approx_off.assign(L + 1, -1);
        for (int l = 1; l < L; l++) {
            p->approx_off[l] = off;
            off += pad64((long long)B * p->lr[l] * p->lc[l]);
        }
    } else if (L >= 2) {
        for (int k = 0; k < 2; k++) {
            p->approx_off.push_back(off);
            off += pad64((long long)B * p->info.Nr * p->info.Nc);
        }
    }
    p->image_off = off;
    off += pad64((long long)B * p->info.Nr * p->info.Nc);
    p->arena_elems = off;
    HIP_TRY(pool_alloc(p->device, (void**)&p->arena, (size_t)off * sizeof(real_t), &p->arena_bytes));
    HIP_TRY(hipMemsetAsync(p->arena, 0, (size_t)off * sizeof(real_t), p->stream));
    {
        // the two fp64 results of the norms + the per-block partial sums behind them (launch_norms)
        HIP_TRY(pool_alloc(p->device, (void**)&p->d_red, (size_t)norms_scratch_doubles() * sizeof(double), &p->d_red_bytes));
    }
    return PDWT_OK;
}

// outer products for the built-in non-separable banks.  Band naming follows the separable path
// and pywt: H = high-pass along y (rows index), low-pass along x.  (The reference builds
// LH = outer(lo, hi) with the row index on `lo`, pdwt/src/nonseparable.cu:70-74 "CHECKME",
// which swaps H and V relative to its own separable path; not reproduced.)
int upload_builtin_f2d(pdwt_plan* p) {
    const int n = p->info.hlen;
    std::vector<real_t> h((size_t)8 * n * n);
    const real_t* flo[2] = {p->dec.lo, p->rec.lo};
    const real_t* fhi[2] = {p->dec.hi, p->rec.hi};
    for (int d = 0; d < 2; d++) {
        real_t* LL = h.data() + (size_t)(4 * d + 0) * n * n;
        real_t* LH = h.data() + (size_t)(4 * d + 1) * n * n;  // band H
        real_t* HL = h.data() + (size_t)(4 * d + 2) * n * n;  // band V
        real_t* HH = h.data() + (size_t)(4 * d + 3) * n * n;
        for (int i = 0; i < n; i++)      // i: y tap
            for (int j = 0; j < n; j++) {  // j: x tap
                LL[i * n + j] = flo[d][i] * flo[d][j];
                LH[i * n + j] = fhi[d][i] * flo[d][j];
                HL[i * n + j] = flo[d][i] * fhi[d][j];
                HH[i * n + j] = fhi[d][i] * fhi[d][j];
            }
    }
    if (!p->d_f2d) HIP_TRY(device_malloc((void**)&p->d_f2d, (size_t)8 * kMaxTaps * kMaxTaps * sizeof(real_t)));
    HIP_TRY(hipMemcpyAsync(p->d_f2d, h.data(), h.size() * sizeof(real_t), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return PDWT_OK;
}

void build_schedule(pdwt_plan* p);

int create_impl(const real_t* img, int batch, int Nr, int Nc, const char* wname, int levels, int mem_is_on_host,
                int do_separable, int do_cycle_spinning, int do_swt, int ndim, int device_id, void* stream,
                pdwt_handle* out) {
    if (!out) return fail(PDWT_ERR_ARG, "pdwt_create: out is null");
    *out = nullptr;
    if (Nr < 1 || Nc < 1 || batch < 1) return fail(PDWT_ERR_ARG, "pdwt_create: bad shape (%d, %d, %d)", batch, Nr, Nc);
    if (!wname) return fail(PDWT_ERR_ARG, "pdwt_create: wname is null");
    const WaveletEntry* w = find_wavelet(wname);
    if (!w) return fail(PDWT_ERR_WAVELET, "unknown wavelet name %s", wname);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(PDWT_ERR_HIP, "no HIP device available (this library has no CPU path)");
    if (device_id < 0) HIP_TRY(hipGetDevice(&device_id));
    if (device_id >= ndev) return fail(PDWT_ERR_ARG, "device %d out of range (%d devices)", device_id, ndev);

    pdwt_plan* p = new pdwt_plan();
    p->device = device_id;
    DeviceGuard guard(device_id);
    if (!guard.ok) { delete p; return fail(PDWT_ERR_HIP, "cannot select device %d", device_id); }

    p->batch = batch;
    p->info.Nr = Nr;
    p->info.Nc = Nc;
    p->info.do_swt = do_swt ? 1 : 0;
    p->do_separable = do_separable ? 1 : 0;
    p->do_cycle_spinning = do_cycle_spinning ? 1 : 0;
    p->state = PDWT_INIT;
    snprintf(p->wname, sizeof(p->wname), "%s", wname);

    // dimensions (wt.cu:133-142)
    ndim = ndim < 2 ? 1 : 2;
    if (Nr == 1) ndim = 1;
    p->info.ndims = ndim;
    if (ndim == 1 && !p->do_separable) {
        puts("Warning: 1D DWT was requestred, which is incompatible with non-separable transform.");
        puts("Ignoring the do_separable option.");
        p->do_separable = 1;
    }
    // levels (wt.cu:111-114, 155-165)
    if (levels < 1) {
        puts("Warning: cannot initialize wavelet coefficients with nlevels < 1. Forcing nlevels = 1");
        levels = 1;
    }
    p->info.hlen = w->hlen;
    set_bank(p->dec, w->dec_lo, w->dec_hi, w->hlen);
    set_bank(p->rec, w->rec_lo, w->rec_hi, w->hlen);
    const int N = (ndim == 2) ? (Nr < Nc ? Nr : Nc) : Nc;
    int wmaxlev = ilog2(N / (w->hlen - 1));
    if (wmaxlev < 1) wmaxlev = 1;  // the reference would set 0 levels and index out of bounds
    if (levels > wmaxlev) {
        printf("Warning: required level (%d) is greater than the maximum possible level for %s (%d) on a %dx%d image.\n",
               levels, wname, wmaxlev, Nc, Nr);
        printf("Forcing nlevels = %d\n", wmaxlev);
        levels = wmaxlev;
    }
    p->info.nlevels = levels;
    if (p->do_cycle_spinning && p->info.do_swt)
        puts("Warning: makes little sense to use Cycle spinning with stationary Wavelet transform");
    if (p->do_cycle_spinning && ndim == 1) {  // wt.cu:179-183
        delete p;
        return fail(PDWT_ERR_UNSUPPORTED, "cycle spinning is not implemented for 1D. Use SWT instead.");
    }

    int rc = PDWT_OK;
    auto bail = [&](int code) {
        pdwt_destroy(p);
        return code;
    };
    if (stream) {
        p->stream = (hipStream_t)stream;
        p->own_stream = false;
    } else {
        hipError_t e = pool_stream(p->device, &p->stream);
        if (e != hipSuccess) { delete p; return fail(PDWT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        p->own_stream = true;
    }
    if ((rc = build_layout(p)) != PDWT_OK) return bail(rc);
    p->tune = current_tuning();
    build_schedule(p);
    if (img) {
        hipError_t e = hipMemcpyAsync(p->image(), img, (size_t)batch * Nr * Nc * sizeof(real_t),
                                      mem_is_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e != hipSuccess) {
            fail(PDWT_ERR_HIP, "image upload failed: %s", hipGetErrorString(e));
            return bail(PDWT_ERR_HIP);
        }
    }
    if (!p->do_separable && (rc = upload_builtin_f2d(p)) != PDWT_OK) return bail(rc);
    *out = p;
    return PDWT_OK;
}

// ---------------------------------------------------------------- level loops
//
// The launch list of a plan is decided ONCE (create, clone, set_filters_forward: build_schedule) and
// replayed by forward_impl / inverse_impl: no environment lookups, no allocations per call.  A step is
// one launch: a single level, a two-level tile pyramid (small 2D levels), a two-level streaming strip
// (batches of large images) or a fused run of K 1D levels.  The planner's eligibility tests are the
// launchers' own predicates (dwt2_pyramid_supported, dwt1_fused_supported); should a launcher still
// answer hipErrorNotSupported, the step's levels run through the per-level kernels instead.

constexpr int kMaxFusedLevelsHost = 10;  // == kMaxFusedLevels of dwt1_fused_kernels.hpp
constexpr int kChainMaxLevelsHost = 6;  // == kChainMaxLevels of dwt2_chain_kernels.hpp
constexpr int kTailMaxLevelsHost = 14;  // == kTailMaxLevels of dwt2_tail_kernels.hpp

// Where the approximation of level l lives: l = 0 the image, l = L band 0, otherwise its slot in the
// arena (SWT: two ping-pong planes).  Forward level l reads slot l-1 and writes slot l; inverse level l
// reads slot l and writes slot l-1 (band 0 is never overwritten).
real_t* approx_slot(const pdwt_plan* p, int l) {
    const int L = p->info.nlevels;
    if (l <= 0) return p->image();
    if (l >= L) return p->band(0);
    return p->arena + (p->info.do_swt ? p->approx_off[l & 1] : p->approx_off[l]);
}

void build_schedule(pdwt_plan* p) {
    using pdwt::Step;
    p->sched_fwd.clear();
    p->sched_inv.clear();
    const int L = p->info.nlevels, hlen = p->info.hlen;
    const bool swt = p->info.do_swt != 0;
    if (p->info.ndims == 2) {
        const bool fusable = !swt && p->do_separable;
        // Tile pyramid: SMALL level pairs (at most 2^20 samples enter the pair), where the fixed cost of a
        // launch is what a level costs (launch_dwt2_pyramid.hip has the numbers).  Streaming strips: the
        // opposite regime, >= 2^26 samples enter the pair (a batch of large images): the level is
        // bandwidth-bound and not writing + re-reading A_l pays (forward 284 -> 232 us for 8 x 4096^2,
        // profiles/r01f_kbench_strip.txt); the INVERSE strip kernel is no faster than two launches and only
        // runs under PDWT_FORCE_STRIP=1 (tests), which also drops the size threshold.
        const bool no_pyr = getenv("PDWT_NO_PYRAMID") != nullptr, no_strip = lab_env("PDWT_NO_STRIP") != nullptr;
        const bool force_strip = lab_env("PDWT_FORCE_STRIP") != nullptr;
        auto samples = [&](int l) { return (long long)p->batch * p->lr[l - 1] * p->lc[l - 1]; };
        auto pair_ok = [&](int l, bool inverse = true) { return l + 1 <= L && dwt2_pyramid_supported(hlen, p->lr[l - 1], p->lc[l - 1], inverse); };
        auto strip_at = [&](int l, bool inverse) {
            static const bool inv_strip_l1 = lab_env("PDWT_INV_STRIP") != nullptr;  // A/B: the inverse strips for levels 1+2 of a batch
            if (!fusable || no_strip || !pair_ok(l) || !dwt2_strip_supported(hlen, p->lr[l - 1], p->lc[l - 1]) ||
                (inverse && !force_strip && !(inv_strip_l1 && l == 1)))
                return false;
            // (2-tap filters: no rows carried between chunks, the strips are ahead from 2^25 samples on -- forward of 8 x 2048^2
            // 63 -> 56 us, 32 x 1024^2 60 -> 52, 3 x 4096^2 106 -> 88; longer filters lose there: db4 2 x 4096^2 78 -> 83)
            static const int strip_min = lab_env("PDWT_STRIP_MIN_LOG2") ? atoi(lab_env("PDWT_STRIP_MIN_LOG2")) : 0;  // A/B measurements
            const int min_log2 = strip_min > 0 ? strip_min : (int)(hlen == 2 && sizeof(real_t) == 4 ? tune::strip2_min_log2_haar : tune::strip2_min_log2);
            return force_strip || samples(l) >= (1LL << min_log2);
        };
        static const bool inv_pyr_l1 = lab_env("PDWT_INV_PYR_L1") != nullptr;  // A/B: the tile pyramid for levels 1+2 of any inverse
        auto pyr_at = [&](int l, bool inverse) { return fusable && !no_pyr && pair_ok(l, inverse) && samples(l) <= (1LL << tune::pyr2_max_log2); };
        // Three levels per launch where a small image (at most 2^19 samples, 2^20 over the batch) has three (or five, six, ...) levels
        // left: one launch fewer per direction -- 512^2 db2 L3: 15.6 -> 9.9 us per forward+inverse, 256^2 db4 L5: 26.8 ->
        // 18.4 us, 64 x 128^2 db4 L3: 27.0 -> 20.3 us; at 1024^2 the pairs are ahead (db4 L3: 18.6 against 20.4 us).  Four levels left stay two tile
        // pyramids.  PDWT_NO_PYR3=1 keeps the pairs (A/B measurements).
        const bool no_pyr3 = no_pyr || lab_env("PDWT_NO_PYR3") != nullptr;
        auto pyr3_at = [&](int l, bool inverse) {
            const int left = L - l + 1;
            // (round 4: the FORWARD three-level kernel of 10-16 taps is behind three launches of the small LDS tiles of
            // launch_dwt2_fast.hip -- sym8 256^2: 11.0 against 9-10 us, 16 images of 256^2: 44.8 against 15.6 us; the inverse
            // stays ahead: 5.3 against 12.1 us; profiles/r04r_cliffs_batch_odd.txt)
            if (!inverse && hlen > 8 && lab_env("PDWT_PYR3_FWD_LONG") == nullptr) return false;  // (the knob: tests keep the kernel covered)
            // filters of 10-16 taps recompute a 16x larger halo: ahead up to 512^2 only (sym8 512^2 L3: 21.6 against 25.0 us,
            // 1024 x 512: 37.3 against 25.4 us; profiles/r02y_pyr3_sweep.txt)
            // (round 5: 2- and 4-tap plans with three levels left take the launch up to 2^20 samples -- haar L3 forward+inverse
            // 1024^2 14.2 -> 9.1 us, 1000^2 16.8 -> 12.6, 768 x 1024 14.4 -> 10.3; db2 1000^2 18.8 -> 16.4, 600 x 1000 18.8 -> 14.8, 1024^2
            // 16.0 -> 16.4; with five levels left the pair first stays ahead: 1024^2 L5 haar 15.1 against 16.7, db2 17.1 against 25.6)
            // ... and six or nine levels left likewise (whole groups of three): 1024^2 L6 haar 23.9 -> 15.6 us, db2 27.4 -> 23.5; 2048^2 L7
            // (behind its level-1 launch) haar 36.5 -> 29.3, db2 41.6 -> 37.1
            // 6- and 8-tap plans with six or nine levels left too (with three left their pair + level stays ahead: 1024^2 db4 L3 18.6
            // against 20.4 us): 1024^2 L6 db3 29.7 -> 26.4, db4 31.7 -> 29.0; 2048^2 L7 47.5 -> 42.2, 47.7 -> 43.4
            const bool short_whole = (hlen <= 4 && left % 3 == 0) || (hlen <= 8 && left % 3 == 0 && left >= 6);
            const long long per_image = 1LL << (short_whole ? tune::pyr3_image_short_log2 : (hlen <= 8 ? tune::pyr3_image_log2 : tune::pyr3_image_long_log2));
            return fusable && !no_pyr3 && left >= 3 && left != 4 && samples(l) <= (1LL << tune::pyr3_max_log2) &&
                   (long long)p->lr[l - 1] * p->lc[l - 1] <= per_image && dwt2_pyr3_supported(hlen, p->lr[l - 1], p->lc[l - 1]);
        };
        // Two levels per WAVEFRONT (dwt2_fwd2_wave: A_l stays in registers, overlapping strips).  Correct and
        // tested, but NOT faster than two launches on MI355X (profiles/r02g_wbench_*.txt: 4096^2 31.3 us against
        // 21.9 + 7.9 us; 8 x 4096^2 267 us against 209 + ~50 us, streaming strips 229 us): one wavefront per SIMD is
        // issue-bound, and the second level's shifts, descriptors and 4-B stores add 50 % instructions for 20 %
        // fewer bytes.  Opt-in only: PDWT_WAVE2=1 or pdwt_set_tuning("wave2", 1) (tests keep it covered).
        const bool wave2_on = p->tune.wave2 != 0;
        const int wmin = p->tune.wave_min_log2;
        auto wave2_at = [&](int l, bool inverse) {
            return fusable && !inverse && wave2_on && wmin < 63 && l + 1 <= L && samples(l) >= (1LL << wmin) &&
                   dwt2_wave2_supported(hlen, p->lr[l - 1], p->lc[l - 1]);
        };
        // 2-tap SWT: levels 1-3 and 4-6 in one launch each (swt2_fused_kernels.hpp).  The approximations between
        // groups live in the two ping-pong planes (slot l & 1): a two-level group in the MIDDLE would read and
        // write the same plane, so it is only taken next to the image or to band 0.
        // Only while the transform's planes (3 L + 2 of them) about fit the 256 MiB Infinity Cache: the fused kernels
        // keep 7-10 output (input) streams per wavefront going, 1 KiB per row each, which HBM serves badly once the
        // planes are cold -- two 2048^2 images: 343 us fused against 329 us level by level, four: 746 against 715 us,
        // one: 137 against 183 us (profiles/r02y_bench_cfg4_batch.txt).  Round 4 measured the two directions apart
        // (profiles/r04_ab_batchrange.txt): beyond the cache it is the fused FORWARD that loses (2 x 2048^2 haar L3 98 -> 109 us,
        // db2 102 -> 125) while the fused INVERSE wins by 15-40 % (109 -> 93, db2 158 -> 97; 64 x 512^2 haar 237 -> 174): its
        // input streams are the ones a launch per level reads too, and it writes ONE plane instead of one per level.  fp32
        // plans therefore keep the fused inverse at any size.
        const long long swt_bytes = (long long)sizeof(real_t) * (3 * L + 2) * p->batch * p->info.Nr * p->info.Nc;
        auto swt_group = [&](int l, bool inverse) {
            if (!swt || !p->do_separable || !p->tune.swt_fused) return 0;
            // Round 5: the fused forward of a 2-tap batch beyond the cache runs IMAGE BY IMAGE (forward_impl): one image's 3 K + 1 output
            // streams at a time instead of the whole batch's -- 2 / 4 x 2048^2 haar L3 forward 97.7 / 212.7 us level by level, 108.6 /
            // 217.9 fused over the batch, 92.0 / 194.0 fused image by image; L5, 4 images: 349 / 409 / 319 (tools/swt_batch_probe.py,
            // profiles/r05d_swt_batch_probe.txt).  4-tap pairs stay level by level there (db2: 101 against 115 us).
            // Round 6: on rows that are not whole 16-B groups the level kernels are the ones that lose (4095 x 4093 db2 L4 forward 434 us level
            // by level, 327 fused; 16 x 1000 x 1002: 405 / 307; aligned sizes 296 / 323: profiles/r06_swt4_fwd_ab.txt)
            const bool odd4 = hlen == 4 && sizeof(real_t) == 4 && tune::swt4_fwd_fused_odd && (p->info.Nc % 4) != 0;
            if (swt_bytes > (tune::swt_fused_max_mib << 20) && p->tune.swt_fused < 2 && !(inverse && sizeof(real_t) == 4) && !(hlen == 2 && sizeof(real_t) == 4) && !odd4)
                return 0;  // "swt_fused" = 2 forces both directions (tests)
            for (int K = (L - l + 1 < 3 ? L - l + 1 : 3); K >= 2; --K) {
                const bool same_plane = K == 2 && l - 1 >= 1 && l + K - 1 <= L - 1;
                if (!same_plane && swt2_fused_supported(hlen, p->info.Nr, p->info.Nc, l, K, inverse)) return K;
            }
            return 0;
        };
        // Levels 1..K in ONE launch with in-launch hand-offs between the levels (dwt2_chain_kernels.hpp): the level-per-launch
        // data flow without the launch boundaries.  Correct (tests/test_gpu_chain.py) and OPT-IN: measured on MI355X it is
        // break-even at K = 2 and slower beyond -- 4096^2 db4, rocprofv3: levels 1+2 31.4 us as one launch against 21.6 + 7.7
        // + a 1.8 us boundary, every further level +5 us (a poll round trip before the tile's loads and a store drain behind
        // them cost what a launch boundary costs; the two-level tile pyramid does levels 3+4 in 5.3 us); the inverse of a
        // batch loses 16 % (every level-1 tile polls before it may load).  profiles/r03h_chain_sweep.txt.  "chain" knob
        // (PDWT_CHAIN): 0 never (default), 1 one cache-resident image (2^22 < samples <= 2^24) and the inverse of batches
        // of >= 2^26 samples, 2 wherever the kernel applies (tests), 3 = 2 and the batch forward too.
        const int chain_mode = p->tune.chain;
        auto chain_at = [&](int l, bool inverse) {
            if (!fusable || chain_mode == 0 || l != 1 || L < 2) return 0;
            const long long per_image = (long long)p->lr[0] * p->lc[0];
            const bool one_image = per_image > (1LL << 22) && samples(1) <= (1LL << 24);
            const bool big_batch = samples(1) >= (1LL << 26) && (inverse || chain_mode >= 3);
            if (!(chain_mode >= 2 || one_image || big_batch)) return 0;
            int K = L < kChainMaxLevelsHost ? L : kChainMaxLevelsHost;
            if (const char* e = lab_env("PDWT_CHAIN_K")) K = atoi(e) < K ? atoi(e) : K;  // A/B measurements
            while (K >= 2 && !dwt2_chain_supported(hlen, p->lr[0], p->lc[0], K)) --K;
            return K >= 2 ? K : 0;
        };
        {   // flag memory of the chains: batch x tiles words per direction, zeroed once (epochs only grow)
            const int Kf = chain_at(1, false), Ki = chain_at(1, true);
            const int K = Kf > Ki ? Kf : Ki;
            const long long words = K ? (long long)p->batch * dwt2_chain_tiles(p->lr[0], p->lc[0], K) : 0;
            if (words > p->chain_words) {
                if (p->chain_flags) (void)hipFree(p->chain_flags);
                p->chain_flags = nullptr;
                p->chain_words = 0;
                if (device_malloc((void**)&p->chain_flags, (size_t)(2 * words) * sizeof(unsigned)) == hipSuccess &&
                    hipMemsetAsync(p->chain_flags, 0, (size_t)(2 * words) * sizeof(unsigned), p->stream) == hipSuccess) {
                    p->chain_words = words;
                    p->chain_epoch = 0;
                } else {
                    (void)hipGetLastError();
                    if (p->chain_flags) (void)hipFree(p->chain_flags);
                    p->chain_flags = nullptr;
                }
            }
        }
        // The tail: once one image's approximation is small enough for ONE CU, ALL remaining levels in one launch, one
        // workgroup per image (dwt2_tail_kernels.hpp).  The reference's benchmark and test plans ask for the maximum number
        // of levels: 2048^2 haar L11 ended in PYR3[7-9] LEVEL[10] LEVEL[11].  The launch costs about 2.5 us + 0.5 us per
        // level + the arithmetic of its first levels on one CU (~1 us per tap at 128^2), a pyramid launch on such sizes ~3 us
        // for at most three levels: the tail pays from four or five levels on and only on small planes (samples x taps <=
        // 2^14: 2-tap filters from 64 x 128 on; longer filters run out of levels first).  tools/tailbench.sh, forward /
        // forward+inverse us: haar 128^2 L7 12.3 / 36.9 -> 6.7 / 16.6, 1024^2 L10 15.6 / 44.9 -> 11.4 / 27.8, 2048^2 L11
        // 21.5 / 51.1 -> 17.4 / 40.8, 16 x 128^2 L7 18.9 / 39.8 -> 9.7 / 18.1 (profiles/r04zb_tailbench.txt); wider rules
        // (2^15-2^16, three levels) lose on 4-tap plans (db2 1024^2 L8 11.5 -> 15.7).
        // PDWT_NO_TAIL / PDWT_TAIL_WORK_LOG2 / PDWT_TAIL_MIN_K: A/B measurements.
        // (read per plan, not once per process: the parity tests widen the rule to reach every instantiation)
        const bool no_tail = getenv("PDWT_NO_TAIL") != nullptr;
        const int tail_work_log2 = lab_env("PDWT_TAIL_WORK_LOG2") ? atoi(lab_env("PDWT_TAIL_WORK_LOG2")) : (int)tune::tail_work_log2;
        const int tail_min_k = lab_env("PDWT_TAIL_MIN_K") ? atoi(lab_env("PDWT_TAIL_MIN_K")) : (int)tune::tail_min_levels;
        const long long tail_batch = lab_env("PDWT_TAIL_BATCH") ? atoll(lab_env("PDWT_TAIL_BATCH")) : tune::tail_batch_image;  // largest image taken in batch mode (0 = off)
        auto tail_at = [&](int l) {
            const int K = L - l + 1;
            const long long per_image = (long long)p->lr[l - 1] * p->lc[l - 1];
            // a BATCH of tiny images (<= 64 x 64) is the opposite regime, throughput: thousands of independent workgroups, four per
            // CU, each transforming its image out of LDS with no halo and no padding, where the tile kernels launch eight mostly
            // empty workgroups per image and level -- 4096 images of 64^2, db4 L3: forward 196 us on the level kernels
            // (B x n^2, forward+inverse us: 4096 x 64^2 db4 L3 313 -> 121, 4096 x 32^2 db2 L3 453 -> 38, 1024 x 64^2 sym8 L2 198 -> 54;
            // 128 x 128 images only with five levels and more: 256 x 128^2 db2 L5 43.5 -> 29.5, but 1024 x 128^2 db4 L3 102 -> 136;
            // profiles/r04zc_small_batches.txt)
            // Any sizes (powers of two: the mask / shift kernels; the rest, odd sizes included: the general ones, sizes by ceil-halving
            // like the level kernels).  20000 x 28^2 db2 L2: 1286 us forward+inverse on the level kernels, ten times its power-of-two
            // neighbour 16384 x 32^2; with a third level (7 x 7 out of 14 x 14) the level kernels spent 700 of 920 us on that level alone.
            // Images of 4097 .. 16384 samples (one 1024-thread workgroup per CU): with five levels and more, or with three and more when
            // the batch is at most a round and a half of the chip's 256 CUs -- 256 x 128^2 db2 L3 42.4 -> 25.0 us, 120 x 100^2 db4 L3
            // 43.2 -> 33.7, 300 x 80x120 db2 L4 74.1 -> 40.9; larger batches keep the level kernels (1024 x 128^2 db4 L3 95 -> 118,
            // 4096 x 128^2 db2 L4 278 -> 361; profiles/r04zs_tail_mid_images.txt)
            const bool tiny = per_image <= tail_batch && per_image * hlen <= 16 * tail_batch;
            const bool deep = per_image <= 4 * tail_batch && K >= 5 && per_image * hlen <= 16 * tail_batch;
            // (beyond 2^20 samples only: up to there the three-level tile pyramid takes such batches and is ahead -- 64 x 128^2 db4 L3
            // 20.9 us against 27.4, haar 9.5 against 17.5)
            const bool few = per_image <= 4 * tail_batch && K >= 3 && p->batch <= tune::tail_few_batch && per_image * hlen <= 32 * tail_batch &&
                             samples(l) > (1LL << tune::tail_max_log2);
            if (fusable && !no_tail && (tiny || deep || few) && samples(l) >= (1LL << tune::tail_max_log2)) {
                const int Kb = dwt2_tail_max_levels(hlen, p->lr[l - 1], p->lc[l - 1], K);
                if (Kb >= 1 && (tiny || (deep && Kb >= 5) || (few && Kb >= 3))) return Kb;
            }
            if (!fusable || no_tail || samples(l) > (1LL << tune::tail_max_log2) || per_image * hlen > (1LL << tail_work_log2)) return 0;
            if (K < tail_min_k && !(K >= tail_min_k - 1 && per_image <= 1024)) return 0;
            return dwt2_tail_supported(hlen, p->lr[l - 1], p->lc[l - 1], K) ? K : 0;
        };
        // The undecimated twin (swt2_tail_kernels.hpp), batch mode only: the whole L-level SWT of every image of a large batch of
        // tiny images (at most 4096 samples, any sizes) in one launch per direction -- 4096 images of 64^2, db4 L2
        // forward+inverse 1006 us through the level kernels, 16384 of 32^2 haar L3 1177 us (profiles/r04zm_swt_tiny_batches.txt)
        auto swt_tail = [&]() {
            const long long per_image = (long long)p->info.Nr * p->info.Nc;
            // (64 x 64 images -- four 16 KiB planes, two workgroups per CU -- only from 8 taps on, where the level kernels' inverse reads
            // four bands x hlen taps per output from global memory: db4 L2 1015 -> 739 us, but haar L3 458 -> 482; 32 x 32 and below
            // always: haar L3 16384 x 32^2 1176 -> 352, 65536 x 16^2 4520 -> 674, db4 L2 8192 x 32^2 1047 -> 202)
            const long long total = (long long)p->batch * per_image;
            static const int few = lab_env("PDWT_SWT_TAIL_FEW") ? atoi(lab_env("PDWT_SWT_TAIL_FEW")) : (int)tune::swt_tail_few;  // A/B measurements
            // ... and the latency regime: a FEW tiny images with two levels and more -- one launch instead of one per level
            // (32 x 32 and below: 16 x 32^2 haar L5 forward+inverse 47 -> 32 us, 8 x 16^2 db2 L2 30 -> 11; one workgroup of 256 threads is
            // too slow for a single 64 x 64 image: haar L6 50 -> 105 us)
            const bool latency = p->batch <= few && L >= 2 && per_image <= 1024;
            if (!latency && per_image > 1024 && hlen < 8) return false;
            return swt && p->do_separable && !no_tail && tail_batch > 0 && per_image <= tune::swt_tail_image && per_image * hlen <= (1LL << 17) &&
                   (total >= (1LL << 20) || latency) && swt2_tail_supported(hlen, p->info.Nr, p->info.Nc, L);
        };
        for (int dir = 0; dir < 2; dir++) {
            std::vector<Step>& out = dir ? p->sched_inv : p->sched_fwd;
            for (int l = 1; l <= L; l++) {
                if (const int K = tail_at(l)) { out.push_back({Step::TAIL, l, K}); l += K - 1; continue; }
                if (l == 1 && swt_tail()) { out.push_back({Step::TAIL, 1, L}); break; }
                if (const int K = swt_group(l, dir != 0)) { out.push_back({Step::SWTF, l, K}); l += K - 1; continue; }
                if (const int K = p->chain_flags ? chain_at(l, dir != 0) : 0) { out.push_back({Step::CHAIN, l, K}); l += K - 1; continue; }
                if (strip_at(l, dir != 0)) { out.push_back({Step::STRIP2, l, 2}); l++; }
                else if (wave2_at(l, dir != 0)) { out.push_back({Step::WAVE2, l, 2}); l++; }
                else if (pyr3_at(l, dir != 0)) { out.push_back({Step::PYR3, l, 3}); l += 2; }
                else if ((pyr_at(l, dir != 0) && !strip_at(l + 1, dir != 0)) || (inv_pyr_l1 && dir == 1 && l == 1 && fusable && pair_ok(l))) { out.push_back({Step::PYR2, l, 2}); l++; }
                else out.push_back({Step::LEVEL, l, 1});
            }
        }
        // the inverse undoes the deepest level first
        std::vector<Step> rev(p->sched_inv.rbegin(), p->sched_inv.rend());
        p->sched_inv.swap(rev);
    } else {
        // 1D: runs of K >= 2 levels in ONE fused launch (dwt1_fused_kernels.hpp); the rest level by level
        const bool fuse = !swt && getenv("PDWT_NO_FUSED_1D") == nullptr;
        const int cap = dwt1_fused_max_levels(hlen);
        const int reg = swt ? 0 : p->tune.reg1d;  // bit 0: forward, bit 1: inverse
        for (int dir = 0; dir < 2; dir++) {
            std::vector<Step>& out = dir ? p->sched_inv : p->sched_fwd;
            int l = 0;
            while (l < L) {
                // three levels at a time in registers (dwt1_reg_kernels.hpp) where the rows qualify ...
                // (the FORWARD only while the rows are cache-sized: for a batch of long rows beyond 2^25 samples the LDS
                // pyramid, which reads and writes every sample once in ONE launch, is ahead -- 16 x 2^24: 462 us against
                // 513 + 71 us -- while the inverse register kernels stay ahead, 440 + 53 against 607 us;
                // profiles/r02y_bench_cfg3_batch.txt.  Bit 2 of the knob forces the forward too.)
                // (The fp64 build has no LDS pyramid to prefer.)
                bool reg_here = ((reg >> dir) & 1) && (dir == 1 || ((reg >> 2) & 1) || sizeof(real_t) == 8 ||
                                                       (long long)p->batch * p->info.Nr * p->info.Nc <= (1LL << tune::reg1d_fwd_max_log2));
                // Round 4 (tools/reg1d_rows.sh, profiles/r04v_reg1d_rows.txt): the register kernels' overlapping 1024-sample
                // blocks and their launch per three levels only pay on LONG rows of a LARGE transform -- rows of 4096 samples
                // lose a quarter of every row's blocks to the overlap (4096 x 4096 sym8 L5 forward+inverse 86.6 us against 83.6
                // for the LDS pyramids, db4 79.6 against 70.6), and below ~2^23 samples one launch of the pyramid beats two
                // (2^20 samples: db4 16.2 against 14.2 us, 256 rows of 4096: 18.0 against 13.8).  fp32 only (the fp64 build has
                // no LDS pyramid); bit 3 of the "reg1d" knob lifts the limits (tests).
                static const int min_log2 = lab_env("PDWT_REG1D_MIN_LOG2") ? atoi(lab_env("PDWT_REG1D_MIN_LOG2")) : (int)tune::reg1d_min_log2;
                static const int min_row = lab_env("PDWT_REG1D_MIN_ROW") ? atoi(lab_env("PDWT_REG1D_MIN_ROW")) : (int)tune::reg1d_min_row;
                // Decided for the PLAN, not per launch: a register first stage followed by a pyramid on the remaining 2^21
                // samples measured slower than either pure schedule (2^24 samples, db4 L5: 64.6 us against 54.5 / 60.3).
                if (reg_here && sizeof(real_t) == 4 && !((reg >> 3) & 1)) {
                    const long long total = (long long)p->batch * p->info.Nr * p->info.Nc;
                    if (total < (1LL << min_log2) || p->lc[0] < min_row) reg_here = false;
                }
                // A/B (PDWT_REG1D_THEN_FUSED=1): only the FIRST group in registers, the remainder through the LDS pyramid -- slower when the
                // pyramid's forward segment shrank with the number of levels (64.6 us), re-measured after that was fixed in round 4
                static const bool then_fused = lab_env("PDWT_REG1D_THEN_FUSED") != nullptr;
                if (then_fused && l > 0) reg_here = false;
                int K = reg_here ? (L - l < 3 ? L - l : 3) : 0;
                while (K >= 1 && !dwt1_reg_supported(hlen, p->lc[l], K)) --K;
                if (K >= 1) {
                    out.push_back({Step::REG1D, l + 1, K});
                    l += K;
                    continue;
                }
                // ... else runs of K >= 2 levels out of LDS (dwt1_fused_kernels.hpp), else level by level
                K = fuse ? (L - l < cap ? L - l : cap) : 1;
                while (K >= 2 && !dwt1_fused_supported(hlen, p->lc[l], K, false)) --K;
                // (a single level too where the several-rows-per-wavefront kernels take the batch: 131072 rows of 32 samples, db10 L1
                // forward+inverse 209 us on the level kernels)
                const bool rows1 = K < 2 && fuse && sizeof(real_t) == 4 && dwt1_fused_supported(hlen, p->lc[l], 1, true) &&
                                   dwt1_rows_tail_applies(p->batch * p->info.Nr, p->lc[l], 1, hlen);
                if (K >= 2 || rows1) {
                    if (K < 2) K = 1;
                    out.push_back({Step::FUSED1D, l + 1, K});
                    l += K;
                } else {
                    out.push_back({Step::LEVEL, l + 1, 1});
                    ++l;
                }
            }
        }
        std::vector<Step> rev(p->sched_inv.rbegin(), p->sched_inv.rend());
        p->sched_inv.swap(rev);
    }
}

// ---- one 2D level (any kind of plan)
int fwd_level_2d(pdwt_plan* p, int l, bool run) {
    const int B = p->batch, hlen = p->info.hlen;
    const bool swt = p->info.do_swt != 0;
    const real_t* src = approx_slot(p, l - 1);
    real_t* dstA = approx_slot(p, l);
    real_t* H = p->band(3 * (l - 1) + 1);
    real_t* V = p->band(3 * (l - 1) + 2);
    real_t* D = p->band(3 * (l - 1) + 3);
    if (!p->do_separable) {
        NonsepArgs a;
        a.in = src; a.A = dstA; a.H = H; a.V = V; a.D = D; a.out = nullptr;
        a.filt = p->d_f2d;  // forward banks
        a.Nr = p->lr[l - 1]; a.Nc = p->lc[l - 1]; a.Nrc = p->lr[l]; a.Ncc = p->lc[l];
        a.f = 1 << (l - 1); a.do_swt = swt ? 1 : 0;
        a.img_bstride = (long long)a.Nr * a.Nc; a.coef_bstride = (long long)a.Nrc * a.Ncc;
        a.hlen = hlen;
        Stamp st(p, "nonsep_fwd_level");
        if (run) HIP_TRY(launch_nonsep_fwd(a, B, p->stream));
    } else if (!swt) {
        Fwd2DArgs a;
        a.in = src; a.A = dstA; a.H = H; a.V = V; a.D = D;
        a.Nr = p->lr[l - 1]; a.Nc = p->lc[l - 1]; a.Nr2 = p->lr[l]; a.Nc2 = p->lc[l];
        a.in_bstride = (long long)a.Nr * a.Nc;
        a.out_bstride = (long long)a.Nr2 * a.Nc2;
        a.hlen = hlen;
        a.fb = p->dec;
        // long filters: a row launch + a column launch through scratch (dwt2_split_kernels.hpp); a decline falls through
        bool done = false;
        if (dwt2_split_supported(hlen, a.Nr, a.Nc, false, (long long)B * a.Nr * a.Nc) &&
            ensure_tmp(p, (long long)B * a.Nr * a.Nc) == PDWT_OK) {
            Stamp st(p, "dwt2_fwd_split");
            const hipError_t e = run ? launch_dwt2_split_fwd(a, p->tmp, B, p->stream) : hipSuccess;
            if (e == hipErrorNotSupported) {
                st.drop();
            } else {
                HIP_TRY(e);
                done = true;
            }
        }
        if (!done) {
            Stamp st(p, "dwt2_fwd_level");
            if (run) HIP_TRY(launch_dwt2_fwd(a, B, p->stream));
        }
    } else {
        const int f = 1 << (l - 1);
        const int Nr = p->info.Nr, Nc = p->info.Nc;
        Swt2DArgs a;
        a.in = src; a.A = dstA; a.H = H; a.V = V; a.D = D; a.out = nullptr;
        a.Nr = Nr; a.Nc = Nc; a.f = f;
        a.bstride = (long long)Nr * Nc;
        a.hlen = hlen;
        a.soft_beta = 0.f;
        a.fb = p->dec;
        if (swt2_fwd_stream_takes(a, B)) {  // row and column pass in one launch, streamed down strips (swt_fwdstream_kernels.hpp)
            Stamp st(p, "swt2_fwd_stream");
            if (run) HIP_TRY(try_launch_swt2_fwd_stream(a, B, p->stream));
            return PDWT_OK;
        }
        bool split = swt2_split_supported(hlen, Nr, Nc, f, false, (long long)B * Nr * Nc) && ensure_tmp(p, 2LL * Nr * Nc * B) == PDWT_OK;
        if (split) {
            // the launcher reads the (process-wide, atomic) threshold again and checks alignments the predicate does not:
            // hipErrorNotSupported is a DECLINE (another thread may have moved the knob in between), not a failure
            Stamp st(p, "swt2_fwd_split");
            const hipError_t e = run ? launch_swt2_split(a, p->tmp, false, B, p->stream) : hipSuccess;
            if (e == hipErrorNotSupported) {
                split = false;
                st.drop();
            } else if (e != hipSuccess) {
                HIP_TRY(e);
            }
        }
        if (!split) {
            // (any Nr: where the dilation does not divide the row count the tiles wrap ROWS, not phase indices -- round 5; until then such
            // levels ran as three one-sample-per-thread passes through scratch, 2047^2 at twice the time of 2048^2)
            Stamp st(p, "swt2_fwd_level");
            if (run) HIP_TRY(launch_swt2_fwd(a, B, p->stream));
        }
    }
    return PDWT_OK;
}

// deferred soft_threshold: beta of level l's details (/ sqrt(2)^l when normalised)
real_t pending_beta_of_level(const pdwt_plan* p, int l) {
    real_t b = p->pend_beta;
    if (p->pend_normalize > 0)
        for (int i = 0; i < l; i++) b = (real_t)(b / 1.4142135623730951);
    return b;
}

int inv_level_2d(pdwt_plan* p, int l, bool run) {
    const int B = p->batch, hlen = p->info.hlen;
    const bool swt = p->info.do_swt != 0;
    const real_t* cur = approx_slot(p, l);
    real_t* dst = approx_slot(p, l - 1);
    const real_t* H = p->band(3 * (l - 1) + 1);
    const real_t* V = p->band(3 * (l - 1) + 2);
    const real_t* D = p->band(3 * (l - 1) + 3);
    if (!p->do_separable) {
        NonsepArgs a;
        a.in = nullptr;
        a.A = const_cast<real_t*>(cur); a.H = const_cast<real_t*>(H);
        a.V = const_cast<real_t*>(V); a.D = const_cast<real_t*>(D);
        a.out = dst;
        a.filt = p->d_f2d + (size_t)4 * hlen * hlen;  // inverse banks
        a.Nr = p->lr[l - 1]; a.Nc = p->lc[l - 1]; a.Nrc = p->lr[l]; a.Ncc = p->lc[l];
        a.f = 1 << (l - 1); a.do_swt = swt ? 1 : 0;
        a.img_bstride = (long long)a.Nr * a.Nc; a.coef_bstride = (long long)a.Nrc * a.Ncc;
        a.hlen = hlen;
        Stamp st(p, "nonsep_inv_level");
        if (run) HIP_TRY(launch_nonsep_inv(a, B, p->stream));
    } else if (!swt) {
        Inv2DArgs a;
        a.A = cur; a.H = H; a.V = V; a.D = D; a.out = dst;
        a.Nrc = p->lr[l]; a.Ncc = p->lc[l]; a.Nr = p->lr[l - 1]; a.Nc = p->lc[l - 1];
        a.in_bstride = (long long)a.Nrc * a.Ncc;
        a.out_bstride = (long long)a.Nr * a.Nc;
        a.hlen = hlen;
        a.fb = p->rec;
        bool done = false;
        if (dwt2_split_supported(hlen, a.Nr, a.Nc, true, (long long)B * a.Nr * a.Nc) &&
            ensure_tmp(p, (long long)B * a.Nr * a.Nc) == PDWT_OK) {
            Stamp st(p, "dwt2_inv_split");
            const hipError_t e = run ? launch_dwt2_split_inv(a, p->tmp, B, p->stream) : hipSuccess;
            if (e == hipErrorNotSupported) {
                st.drop();
            } else {
                HIP_TRY(e);
                done = true;
            }
        }
        if (!done) {
            Stamp st(p, "dwt2_inv_level");
            if (run) HIP_TRY(launch_dwt2_inv(a, B, p->stream));
        }
    } else {
        const int f = 1 << (l - 1);
        const int Nr = p->info.Nr, Nc = p->info.Nc;
        Swt2DArgs a;
        a.in = nullptr;
        a.A = const_cast<real_t*>(cur); a.H = const_cast<real_t*>(H);
        a.V = const_cast<real_t*>(V); a.D = const_cast<real_t*>(D);
        a.out = dst;
        a.Nr = Nr; a.Nc = Nc; a.f = f;
        a.bstride = (long long)Nr * Nc;
        a.hlen = hlen;
        a.soft_beta = 0.f;
        if (p->pend_soft) a.soft_beta = pending_beta_of_level(p, l);
        a.fb = p->rec;
        if (swt2_inv_stream_takes(a, B)) {  // row and column synthesis in one launch, streamed down strips (swt_invstream_kernels.hpp)
            Stamp st(p, p->pend_soft ? "swt2_inv_stream+soft" : "swt2_inv_stream");
            if (run) HIP_TRY(try_launch_swt2_inv_stream(a, B, p->stream));
            return PDWT_OK;
        }
        bool split = swt2_split_supported(hlen, Nr, Nc, f, true, (long long)B * Nr * Nc) && ensure_tmp(p, 2LL * Nr * Nc * B) == PDWT_OK;
        if (split) {  // a decline (hipErrorNotSupported) falls through to the other kernels, see fwd_level_2d
            Stamp st(p, p->pend_soft ? "swt2_inv_split+soft" : "swt2_inv_split");
            const hipError_t e = run ? launch_swt2_split(a, p->tmp, true, B, p->stream) : hipSuccess;
            if (e == hipErrorNotSupported) {
                split = false;
                st.drop();
            } else if (e != hipSuccess) {
                HIP_TRY(e);
            }
        }
        if (!split) {
            Stamp st(p, p->pend_soft ? "swt2_inv_level+soft" : "swt2_inv_level");
            if (run) HIP_TRY(launch_swt2_inv(a, B, p->stream));
        }
    }
    return PDWT_OK;
}

// ---- one 1D level
int fwd_level_1d(pdwt_plan* p, int l, bool run) {
    const int rows = p->batch * p->info.Nr, hlen = p->info.hlen;
    const real_t* src = approx_slot(p, l - 1);
    real_t* dstA = approx_slot(p, l);
    real_t* Dl = p->band(l);
    if (!p->info.do_swt) {
        Fwd1DArgs a;
        a.in = src; a.L = dstA; a.H = Dl;
        a.rows = rows; a.Nc = p->lc[l - 1]; a.Nc2 = p->lc[l];
        a.hlen = hlen;
        a.fb = p->dec;
        Stamp st(p, "dwt1_fwd_level");
        if (run) HIP_TRY(launch_dwt1_fwd(a, p->stream));
    } else {
        SwtPassArgs r;
        r.in0 = src; r.in1 = nullptr; r.out0 = dstA; r.out1 = Dl;
        r.Nr = rows; r.Nc = p->info.Nc; r.f = 1 << (l - 1); r.along_y = 0; r.hlen = hlen; r.fb = p->dec;
        Stamp st(p, "swt1_fwd_level");
        if (run) HIP_TRY(launch_swt_pass_fwd(r, p->stream));
    }
    return PDWT_OK;
}

int inv_level_1d(pdwt_plan* p, int l, bool run) {
    const int rows = p->batch * p->info.Nr, hlen = p->info.hlen;
    const real_t* cur = approx_slot(p, l);
    real_t* dst = approx_slot(p, l - 1);
    const real_t* Dl = p->band(l);
    if (!p->info.do_swt) {
        Inv1DArgs a;
        a.L = cur; a.H = Dl; a.out = dst;
        a.rows = rows; a.Ncc = p->lc[l]; a.Nc = p->lc[l - 1];
        a.hlen = hlen;
        a.fb = p->rec;
        Stamp st(p, "dwt1_inv_level");
        if (run) HIP_TRY(launch_dwt1_inv(a, p->stream));
    } else {
        SwtPassArgs r;
        r.in0 = cur; r.in1 = Dl; r.out0 = dst; r.out1 = nullptr;
        r.Nr = rows; r.Nc = p->info.Nc; r.f = 1 << (l - 1); r.along_y = 0; r.hlen = hlen; r.fb = p->rec;
        Stamp st(p, "swt1_inv_level");
        if (run) HIP_TRY(launch_swt_pass_inv(r, p->stream));
    }
    return PDWT_OK;
}

// only == 0: every step; only == l: just the launch whose first (finest) level is l (pdwt_time_level)
int forward_impl(pdwt_plan* p, int only = 0) {
    using pdwt::Step;
    const ActiveTuning tuning_guard(&p->tune);
    const int L = p->info.nlevels, B = p->batch, hlen = p->info.hlen;
    const bool swt = p->info.do_swt != 0;
    const bool two_d = p->info.ndims == 2;
    for (const Step& s : p->sched_fwd) {
        const int l = s.level;
        const bool run = (only == 0 || only == l);
        hipError_t e = hipErrorNotSupported;
        if (s.kind == Step::STRIP2 || s.kind == Step::PYR2 || s.kind == Step::WAVE2) {
            // levels l and l+1 in one launch; A_l never reaches HBM
            real_t* det1[3] = {p->band(3 * (l - 1) + 1), p->band(3 * (l - 1) + 2), p->band(3 * (l - 1) + 3)};
            real_t* band2[4] = {approx_slot(p, l + 1), p->band(3 * l + 1), p->band(3 * l + 2), p->band(3 * l + 3)};
            Stamp st(p, s.kind == Step::STRIP2 ? "dwt2_fwd_strip2" : (s.kind == Step::WAVE2 ? "dwt2_fwd_wave2" : "dwt2_fwd_pyr2"), &e);
            if (!run) continue;
            const real_t* src = approx_slot(p, l - 1);
            const int r0 = p->lr[l - 1], c0 = p->lc[l - 1];
            e = s.kind == Step::STRIP2  ? launch_dwt2_fwd_strip2(src, det1, band2, r0, c0, hlen, p->dec, B, p->stream)
                : s.kind == Step::WAVE2 ? launch_dwt2_fwd_wave2(src, det1, band2, r0, c0, hlen, p->dec, B, p->stream)
                                        : launch_dwt2_fwd_pyr2(src, det1, band2, r0, c0, hlen, p->dec, B, p->stream);
        } else if (s.kind == Step::CHAIN) {
            real_t* det[3 * kChainMaxLevelsHost] = {};
            real_t* app[kChainMaxLevelsHost] = {};
            for (int k = 0; k < s.K; k++) {
                app[k] = approx_slot(p, l + k);
                for (int b = 0; b < 3; b++) det[3 * k + b] = p->band(3 * (l + k - 1) + 1 + b);
            }
            Stamp st(p, "dwt2_fwd_chain", &e);
            if (!run) continue;
            e = launch_dwt2_fwd_chain(approx_slot(p, l - 1), det, app, p->lr[l - 1], p->lc[l - 1], s.K, hlen, p->dec, B,
                                      p->chain_flags, ++p->chain_epoch, p->stream);
        } else if (s.kind == Step::TAIL && swt) {
            real_t* det[3 * kTailMaxLevelsHost] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(1 + k);
            Stamp st(p, "swt2_fwd_tail", &e);
            if (!run) continue;
            e = launch_swt2_tail(approx_slot(p, 0), det, approx_slot(p, s.K), p->info.Nr, p->info.Nc, s.K, hlen, false, p->dec, nullptr, B,
                                 p->stream);
        } else if (s.kind == Step::TAIL) {
            real_t* det[3 * kTailMaxLevelsHost] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            Stamp st(p, "dwt2_fwd_tail", &e);
            if (!run) continue;
            e = launch_dwt2_tail(approx_slot(p, l - 1), det, approx_slot(p, l + s.K - 1), p->lr[l - 1], p->lc[l - 1], s.K, hlen, false,
                                 p->dec, B, p->stream);
        } else if (s.kind == Step::PYR3) {
            real_t* det[9];
            for (int k = 0; k < 9; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            Stamp st(p, "dwt2_fwd_pyr3", &e);
            if (!run) continue;
            e = launch_dwt2_fwd_pyr3(approx_slot(p, l - 1), det, approx_slot(p, l + 2), p->lr[l - 1], p->lc[l - 1], hlen, p->dec, B,
                                     p->stream);
        } else if (s.kind == Step::SWTF) {
            real_t* det[9] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            Stamp st(p, "swt2_fwd_fused", &e);
            if (!run) continue;
            const long long plane = (long long)p->info.Nr * p->info.Nc;
            const bool beyond_cache = (long long)sizeof(real_t) * (3 * L + 2) * B * plane > (320LL << 20);
            if (B > 1 && beyond_cache && p->tune.swt_fused < 2) {
                // image by image (build_schedule: swt_group): the same kernel, one image's output streams at a time
                for (int b = 0; b < B; b++) {
                    real_t* db[9] = {};
                    for (int k = 0; k < 3 * s.K; k++) db[k] = det[k] + b * plane;
                    e = launch_swt2_fused(approx_slot(p, l - 1) + b * plane, approx_slot(p, l + s.K - 1) + b * plane, db, p->info.Nr, p->info.Nc, l,
                                          s.K, false, p->info.hlen, p->dec, nullptr, 1, p->stream);
                    if (e != hipSuccess) break;
                }
            } else {
                e = launch_swt2_fused(approx_slot(p, l - 1), approx_slot(p, l + s.K - 1), det, p->info.Nr, p->info.Nc, l, s.K, false,
                                      p->info.hlen, p->dec, nullptr, B, p->stream);
            }
        } else if (s.kind == Step::REG1D) {
            real_t* det[kMaxFusedLevelsHost] = {};
            for (int k = 0; k < s.K; k++) det[k] = p->band(l + k);
            Stamp st(p, "dwt1_fwd_reg", &e);
            if (!run) continue;
            e = launch_dwt1_fwd_reg(approx_slot(p, l - 1), det, approx_slot(p, l + s.K - 1), B * p->info.Nr, p->lc[l - 1],
                                    s.K, hlen, p->dec, p->stream);
        } else if (s.kind == Step::FUSED1D) {
            real_t* det[kMaxFusedLevelsHost] = {};
            for (int k = 0; k < s.K; k++) det[k] = p->band(l + k);
            Stamp st(p, "dwt1_fwd_fused", &e);
            if (!run) continue;
            e = launch_dwt1_fwd_fused(approx_slot(p, l - 1), det, approx_slot(p, l + s.K - 1), B * p->info.Nr, p->lc[l - 1],
                                      s.K, hlen, p->dec, p->stream);
        }
        if (e == hipSuccess) continue;
        if (e != hipErrorNotSupported) HIP_TRY(e);
        // a single level, or a fused step its launcher declined: level by level
        for (int k = 0; k < s.K && l + k <= L; k++) {
            const int rc = two_d ? fwd_level_2d(p, l + k, run) : fwd_level_1d(p, l + k, run);
            if (rc != PDWT_OK) return rc;
        }
    }
    return PDWT_OK;
}

int inverse_impl(pdwt_plan* p, int only = 0) {
    using pdwt::Step;
    const ActiveTuning tuning_guard(&p->tune);
    const int B = p->batch, hlen = p->info.hlen;
    const bool two_d = p->info.ndims == 2;
    const bool swt = p->info.do_swt != 0;
    for (const Step& s : p->sched_inv) {
        const int l = s.level;  // the step undoes levels l+K-1 .. l and writes approximation slot l-1
        const bool run = (only == 0 || only == l);
        hipError_t e = hipErrorNotSupported;
        if (s.kind == Step::STRIP2 || s.kind == Step::PYR2) {
            const real_t* band2[4] = {approx_slot(p, l + 1), p->band(3 * l + 1), p->band(3 * l + 2), p->band(3 * l + 3)};
            const real_t* det1[3] = {p->band(3 * (l - 1) + 1), p->band(3 * (l - 1) + 2), p->band(3 * (l - 1) + 3)};
            Stamp st(p, s.kind == Step::STRIP2 ? "dwt2_inv_strip2" : "dwt2_inv_pyr2", &e);
            if (!run) continue;
            e = s.kind == Step::STRIP2
                    ? launch_dwt2_inv_strip2(band2, det1, approx_slot(p, l - 1), p->lr[l - 1], p->lc[l - 1], hlen, p->rec, B, p->stream)
                    : launch_dwt2_inv_pyr2(band2, det1, approx_slot(p, l - 1), p->lr[l - 1], p->lc[l - 1], hlen, p->rec, B, p->stream);
        } else if (s.kind == Step::CHAIN) {
            real_t* det[3 * kChainMaxLevelsHost] = {};
            real_t* app[kChainMaxLevelsHost] = {};
            for (int k = 0; k < s.K; k++) {
                app[k] = approx_slot(p, l + k);
                for (int b = 0; b < 3; b++) det[3 * k + b] = p->band(3 * (l + k - 1) + 1 + b);
            }
            Stamp st(p, "dwt2_inv_chain", &e);
            if (!run) continue;
            e = launch_dwt2_inv_chain(approx_slot(p, l - 1), det, app, p->lr[l - 1], p->lc[l - 1], s.K, hlen, p->rec, B,
                                      p->chain_flags + p->chain_words, ++p->chain_epoch, p->stream);
        } else if (s.kind == Step::TAIL && swt) {
            real_t* det[3 * kTailMaxLevelsHost] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(1 + k);
            real_t beta[kTailMaxLevelsHost] = {};
            if (p->pend_soft)  // deferred soft_threshold, applied as the details are staged (see inv_level_2d)
                for (int k = 0; k < s.K; k++) beta[k] = pending_beta_of_level(p, 1 + k);
            Stamp st(p, p->pend_soft ? "swt2_inv_tail+soft" : "swt2_inv_tail", &e);
            if (!run) continue;
            e = launch_swt2_tail(approx_slot(p, s.K), det, approx_slot(p, 0), p->info.Nr, p->info.Nc, s.K, hlen, true, p->rec, beta, B,
                                 p->stream);
        } else if (s.kind == Step::TAIL) {
            real_t* det[3 * kTailMaxLevelsHost] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            Stamp st(p, "dwt2_inv_tail", &e);
            if (!run) continue;
            e = launch_dwt2_tail(approx_slot(p, l + s.K - 1), det, approx_slot(p, l - 1), p->lr[l - 1], p->lc[l - 1], s.K, hlen, true,
                                 p->rec, B, p->stream);
        } else if (s.kind == Step::PYR3) {
            real_t* det[9];
            for (int k = 0; k < 9; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            Stamp st(p, "dwt2_inv_pyr3", &e);
            if (!run) continue;
            e = launch_dwt2_inv_pyr3(approx_slot(p, l + 2), det, approx_slot(p, l - 1), p->lr[l - 1], p->lc[l - 1], hlen, p->rec, B,
                                     p->stream);
        } else if (s.kind == Step::SWTF) {
            real_t* det[9] = {};
            for (int k = 0; k < 3 * s.K; k++) det[k] = p->band(3 * (l - 1) + 1 + k);
            real_t beta[3] = {0, 0, 0};
            if (p->pend_soft)  // deferred soft_threshold, applied as the details are loaded (see inv_level_2d)
                for (int k = 0; k < s.K; k++) beta[k] = pending_beta_of_level(p, l + k);
            Stamp st(p, p->pend_soft ? "swt2_inv_fused+soft" : "swt2_inv_fused", &e);
            if (!run) continue;
            e = launch_swt2_fused(approx_slot(p, l + s.K - 1), approx_slot(p, l - 1), det, p->info.Nr, p->info.Nc, l, s.K, true,
                                  p->info.hlen, p->rec, beta, B, p->stream);
        } else if (s.kind == Step::REG1D) {
            const real_t* det[kMaxFusedLevelsHost] = {};
            for (int k = 0; k < s.K; k++) det[k] = p->band(l + k);
            Stamp st(p, "dwt1_inv_reg", &e);
            if (!run) continue;
            e = launch_dwt1_inv_reg(approx_slot(p, l + s.K - 1), det, approx_slot(p, l - 1), B * p->info.Nr, p->lc[l - 1], s.K, hlen,
                                    p->rec, p->stream);
        } else if (s.kind == Step::FUSED1D) {
            const real_t* det[kMaxFusedLevelsHost] = {};
            for (int k = 0; k < s.K; k++) det[k] = p->band(l + k);
            Stamp st(p, "dwt1_inv_fused", &e);
            if (!run) continue;
            e = launch_dwt1_inv_fused(approx_slot(p, l + s.K - 1), det, approx_slot(p, l - 1), B * p->info.Nr, p->lc[l - 1], s.K, hlen,
                                      p->rec, p->stream);
        }
        if (e == hipSuccess) continue;
        if (e != hipErrorNotSupported) HIP_TRY(e);
        for (int k = s.K - 1; k >= 0; k--) {
            const int rc = two_d ? inv_level_2d(p, l + k, run) : inv_level_1d(p, l + k, run);
            if (rc != PDWT_OK) return rc;
        }
    }
    return PDWT_OK;
}

int circshift_impl(pdwt_plan* p, int sr, int sc, int inplace) {
    // pdwt/src/common.cu:378-396
    const int Nr = p->info.Nr, Nc = p->info.Nc;
    sr %= Nr; sc %= Nc;
    if (sr < 0) sr += Nr;
    if (sc < 0) sc += Nc;
    if (p->info.ndims == 1) sr = 0;
    const long long n = (long long)p->batch * Nr * Nc;
    int rc = ensure_tmp(p, n);
    if (rc != PDWT_OK) return rc;
    Stamp st(p, "circshift");
    if (inplace) {
        HIP_TRY(hipMemcpyAsync(p->tmp, p->image(), (size_t)n * sizeof(real_t), hipMemcpyDeviceToDevice, p->stream));
        HIP_TRY(launch_circshift(p->tmp, p->image(), p->batch, Nr, Nc, sr, sc, p->stream));
    } else {
        HIP_TRY(launch_circshift(p->image(), p->tmp, p->batch, Nr, Nc, sr, sc, p->stream));
    }
    return PDWT_OK;
}

// beta / sqrt(2)^levels for the approximation band (pdwt/src/common.cu:229-236)
real_t app_beta(real_t beta, int levels, int normalize) {
    if (normalize > 0) {
        const int n2 = levels / 2;
        beta /= (real_t)(1 << n2);
        if (n2 * 2 != levels) beta = (real_t)(beta / 1.4142135623730951);
    }
    return beta;
}

int threshold_impl(pdwt_plan* p, int op, real_t beta, int do_app, int normalize, const char* what);
int threshold_sweep(pdwt_plan* p, int op, real_t beta, int do_app, int normalize, const char* what);

// apply a deferred soft_threshold now (every consumer of the coefficients other than the fused SWT
// inverse calls this first)
int materialize_pending(pdwt_plan* p) {
    if (!p->pend_soft) return PDWT_OK;
    p->pend_soft = false;
    return threshold_impl(p, EW_SOFT, p->pend_beta, 0, p->pend_normalize, "soft_threshold");
}

// A deferred threshold that the fused SWT inverse applied on the fly never reached the stored detail
// bands.  Before they become observable again (set_coeff re-arming the inverse, a device pointer handed
// out) it is applied to them, so that the result does not depend on whether the fused path was taken.
int materialize_consumed(pdwt_plan* p) {
    if (!p->soft_consumed) return PDWT_OK;
    p->soft_consumed = false;
    return threshold_sweep(p, EW_SOFT, p->consumed_beta, 0, p->consumed_normalize, "soft_threshold");
}

// can the inverse of this plan apply a soft threshold on the fly?  (fused 2D SWT kernels on every level)
bool can_defer_soft(const pdwt_plan* p) {
    static const bool no_lazy = getenv("PDWT_NO_LAZY_THRESHOLD") != nullptr;  // read once, like every other knob
    // (every level launch of a separable 2D SWT plan applies a pending threshold as it loads the details: tiles, split pair, fused
    // groups, the one-workgroup launch; until round 5 levels whose dilation does not divide the rows ran as direct passes that did not)
    return p->info.do_swt && p->info.ndims == 2 && p->do_separable && !no_lazy;
}

// soft / hard / proj_linf share one driver (pdwt/src/common.cu:219-308)
int threshold_impl(pdwt_plan* p, int op, real_t beta, int do_app, int normalize, const char* what) {
    if (p->state == PDWT_INVERSE)
        return fail(PDWT_ERR_STATE, "%s: cannot threshold coefficients, as they were modified by inverse()", what);
    {
        const int rc = materialize_pending(p);
        if (rc != PDWT_OK) return rc;
    }
    return threshold_sweep(p, op, beta, do_app, normalize, what);
}

int threshold_sweep(pdwt_plan* p, int op, real_t beta, int do_app, int normalize, const char* what) {
    const int L = p->info.nlevels, B = p->batch;
    if (do_app) {
        Stamp st(p, what);
        HIP_TRY(launch_ew(op, p->band(0), pad64(p->bands[0].elems(B)), app_beta(beta, L, normalize), p->stream));
    }
    const int per = p->info.ndims == 2 ? 3 : 1;
    if (normalize <= 0) {
        // every detail band shares beta: ONE sweep over the contiguous detail region
        Stamp st(p, what);
        const long long first = p->bands[1].off;
        HIP_TRY(launch_ew(op, p->arena + first, p->coeff_elems - first, beta, p->stream));
    } else {
        for (int l = 1; l <= L; l++) {
            beta = (real_t)(beta / 1.4142135623730951);  // common.cu:244
            const long long first = p->bands[per * (l - 1) + 1].off;
            const long long last = (l == L) ? p->coeff_elems : p->bands[per * l + 1].off;
            Stamp st(p, what);
            HIP_TRY(launch_ew(op, p->arena + first, last - first, beta, p->stream));
        }
    }
    return PDWT_OK;
}

std::string info_text(pdwt_plan* p) {
    // same lines as Wavelets::print_informations (pdwt/src/wt.cu:511-550)
    char buf[1024];
    std::string s;
    const char* yn[2] = {"no", "yes"};
    s += "------------- Wavelet transform infos ------------\n";
    if (p->info.ndims == 2) snprintf(buf, sizeof(buf), "Data dimensions : (%d, %d)\n", p->info.Nr, p->info.Nc);
    else if (p->info.Nr == 1) snprintf(buf, sizeof(buf), "Data dimensions : %d\n", p->info.Nc);
    else snprintf(buf, sizeof(buf), "Data dimensions : (%d, %d) [batched 1D transform]\n", p->info.Nr, p->info.Nc);
    s += buf;
    if (p->batch > 1) { snprintf(buf, sizeof(buf), "Batch : %d\n", p->batch); s += buf; }
    snprintf(buf, sizeof(buf), "Wavelet name : %s\nNumber of levels : %d\nStationary WT : %s\nCycle spinning : %s\n"
             "Separable transform : %s\n", p->wname, p->info.nlevels, yn[p->info.do_swt ? 1 : 0],
             yn[p->do_cycle_spinning ? 1 : 0], yn[p->do_separable ? 1 : 0]);
    s += buf;
    snprintf(buf, sizeof(buf), "Estimated memory footprint : %.2f MB\n",
             (double)(p->arena_elems + p->tmp_elems) * sizeof(real_t) / 1e6);
    s += buf;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess)
        snprintf(buf, sizeof(buf), "Running on device : %s\n", prop.name);
    else snprintf(buf, sizeof(buf), "Running on device : (unknown)\n");
    s += buf;
    s += "--------------------------------------------------\n";
    return s;
}

}  // namespace

// =============================================================================
#pragma GCC visibility push(default)
extern "C" {

const char* pdwt_last_error(void) { return g_last_error.c_str(); }
const char* pdwt_version(void) { return "pypwt_amd 0.1.0 (gfx950); API of pycudwt 1.0.3"; }

int pdwt_create(const real_t* img, int Nr, int Nc, const char* wname, int levels, int mem_is_on_host,
                int do_separable, int do_cycle_spinning, int do_swt, int ndim, pdwt_handle* out) {
    return create_impl(img, 1, Nr, Nc, wname, levels, mem_is_on_host, do_separable, do_cycle_spinning, do_swt,
                       ndim, -1, nullptr, out);
}

int pdwt_create_batched(const real_t* img, int batch, int Nr, int Nc, const char* wname, int levels,
                        int mem_is_on_host, int do_separable, int do_cycle_spinning, int do_swt, int ndim,
                        int device_id, void* hip_stream, pdwt_handle* out) {
    return create_impl(img, batch, Nr, Nc, wname, levels, mem_is_on_host, do_separable, do_cycle_spinning, do_swt,
                       ndim, device_id, hip_stream, out);
}

int pdwt_clone(pdwt_handle src, pdwt_handle* out) {
    CHECK_HANDLE(src);
    if (!out) return fail(PDWT_ERR_ARG, "pdwt_clone: out is null");
    *out = nullptr;
    DeviceGuard guard(src->device);
    {
        // the copy must not depend on whether the source took the fused path: a deferred threshold is applied and one
        // that the fused inverse consumed on the fly is written back before the arena is copied
        int rc0 = materialize_pending(src);
        if (rc0 == PDWT_OK) rc0 = materialize_consumed(src);
        if (rc0 != PDWT_OK) return rc0;
    }
    pdwt_plan* p = new pdwt_plan();
    p->device = src->device;
    p->batch = src->batch;
    p->info = src->info;
    p->do_separable = src->do_separable;
    p->do_cycle_spinning = src->do_cycle_spinning;
    p->state = src->state;
    memcpy(p->wname, src->wname, sizeof(p->wname));
    p->shift_r = src->shift_r;
    p->shift_c = src->shift_c;
    p->dec = src->dec;
    p->rec = src->rec;
    hipError_t e = pool_stream(p->device, &p->stream);
    if (e != hipSuccess) { delete p; return fail(PDWT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    p->own_stream = true;
    int rc = build_layout(p);
    if (rc != PDWT_OK) { pdwt_destroy(p); return rc; }
    p->tune = src->tune;
    build_schedule(p);  // not copied: a chain step needs the clone's own hand-off flags
    e = hipStreamSynchronize(src->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(p->arena, src->arena, (size_t)p->arena_elems * sizeof(real_t), hipMemcpyDeviceToDevice,
                           p->stream);
    // a bound source (pdwt_bind_image) keeps its image outside its arena: the clone OWNS a copy of it (it is not bound)
    if (e == hipSuccess && src->image_ext)
        e = hipMemcpyAsync(p->arena + p->image_off, src->image_ext, (size_t)p->batch * p->info.Nr * p->info.Nc * sizeof(real_t),
                           hipMemcpyDeviceToDevice, p->stream);
    if (e == hipSuccess && src->d_f2d) {
        e = device_malloc((void**)&p->d_f2d, (size_t)8 * kMaxTaps * kMaxTaps * sizeof(real_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(p->d_f2d, src->d_f2d, (size_t)8 * kMaxTaps * kMaxTaps * sizeof(real_t),
                               hipMemcpyDeviceToDevice, p->stream);
        p->f2d_custom = src->f2d_custom;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) { pdwt_destroy(p); return fail(PDWT_ERR_HIP, "clone copy: %s", hipGetErrorString(e)); }
    *out = p;
    return PDWT_OK;
}

int pdwt_destroy(pdwt_handle h) {
    if (!h) return PDWT_OK;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    clear_stamps(h);
    pool_free(h->device, h->arena, h->arena_bytes);
    if (h->tmp) (void)hipFree(h->tmp);
    pool_free(h->device, h->d_red, h->d_red_bytes);
    if (h->h_red) (void)hipHostFree(h->h_red);
    if (h->d_f2d) (void)hipFree(h->d_f2d);
    if (h->chain_flags) (void)hipFree(h->chain_flags);
    if (h->own_stream && h->stream) pool_return_stream(h->device, h->stream);
    delete h;
    return PDWT_OK;
}

int pdwt_forward(pdwt_handle h) {  // Wavelets::forward, wt.cu:236-269
    CHECK_HANDLE(h);
    if (h->state == PDWT_CREATION_ERROR)
        return fail(PDWT_ERR_STATE, "forward transform not computed, as there was an error when creating the wavelets");
    DeviceGuard guard(h->device);
    h->pend_soft = false;  // the coefficients a deferred threshold referred to are about to be overwritten
    h->soft_consumed = false;
    if (h->do_cycle_spinning) {  // wt.cu:242-246
        h->shift_r = rand() % h->info.Nr;
        h->shift_c = rand() % h->info.Nc;
        int rc = circshift_impl(h, h->shift_r, h->shift_c, 1);
        if (rc != PDWT_OK) return rc;
    }
    int rc = forward_impl(h);
    h->state = (rc == PDWT_OK) ? PDWT_FORWARD : PDWT_FORWARD_ERROR;
    return rc;
}

int pdwt_inverse(pdwt_handle h) {  // Wavelets::inverse, wt.cu:271-305
    CHECK_HANDLE(h);
    if (h->state == PDWT_INVERSE)
        return fail(PDWT_ERR_STATE, "W.inverse() has already been run. Inverse is available in W.get_image()");
    if (h->state == PDWT_FORWARD_ERROR || h->state == PDWT_THRESHOLD_ERROR || h->state == PDWT_CREATION_ERROR)
        return fail(PDWT_ERR_STATE, "inverse transform not computed, as there was an error in a previous stage");
    DeviceGuard guard(h->device);
    int rc = inverse_impl(h);
    if (h->pend_soft && rc == PDWT_OK) {  // consumed by the fused kernels, not written back (materialize_consumed)
        h->soft_consumed = true;
        h->consumed_beta = h->pend_beta;
        h->consumed_normalize = h->pend_normalize;
    }
    h->pend_soft = false;
    if (rc == PDWT_OK && h->do_cycle_spinning) rc = circshift_impl(h, -h->shift_r, -h->shift_c, 1);  // wt.cu:303
    h->state = (rc == PDWT_OK) ? PDWT_INVERSE : PDWT_INVERSE_ERROR;
    return rc;
}

int pdwt_soft_threshold(pdwt_handle h, real_t beta, int do_app, int normalize) {  // wt.cu:308-315
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    if (h->state != PDWT_INVERSE && !do_app && beta >= 0.f && can_defer_soft(h)) {
        const int rc = materialize_pending(h);  // an earlier pending threshold composes: apply it first
        if (rc != PDWT_OK) return rc;
        h->pend_soft = true;
        h->pend_beta = beta;
        h->pend_normalize = normalize;
        return PDWT_OK;
    }
    return threshold_impl(h, EW_SOFT, beta, do_app, normalize, "soft_threshold");
}

int pdwt_hard_threshold(pdwt_handle h, real_t beta, int do_app, int normalize) {  // wt.cu:318-325
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    return threshold_impl(h, EW_HARD, beta, do_app, normalize, "hard_threshold");
}

int pdwt_proj_linf(pdwt_handle h, real_t beta, int do_app) {  // wt.cu:349-356
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    return threshold_impl(h, EW_LINF, beta, do_app, 0, "proj_linf");
}

int pdwt_group_soft_threshold(pdwt_handle h, real_t beta, int do_app, int normalize) {  // wt.cu:329-336
    CHECK_HANDLE(h);
    if (h->state == PDWT_INVERSE)
        return fail(PDWT_ERR_STATE, "cannot threshold coefficients, as they were modified by inverse()");
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const int L = h->info.nlevels, B = h->batch;
    const int per = h->info.ndims == 2 ? 3 : 1;
    for (int l = 1; l <= L; l++) {  // common.cu:311-341
        if (normalize > 0) beta = (real_t)(beta / 1.4142135623730951);
        real_t* d0 = h->band(per * (l - 1) + 1);
        real_t* d1 = per == 3 ? h->band(per * (l - 1) + 2) : nullptr;
        real_t* d2 = per == 3 ? h->band(per * (l - 1) + 3) : nullptr;
        real_t* ap = (do_app && l == L) ? h->band(0) : nullptr;
        Stamp st(h, "group_soft_threshold");
        HIP_TRY(launch_group_soft(d0, d1, d2, ap, h->bands[per * (l - 1) + 1].elems(B), beta, per, h->stream));
    }
    return PDWT_OK;
}

int pdwt_shrink(pdwt_handle h, real_t beta, int do_app) {  // wt.cu:340-347, common.cu:347-371
    CHECK_HANDLE(h);
    if (h->state == PDWT_INVERSE)
        return fail(PDWT_ERR_STATE, "cannot threshold coefficients, as they were modified by inverse()");
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const long long first = do_app ? 0 : h->bands[1].off;
    Stamp st(h, "shrink");
    HIP_TRY(launch_ew(EW_SCALE, h->arena + first, h->coeff_elems - first, 1.0f / (1.0f + beta), h->stream));
    return PDWT_OK;
}

int pdwt_circshift(pdwt_handle h, int sr, int sc, int inplace) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    return circshift_impl(h, sr, sc, inplace);
}

static int norms_impl(pdwt_handle h, double out[2]) {
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    {
        Stamp st(h, "norms");
        HIP_TRY(launch_norms(h->arena, h->coeff_elems, h->d_red, h->d_red, h->stream));
    }
    if (!h->h_red && hipHostMalloc((void**)&h->h_red, 2 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        h->h_red = nullptr;
    }
    double* land = h->h_red ? h->h_red : out;
    HIP_TRY(hipMemcpyAsync(land, h->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (land != out) { out[0] = land[0]; out[1] = land[1]; }
    return PDWT_OK;
}

// NEW (round 6): the norms WITHOUT the round trip -- the two fp64 results stay on the device (d_out2, or the plan's own slot:
// pdwt_norms_slot), nothing is synchronised: an iterative solver reads them from its own kernels or copies them when it
// wants to.  The blocking getters below stay for API parity (wt.cu:368-416 return host floats).
int pdwt_norms_async(pdwt_handle h, double* d_out2) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    const int rc0 = materialize_pending(h);
    if (rc0 != PDWT_OK) return rc0;
    Stamp st(h, "norms");
    HIP_TRY(launch_norms(h->arena, h->coeff_elems, h->d_red, d_out2 ? d_out2 : h->d_red, h->stream));
    return PDWT_OK;
}

int pdwt_norms_slot(pdwt_handle h, double** d_ptr) {
    CHECK_HANDLE(h);
    if (!d_ptr) return fail(PDWT_ERR_ARG, "pdwt_norms_slot: null argument");
    *d_ptr = h->d_red;
    return PDWT_OK;
}

// soft_threshold (wt.cu:308-315, common.cu:219-249) and the norms of what it leaves (wt.cu:368-416) in ONE sweep over the
// coefficients instead of a sweep and a reduction (4096^2: 20 + 28 us -> one 21-us launch and a one-block final sum).  A plan
// whose inverse applies the threshold as it loads the details (2D SWT) keeps doing so: the sweep is then read-only.
int pdwt_soft_threshold_norms_async(pdwt_handle h, real_t beta, int do_app, int normalize, double* d_out2) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    if (h->state == PDWT_INVERSE)
        return fail(PDWT_ERR_STATE, "soft_threshold: cannot threshold coefficients, as they were modified by inverse()");
    const int rc0 = materialize_pending(h);  // an earlier pending threshold composes: apply it first
    if (rc0 != PDWT_OK) return rc0;
    const bool defer = !do_app && beta >= 0.f && can_defer_soft(h);
    if (defer) {
        h->pend_soft = true;
        h->pend_beta = beta;
        h->pend_normalize = normalize;
    }
    const int L = h->info.nlevels, per = h->info.ndims == 2 ? 3 : 1;
    const long long first = h->bands[1].off;
    double* out = d_out2 ? d_out2 : h->d_red;
    Stamp st(h, "soft_threshold+norms");
    int used = 0, nb = 0;
    if (normalize <= 0) {
        HIP_TRY(launch_soft_norms(h->arena, h->coeff_elems, first, app_beta(beta, L, normalize), beta, !do_app, !defer, h->d_red, 0,
                                  norms_max_blocks(), &nb, h->stream));
        used = nb;
    } else {
        const int share = norms_max_blocks() / (L + 1) > 0 ? norms_max_blocks() / (L + 1) : 1;
        // the approximation band (thresholded with its own beta, or as it is), then every level with beta / sqrt(2)^l (common.cu:244)
        HIP_TRY(launch_soft_norms(h->arena, first, first, app_beta(beta, L, normalize), real_t(0), !do_app, !defer, h->d_red, used, share, &nb,
                                  h->stream));
        used += nb;
        real_t b = beta;
        for (int l = 1; l <= L; l++) {
            b = (real_t)(b / 1.4142135623730951);
            const long long lo = h->bands[per * (l - 1) + 1].off;
            const long long hi = (l == L) ? h->coeff_elems : h->bands[per * l + 1].off;
            HIP_TRY(launch_soft_norms(h->arena + lo, hi - lo, 0, real_t(0), b, false, !defer, h->d_red, used, share, &nb, h->stream));
            used += nb;
        }
    }
    HIP_TRY(launch_norms_final(h->d_red, used, out, h->stream));
    return PDWT_OK;
}

int pdwt_norm1(pdwt_handle h, real_t* out) {  // wt.cu:396-416
    CHECK_HANDLE(h);
    if (!out) return fail(PDWT_ERR_ARG, "pdwt_norm1: out is null");
    double r[2];
    int rc = norms_impl(h, r);
    if (rc == PDWT_OK) *out = (real_t)r[0];
    return rc;
}

int pdwt_norm2sq(pdwt_handle h, real_t* out) {  // wt.cu:368-393
    CHECK_HANDLE(h);
    if (!out) return fail(PDWT_ERR_ARG, "pdwt_norm2sq: out is null");
    double r[2];
    int rc = norms_impl(h, r);
    if (rc == PDWT_OK) *out = (real_t)r[1];
    return rc;
}

int pdwt_add_wavelet(pdwt_handle dst, pdwt_handle src, real_t alpha) {  // wt.cu:622-655
    CHECK_HANDLE(dst);
    CHECK_HANDLE(src);
    if (dst->info.nlevels != src->info.nlevels || strcasecmp(dst->wname, src->wname)) {
        fail(PDWT_ERR_MISMATCH, "add_wavelet(): right operand is not the same transform (wname, level)");
        return -1;
    }
    if (dst->state == PDWT_INVERSE || src->state == PDWT_INVERSE) {
        fail(PDWT_ERR_STATE, "add_wavelet(): this operation makes no sense when wavelet has just been inverted");
        return 1;
    }
    if (dst->info.Nr != src->info.Nr || dst->info.Nc != src->info.Nc || dst->info.ndims != src->info.ndims ||
        dst->batch != src->batch) {
        fail(PDWT_ERR_MISMATCH, "add_wavelet(): operands do not have the same geometry");
        return -2;
    }
    if ((dst->info.do_swt != 0) != (src->info.do_swt != 0)) {
        fail(PDWT_ERR_MISMATCH, "add_wavelet(): operands should both use SWT or DWT");
        return -3;
    }
    if (dst->do_cycle_spinning && src->do_cycle_spinning &&
        (dst->shift_r != src->shift_r || dst->shift_c != src->shift_c)) {
        fail(PDWT_ERR_MISMATCH, "add_wavelet(): operands do not have the same current shift");
        return -4;
    }
    if (dst->device != src->device) {
        fail(PDWT_ERR_MISMATCH, "add_wavelet(): operands live on different devices");
        return -2;
    }
    DeviceGuard guard(dst->device);
    {
        int rc0 = materialize_pending(dst);
        if (rc0 == PDWT_OK) rc0 = materialize_pending(src);
        if (rc0 != PDWT_OK) return rc0;
    }
    if (src->stream != dst->stream) HIP_TRY(hipStreamSynchronize(src->stream));
    Stamp st(dst, "add_wavelet");
    HIP_TRY(launch_axpy(dst->arena, src->arena, dst->coeff_elems, alpha, dst->stream));
    return PDWT_OK;
}

long long pdwt_get_image(pdwt_handle h, real_t* dst) {  // wt.cu:419-422
    if (!h || !dst) return fail(PDWT_ERR_ARG, "pdwt_get_image: null argument");
    DeviceGuard guard(h->device);
    const long long n = (long long)h->batch * h->info.Nr * h->info.Nc;
    HIP_TRY(hipMemcpyAsync(dst, h->image(), (size_t)n * sizeof(real_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return n;
}

long long pdwt_coeff_count(pdwt_handle h, int num, int* rows, int* cols) {
    if (!h) return fail(PDWT_ERR_ARG, "null plan handle");
    if (num < 0 || num >= (int)h->bands.size()) return fail(PDWT_ERR_ARG, "coefficient index %d out of range", num);
    if (rows) *rows = h->bands[num].rows;
    if (cols) *cols = h->bands[num].cols;
    return h->bands[num].elems(h->batch);
}

long long pdwt_get_coeff(pdwt_handle h, real_t* dst, int num) {  // wt.cu:473-506
    if (!h || !dst) return fail(PDWT_ERR_ARG, "pdwt_get_coeff: null argument");
    if (num < 0 || num >= (int)h->bands.size()) return fail(PDWT_ERR_ARG, "coefficient index %d out of range", num);
    if (h->state == PDWT_INVERSE) {
        fail(PDWT_ERR_STATE, "get_coeff(): inverse() has been performed, the coefficients has been modified and do not make sense anymore.");
        return 0;
    }
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const long long n = h->bands[num].elems(h->batch);
    HIP_TRY(hipMemcpyAsync(dst, h->band(num), (size_t)n * sizeof(real_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return n;
}

long long pdwt_coeff_region(pdwt_handle h, long long* band_offsets, int capacity) {
    if (!h) return fail(PDWT_ERR_ARG, "pdwt_coeff_region: null handle");
    if (band_offsets)
        for (int i = 0; i < (int)h->bands.size() && i < capacity; i++) band_offsets[i] = h->bands[i].off - h->bands[0].off;
    return h->coeff_elems - h->bands[0].off;
}

long long pdwt_get_coeff_region(pdwt_handle h, real_t* dst) {
    if (!h || !dst) return fail(PDWT_ERR_ARG, "pdwt_get_coeff_region: null argument");
    if (h->state == PDWT_INVERSE) {
        fail(PDWT_ERR_STATE, "get_coeff(): inverse() has been performed, the coefficients has been modified and do not make sense anymore.");
        return 0;
    }
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const long long n = h->coeff_elems - h->bands[0].off;
    HIP_TRY(hipMemcpyAsync(dst, h->band(0), (size_t)n * sizeof(real_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return n;
}

long long pdwt_get_image_at(pdwt_handle h, real_t* dst, int image_index) {  // wt.cu:419-422, one image of a batch
    if (!h || !dst) return fail(PDWT_ERR_ARG, "pdwt_get_image_at: null argument");
    if (image_index < 0 || image_index >= h->batch)
        return fail(PDWT_ERR_ARG, "pdwt_get_image_at: image %d out of range (batch %d)", image_index, h->batch);
    DeviceGuard guard(h->device);
    const long long n = (long long)h->info.Nr * h->info.Nc;
    HIP_TRY(hipMemcpyAsync(dst, h->image() + (long long)image_index * n, (size_t)n * sizeof(real_t),
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return n;
}

long long pdwt_get_coeff_at(pdwt_handle h, real_t* dst, int num, int image_index) {  // wt.cu:473-506, one image
    if (!h || !dst) return fail(PDWT_ERR_ARG, "pdwt_get_coeff_at: null argument");
    if (num < 0 || num >= (int)h->bands.size()) return fail(PDWT_ERR_ARG, "coefficient index %d out of range", num);
    if (image_index < 0 || image_index >= h->batch)
        return fail(PDWT_ERR_ARG, "pdwt_get_coeff_at: image %d out of range (batch %d)", image_index, h->batch);
    if (h->state == PDWT_INVERSE) {
        fail(PDWT_ERR_STATE, "get_coeff(): inverse() has been performed, the coefficients has been modified and do not make sense anymore.");
        return 0;
    }
    DeviceGuard guard(h->device);
    {
        const int rc0 = materialize_pending(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const long long n = h->bands[num].elems(1);
    HIP_TRY(hipMemcpyAsync(dst, h->band(num) + (long long)image_index * n, (size_t)n * sizeof(real_t),
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return n;
}

int pdwt_set_image(pdwt_handle h, const real_t* src, int mem_is_on_device) {  // wt.cu:425-431
    CHECK_HANDLE(h);
    if (!src) return fail(PDWT_ERR_ARG, "pdwt_set_image: src is null");
    DeviceGuard guard(h->device);
    const long long n = (long long)h->batch * h->info.Nr * h->info.Nc;
    // the plan's own image buffer (the caller filled it through pdwt_image_ptr, on the plan's stream or ordered with it
    // by pdwt_wait_for_stream): nothing to copy, nothing to wait for -- the call only makes the image current
    const bool in_place = mem_is_on_device && src == h->image();
    if (!in_place) {
        HIP_TRY(hipMemcpyAsync(h->image(), src, (size_t)n * sizeof(real_t),
                               mem_is_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    // host AND device sources: the copy has completed when the call returns, so the caller may reuse or free the
    // source at once (a torch temporary goes back to the caching allocator as soon as the Python call returns).
    // Ordering of the source's PRODUCER with this copy is the caller's: pdwt_wait_for_stream / pdwt_sync_producer.
    // a threshold the fused inverse applied on the fly is written back before the state leaves PDWT_INVERSE:
    // get_coeff / norms / add_wavelet are legal again after set_image and must see thresholded details (wt.cu:308-315)
    {
        const int rc0 = materialize_consumed(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    h->state = PDWT_INIT;
    return PDWT_OK;
}

int pdwt_set_coeff(pdwt_handle h, const real_t* src, int num, int mem_is_on_device) {  // wt.cu:435-466
    CHECK_HANDLE(h);
    if (!src) return fail(PDWT_ERR_ARG, "pdwt_set_coeff: src is null");
    if (num < 0 || num >= (int)h->bands.size()) return fail(PDWT_ERR_ARG, "coefficient index %d out of range", num);
    DeviceGuard guard(h->device);
    {
        int rc0 = materialize_pending(h);  // a deferred threshold applies to the OLD contents only
        if (rc0 == PDWT_OK) rc0 = materialize_consumed(h);
        if (rc0 != PDWT_OK) return rc0;
    }
    const long long n = h->bands[num].elems(h->batch);
    if (!(mem_is_on_device && src == h->band(num))) {  // the band's own buffer: in place, see pdwt_set_image
        HIP_TRY(hipMemcpyAsync(h->band(num), src, (size_t)n * sizeof(real_t),
                               mem_is_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));  // see pdwt_set_image
    }
    // The reference forbids coefficient access after inverse() because its inverse overwrites the
    // approximation band d_coeffs[0] (wt.cu:272-275); once the caller has supplied that band again the
    // coefficients are current, so a new inverse() is meaningful (the reference would still refuse it).
    if (num == 0 && h->state == PDWT_INVERSE) h->state = PDWT_FORWARD;
    return PDWT_OK;
}

intptr_t pdwt_image_ptr(pdwt_handle h) { return h ? (intptr_t)h->image() : 0; }

int pdwt_bind_image(pdwt_handle h, void* device_ptr) {
    CHECK_HANDLE(h);
    // 16 bytes: every tuned kernel stages the image with 16-B accesses, and not every launcher checks its input's alignment
    if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15))
        return fail(PDWT_ERR_ARG, "pdwt_bind_image: the pointer must be 16-byte aligned");
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));  // nothing in flight may still use the old location
    h->image_ext = static_cast<real_t*>(device_ptr);
    return PDWT_OK;
}

int pdwt_copy(pdwt_handle h, void* dst, const void* src, long long count, int kind) {
    CHECK_HANDLE(h);
    if (!dst || !src || count < 0 || kind < 0 || kind > 2) return fail(PDWT_ERR_ARG, "pdwt_copy: bad arguments");
    if (count == 0) return PDWT_OK;
    DeviceGuard guard(h->device);
    const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)count * sizeof(real_t), k, h->stream));
    if (kind != 0) HIP_TRY(hipStreamSynchronize(h->stream));  // host memory: the caller may touch it when the call returns
    return PDWT_OK;
}

intptr_t pdwt_coeff_ptr(pdwt_handle h, int num) {
    if (!h || num < 0 || num >= (int)h->bands.size()) return 0;
    {
        DeviceGuard guard(h->device);
        // the caller will read device memory directly
        if (materialize_pending(h) != PDWT_OK || materialize_consumed(h) != PDWT_OK) return 0;
    }
    return (intptr_t)h->band(num);
}

int pdwt_set_filters_forward(pdwt_handle h, const char* name, unsigned int len, const real_t* f1, const real_t* f2,
                             const real_t* f3, const real_t* f4) {  // wt.cu:558-578
    CHECK_HANDLE(h);
    if (len > PDWT_MAX_FILTER_WIDTH || len < 1)
        return fail(PDWT_ERR_FILTER_LEN, "set_filters_forward(): filter length (%u) exceeds the maximum size (%d)", len,
                    PDWT_MAX_FILTER_WIDTH);
    if (!f1 || !f2) return fail(PDWT_ERR_ARG, "set_filters_forward(): filter1/filter2 are required");
    DeviceGuard guard(h->device);
    if (h->do_separable) {
        memset(&h->dec, 0, sizeof(h->dec));
        memcpy(h->dec.lo, f1, len * sizeof(real_t));
        memcpy(h->dec.hi, f2, len * sizeof(real_t));
    } else {
        if (!f3 || !f4)
            return fail(PDWT_ERR_ARG, "set_filters_forward(): expected argument 4 and 5 for non-separable filtering");
        if (!h->d_f2d) HIP_TRY(device_malloc((void**)&h->d_f2d, (size_t)8 * kMaxTaps * kMaxTaps * sizeof(real_t)));
        const real_t* f[4] = {f1, f2, f3, f4};
        for (int k = 0; k < 4; k++)
            HIP_TRY(hipMemcpyAsync(h->d_f2d + (size_t)k * len * len, f[k], (size_t)len * len * sizeof(real_t),
                                   hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->f2d_custom = true;
    }
    h->info.hlen = (int)len;
    if (name) snprintf(h->wname, sizeof(h->wname), "%s", name);
    build_schedule(h);  // the fused kernels' eligibility depends on the filter length
    return PDWT_OK;
}

int pdwt_set_filters_inverse(pdwt_handle h, const real_t* f1, const real_t* f2, const real_t* f3,
                             const real_t* f4) {  // wt.cu:583-600
    CHECK_HANDLE(h);
    if (!f1 || !f2) return fail(PDWT_ERR_ARG, "set_filters_inverse(): filter1/filter2 are required");
    const unsigned len = (unsigned)h->info.hlen;
    DeviceGuard guard(h->device);
    if (h->do_separable) {
        memset(&h->rec, 0, sizeof(h->rec));
        memcpy(h->rec.lo, f1, len * sizeof(real_t));
        memcpy(h->rec.hi, f2, len * sizeof(real_t));
    } else {
        if (!f3 || !f4)
            return fail(PDWT_ERR_ARG, "set_filters_inverse(): expected argument 4 and 5 for non-separable filtering");
        if (!h->d_f2d) HIP_TRY(device_malloc((void**)&h->d_f2d, (size_t)8 * kMaxTaps * kMaxTaps * sizeof(real_t)));
        const real_t* f[4] = {f1, f2, f3, f4};
        for (int k = 0; k < 4; k++)
            HIP_TRY(hipMemcpyAsync(h->d_f2d + (size_t)(4 + k) * len * len, f[k], (size_t)len * len * sizeof(real_t),
                                   hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->f2d_custom = true;
    }
    return PDWT_OK;
}

int pdwt_get_info(pdwt_handle h, pdwt_info* info, int* do_separable, int* do_cycle_spinning, int* state,
                  int* batch) {
    CHECK_HANDLE(h);
    if (info) *info = h->info;
    if (do_separable) *do_separable = h->do_separable;
    if (do_cycle_spinning) *do_cycle_spinning = h->do_cycle_spinning;
    if (state) *state = h->state;
    if (batch) *batch = h->batch;
    return PDWT_OK;
}

int pdwt_print_info(pdwt_handle h) {
    CHECK_HANDLE(h);
    fputs(info_text(h).c_str(), stdout);
    fflush(stdout);
    return PDWT_OK;
}

int pdwt_info_string(pdwt_handle h, char* buf, size_t n) {
    CHECK_HANDLE(h);
    if (!buf || n == 0) return fail(PDWT_ERR_ARG, "pdwt_info_string: empty buffer");
    snprintf(buf, n, "%s", info_text(h).c_str());
    return PDWT_OK;
}

int pdwt_current_shift(pdwt_handle h, int* sr, int* sc) {
    CHECK_HANDLE(h);
    if (sr) *sr = h->shift_r;
    if (sc) *sc = h->shift_c;
    return PDWT_OK;
}

int pdwt_wavelet_count(void) { return wavelet_count(); }

const char* pdwt_wavelet_name(int index) {
    const WaveletEntry* w = wavelet_at(index);
    return w ? w->name : nullptr;
}

int pdwt_wavelet_filters(const char* wname, real_t* banks, int capacity) {
    const WaveletEntry* w = find_wavelet(wname);
    if (!w) return fail(PDWT_ERR_WAVELET, "unknown wavelet name %s", wname ? wname : "(null)");
    if (banks) {
        if (capacity < 4 * w->hlen) return fail(PDWT_ERR_ARG, "pdwt_wavelet_filters: capacity %d < %d", capacity, 4 * w->hlen);
        for (int i = 0; i < w->hlen; i++) {
            banks[i] = (real_t)w->dec_lo[i];
            banks[w->hlen + i] = (real_t)w->dec_hi[i];
            banks[2 * w->hlen + i] = (real_t)w->rec_lo[i];
            banks[3 * w->hlen + i] = (real_t)w->rec_hi[i];
        }
    }
    return w->hlen;
}

int pdwt_synchronize(pdwt_handle h) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PDWT_OK;
}

int pdwt_set_stream(pdwt_handle h, void* hip_stream) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        pool_return_stream(h->device, h->stream);
    }
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return PDWT_OK;
}

void* pdwt_get_stream(pdwt_handle h) { return h ? (void*)h->stream : nullptr; }
int pdwt_device(pdwt_handle h) { return h ? h->device : -1; }

int pdwt_fill_image_hash(pdwt_handle h, uint32_t seed, real_t scale, long long index_offset) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    Stamp st(h, "fill_hash");
    HIP_TRY(launch_fill_hash(h->image(), (long long)h->batch * h->info.Nr * h->info.Nc, seed, scale, index_offset,
                             h->stream));
    h->state = PDWT_INIT;
    return PDWT_OK;
}

int pdwt_time_level(pdwt_handle h, int level, int inverse, int reps, float* ms_per_launch) {
    CHECK_HANDLE(h);
    if (!ms_per_launch || reps < 1 || level < 1 || level > h->info.nlevels)
        return fail(PDWT_ERR_ARG, "pdwt_time_level: bad arguments");
    DeviceGuard guard(h->device);
    struct Scope {  // restores the timing flag and frees the events on every exit path
        pdwt_plan* p;
        bool was;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scope() {
            p->timing = was;
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } sc{h, h->timing};
    h->timing = false;
    HIP_TRY(hipEventCreate(&sc.e0));
    HIP_TRY(hipEventCreate(&sc.e1));
    int rc = PDWT_OK;
    for (int i = 0; i < 3 && rc == PDWT_OK; i++) rc = inverse ? inverse_impl(h, level) : forward_impl(h, level);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipEventRecord(sc.e0, h->stream));
    for (int i = 0; i < reps && rc == PDWT_OK; i++) rc = inverse ? inverse_impl(h, level) : forward_impl(h, level);
    HIP_TRY(hipEventRecord(sc.e1, h->stream));
    HIP_TRY(hipEventSynchronize(sc.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    *ms_per_launch = ms / (float)reps;
    return rc;
}

int pdwt_time_copy(pdwt_handle h, long long elems, int reps, float* ms_per_launch) {
    CHECK_HANDLE(h);
    // source: the image buffer, or -- for a footprint larger than the image (an SWT launch moves several planes) -- the
    // coefficient region, which the copy only reads
    const long long img = (long long)h->batch * h->info.Nr * h->info.Nc;
    if (!ms_per_launch || reps < 1 || elems < 4) return fail(PDWT_ERR_ARG, "pdwt_time_copy: bad arguments");
    const real_t* src = h->image();
    long long cap = img;
    if (elems > img && h->coeff_elems > img) {
        src = h->arena;
        cap = h->coeff_elems;
    }
    if (elems > cap) elems = cap;
    elems &= ~3LL;
    DeviceGuard guard(h->device);
    int rc = ensure_tmp(h, elems);
    if (rc != PDWT_OK) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Ev {
        hipEvent_t *a, *b;
        ~Ev() {
            if (*a) (void)hipEventDestroy(*a);
            if (*b) (void)hipEventDestroy(*b);
        }
    } ev{&e0, &e1};
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) HIP_TRY(launch_copy(src, h->tmp, elems, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; i++) HIP_TRY(launch_copy(src, h->tmp, elems, h->stream));
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / (float)reps;
    return PDWT_OK;
}

long long pdwt_copy_capacity(pdwt_handle h) {
    if (!h) return 0;
    const long long img = (long long)h->batch * h->info.Nr * h->info.Nc;
    return h->coeff_elems > img ? h->coeff_elems : img;
}

int pdwt_trim_pool(void) {
    DevicePool& P = device_pool();
    std::vector<DevicePool::Block> drop;
    std::vector<std::pair<int, hipStream_t>> st;
    {
        std::lock_guard<std::mutex> g(P.m);
        drop.swap(P.blocks);
        st.swap(P.streams);
        P.cached = 0;
    }
    for (auto& b : drop) {
        DeviceGuard guard(b.dev);
        (void)hipFree(b.p);
    }
    for (auto& x : st) {
        DeviceGuard guard(x.first);
        (void)hipStreamDestroy(x.second);
    }
    return (int)drop.size();
}

int pdwt_schedule_string(pdwt_handle h, char* buf, size_t n) {
    CHECK_HANDLE(h);
    if (!buf || n == 0) return fail(PDWT_ERR_ARG, "pdwt_schedule_string: no buffer");
    static const char* kind[] = {"LEVEL", "PYR2", "STRIP2", "FUSED1D", "WAVE2", "REG1D", "SWTF", "PYR3", "CHAIN", "TAIL"};
    std::string out;
    for (int dir = 0; dir < 2; dir++) {
        out += dir ? "inv:" : "fwd:";
        for (const pdwt::Step& s : dir ? h->sched_inv : h->sched_fwd) {
            char t[64];
            if (s.K > 1) snprintf(t, sizeof t, " %s[%d-%d]", kind[s.kind], s.level, s.level + s.K - 1);
            else snprintf(t, sizeof t, " %s[%d]", kind[s.kind], s.level);
            out += t;
        }
        out += "\n";
    }
    const size_t m = out.size() < n - 1 ? out.size() : n - 1;
    memcpy(buf, out.data(), m);
    buf[m] = 0;
    return (int)m;
}

int pdwt_wait_for_stream(pdwt_handle h, void* producer_stream) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    hipStream_t prod = (hipStream_t)producer_stream;
    if (prod == h->stream) return PDWT_OK;  // same queue: already ordered
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, prod);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ev, 0);
    (void)hipEventDestroy(ev);  // destruction is deferred until the event has completed
    if (e != hipSuccess) return fail(PDWT_ERR_HIP, "pdwt_wait_for_stream: %s", hipGetErrorString(e));
    return PDWT_OK;
}

int pdwt_sync_producer(int device_id, void* producer_stream, int whole_device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PDWT_ERR_HIP, "no HIP device available");
    if (device_id < 0 && hipGetDevice(&device_id) != hipSuccess) return fail(PDWT_ERR_HIP, "hipGetDevice failed");  // -1: the current device
    if (device_id >= ndev) return fail(PDWT_ERR_ARG, "device %d out of range (%d devices)", device_id, ndev);
    DeviceGuard guard(device_id);
    hipError_t e = whole_device ? hipDeviceSynchronize() : hipStreamSynchronize((hipStream_t)producer_stream);
    if (e != hipSuccess) return fail(PDWT_ERR_HIP, "pdwt_sync_producer: %s", hipGetErrorString(e));
    return PDWT_OK;
}

int pdwt_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return ndev;
}

int pdwt_device_of_pointer(const void* ptr) {
    if (!ptr) return fail(PDWT_ERR_ARG, "pdwt_device_of_pointer: null pointer");
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(PDWT_ERR_ARG, "pdwt_device_of_pointer: not an address the HIP runtime knows");
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeArray)
        return fail(PDWT_ERR_ARG, "pdwt_device_of_pointer: host memory");
    return at.device;
}

int pdwt_set_tuning(const char* key, int value) {
    if (key && !strcmp(key, "wave_min_log2")) return set_wave_min_log2(value);
    if (key && !strcmp(key, "dwt_split_fwd")) return set_dwt_split_min(0, value);
    if (key && !strcmp(key, "dwt_split_inv")) return set_dwt_split_min(1, value);
    if (key && !strcmp(key, "lds_max_log2")) return set_lds_max_log2(value);
    if (key && !strcmp(key, "ring_min_log2")) return set_ring_min_log2(value);
    if (key && !strcmp(key, "long_fwd")) return set_long_min_taps(0, value);
    if (key && !strcmp(key, "long_inv")) return set_long_min_taps(1, value);
    if (key && !strcmp(key, "swt_colstream")) return set_swt_colstream_min(value);
    if (key && !strcmp(key, "swt_fwdstream")) return set_swt_fwdstream_min(value);
    if (key && !strcmp(key, "swt_invstream")) return set_swt_invstream_min(value);
    if (key && !strcmp(key, "wave2")) return set_wave2_enabled(value);
    if (key && !strcmp(key, "reg1d")) return set_reg1d_enabled(value);
    if (key && !strcmp(key, "swt_fused")) return set_swt_fused_enabled(value);
    if (key && !strcmp(key, "chain")) return set_chain_enabled(value);
    if (key && !strcmp(key, "swt_split_fwd")) return set_swt_split_min(0, value);
    if (key && !strcmp(key, "swt_split_inv")) return set_swt_split_min(1, value);
    if (key && !strcmp(key, "chain_timeout")) return set_chain_timeout(value);
    return fail(PDWT_ERR_ARG, "pdwt_set_tuning: unknown key %s", key ? key : "(null)");
}

int pdwt_enable_kernel_timing(pdwt_handle h, int enable) {
    CHECK_HANDLE(h);
    h->timing = enable != 0;
    return PDWT_OK;
}

int pdwt_reset_kernel_times(pdwt_handle h) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    HIP_TRY(hipStreamSynchronize(h->stream));
    clear_stamps(h);
    return PDWT_OK;
}

int pdwt_kernel_families(pdwt_handle h, char (*families)[16], int cap) {
    CHECK_HANDLE(h);
    const int n = (int)h->stamps.size();
    for (int i = 0; i < n && i < cap; i++)
        if (families) memcpy(families[i], h->stamps[i].family, 16);
    return n;
}

int pdwt_kernel_times(pdwt_handle h, float* ms, char (*names)[48], int cap) {
    CHECK_HANDLE(h);
    DeviceGuard guard(h->device);
    const int n = (int)h->stamps.size();
    if (n > 0 && !h->stamps[n - 1].stop) {  // closing event for the last launch
        HIP_TRY(hipEventCreate(&h->stamps[n - 1].stop));
        HIP_TRY(hipEventRecord(h->stamps[n - 1].stop, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n && i < cap; i++) {
        float t = 0.f;
        hipEvent_t end = (i + 1 < n) ? h->stamps[i + 1].start : h->stamps[i].stop;
        HIP_TRY(hipEventElapsedTime(&t, h->stamps[i].start, end));
        if (ms) ms[i] = t;
        if (names) memcpy(names[i], h->stamps[i].name, 48);
    }
    return n;
}

}  // extern "C"
#pragma GCC visibility pop
