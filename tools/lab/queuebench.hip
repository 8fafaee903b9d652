// queuebench.hip -- kill criterion for "all levels of one image in ONE launch" (developer tool, not shipped).
//
// The 4096^2 forward pyramid moves 64/16/4/1 MiB per level in dependent launches; levels 2-4 cost 13 us for 25 % of the
// bytes because each launch is a latency chain.  Proxy of the alternative: ONE launch of every level's LDS tiles
// (one tile per workgroup, 8 x 64 outputs of four bands from a (16+6) x (128+8) staged region, the real kernel's
// geometry for 8 taps), ordered by a host-built schedule so that a tile's producers always have smaller block ids;
// a level-(l+1) tile polls per-tile-row completion counters of level l, reads A_l with sc1 loads, and every tile of a
// level that has a consumer writes its A band with 16-B sc1 (write-through) stores, drains them and bumps its row's
// counter (MI355X_MICROARCH.md "Valid forms": sc1 stores + vmcnt(0) + barrier + one-lane counter add; sc1 loads).
// The arithmetic is a 2-tap stand-in that touches both ends of the halo, so a stale line shows up in the comparison
// with the launch-per-level result (every word, several epochs with different inputs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queuebench.hip -o tools/bin/queuebench
//   tools/bin/queuebench [lag_rows] [levels] [N]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kMaxL = 6;
constexpr int TX = 64, TY = 8, HL = 8, C = HL / 2 - 1;  // outputs per tile, taps, analysis centre
constexpr int PADL = (4 - (C & 3)) & 3;                 // 1
constexpr int RXA = (PADL + 2 * TX + HL - 2 + 3) & ~3;  // 136 staged columns
constexpr int RY = 2 * TY + HL - 2;                     // 22 staged rows
constexpr int V4 = RXA / 4;
constexpr int NT = 256;

struct QArgs {
    const float* in;        // level-1 input
    float* A[kMaxL + 1];    // A[l]: approximation written by level l (input of level l + 1)
    float* det[kMaxL + 1];  // H, V, D of level l, three planes back to back
    int R[kMaxL + 1], Cn[kMaxL + 1];  // input rows / columns of level l
    int tx[kMaxL + 1], ty[kMaxL + 1]; // tile grid of level l
    const unsigned* sched;  // block id -> level << 28 | tile row << 14 | tile column
    unsigned* done;         // [level][tile row] completion counters (monotonic over epochs)
    unsigned epoch;
    int nlevels;
    int* err;               // set when a bounded spin gave up
    int mode;               // experiments (timing only, results wrong): 1 no counter add / poll, 2 no drain + barrier,
                            // 4 plain A stores, 8 plain A loads, 16 counters on lines of their own
};

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// one tile of level l.  COH: inputs are another workgroup's outputs of this launch (sc1 loads), and the A band has a
// consumer in this launch (sc1 stores).
template <bool COH_IN, bool COH_OUT>
__device__ __forceinline__ void tile(const QArgs& a, int l, int by, int bx, float* smem) {
    const int tid = threadIdx.x;
    const float* in = l == 1 ? a.in : a.A[l - 1];
    const int Nr = a.R[l], Nc = a.Cn[l];
    const int xa = 2 * bx * TX - C - PADL, y0 = 2 * by * TY - C;
    constexpr int TOTAL = RY * V4, TRIPS = (TOTAL + NT - 1) / NT;
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, Nr * Nc * 4, 0x00020000);
    f4 st[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int idx = tid + t * NT;
        idx = idx < TOTAL ? idx : TOTAL - 1;
        const int r = idx / V4, g = idx - r * V4;
        const int sy = wrapi(y0 + r, Nr), sx = wrapi(xa + 4 * g, Nc);
        const int off = (sy * Nc + sx) * 4;
        if (COH_IN) st[t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 16));
        else st[t] = *reinterpret_cast<const f4*>(in + (size_t)sy * Nc + sx);
    }
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        int idx = tid + t * NT;
        idx = idx < TOTAL ? idx : TOTAL - 1;
        *reinterpret_cast<f4*>(smem + 4 * idx) = st[t];
    }
    __syncthreads();
    // stand-in arithmetic: band(r, c) = 0.5 (x(2r - 3 + dy, 2c - 3 + dx) + x(2r + 4 - dy', 2c + 4 - dx')) -- both halo ends
    const int t = tid % (TX / 2), r = tid / (TX / 2);  // two adjacent columns, one row
    const float* p = smem + (2 * r) * RXA + PADL + 4 * t;
    const float a0 = 0.5f * (p[0] + p[7 * RXA + 7]), a1 = 0.5f * (p[2] + p[7 * RXA + 9]);
    const float h0 = 0.5f * (p[1] + p[7 * RXA + 6]), h1 = 0.5f * (p[3] + p[7 * RXA + 8]);
    const float v0 = 0.5f * (p[RXA] + p[6 * RXA + 7]), v1 = 0.5f * (p[RXA + 2] + p[6 * RXA + 9]);
    const float d0 = 0.5f * (p[RXA + 1] + p[6 * RXA + 6]), d1 = 0.5f * (p[RXA + 3] + p[6 * RXA + 8]);
    const int Nr2 = Nr / 2, Nc2 = Nc / 2;
    const int oy = by * TY + r, ox = bx * TX + 2 * t;
    const size_t o = (size_t)oy * Nc2 + ox;
    float* det = a.det[l];
    const size_t plane = (size_t)Nr2 * Nc2;
    f2 w;
    w.x = h0; w.y = h1; *reinterpret_cast<f2*>(det + o) = w;
    w.x = v0; w.y = v1; *reinterpret_cast<f2*>(det + plane + o) = w;
    w.x = d0; w.y = d1; *reinterpret_cast<f2*>(det + 2 * plane + o) = w;
    if (COH_OUT) {
        // 16-B write-through stores: even lanes take the odd neighbour's pair (8-B sc1 stores cost 2.7x per byte)
        const float n0 = __shfl_xor(a0, 1), n1 = __shfl_xor(a1, 1);
        if (!(tid & 1)) {
            f4 q; q.x = a0; q.y = a1; q.z = n0; q.w = n1;
            __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.A[l], (short)0, (int)(plane * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), ro, (int)(o * 4), 0, 16);
        }
    } else {
        w.x = a0; w.y = a1; *reinterpret_cast<f2*>(a.A[l] + o) = w;
    }
}

__global__ void __launch_bounds__(NT) level_kernel(QArgs a, int l) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int by = blockIdx.x / a.tx[l], bx = blockIdx.x - by * a.tx[l];
    tile<false, false>(a, l, by, bx, smem);
}

__global__ void __launch_bounds__(NT) queue_kernel(QArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned e = a.sched[blockIdx.x];
    const int l = __builtin_amdgcn_readfirstlane(e >> 28), by = __builtin_amdgcn_readfirstlane((e >> 14) & 0x3fff),
              bx = __builtin_amdgcn_readfirstlane(e & 0x3fff);
    if (l > 1 && (a.mode & 32)) {
        // per-tile flags (plain sc1 stores, no read-modify-write): 16 lanes poll the 4 x 4 producer tiles
        if (threadIdx.x < 16) {
            const int pr = wrapi(2 * by - 1 + (int)(threadIdx.x >> 2), a.ty[l - 1]);
            const int pc = wrapi(2 * bx - 1 + (int)(threadIdx.x & 3), a.tx[l - 1]);
            const unsigned want = a.epoch + 1;
            const unsigned* c = a.done + ((l - 1) * 4096 + pr) * 64 + pc;
            int spins = 0;
            while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (a.mode & 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { *a.err = 1; break; }
            }
        }
        __syncthreads();
    } else if (l > 1) {
        // producers: tile rows 2by-1 .. 2by+2 of level l-1 (periodic), all of their tiles
        if (threadIdx.x < 4) {
            const int pr = wrapi(2 * by - 1 + (int)threadIdx.x, a.ty[l - 1]);
            const unsigned want = (a.epoch + 1) * (unsigned)a.tx[l - 1];
            const unsigned* c = a.done + ((l - 1) * 4096 + pr) * ((a.mode & 16) ? 32 : 1);
            int spins = 0;
            while (!(a.mode & 1) && (int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { *a.err = 1; break; }
            }
        }
        __syncthreads();
    }
    const bool last = l == a.nlevels;
    const bool cin = l > 1 && !(a.mode & 8), cout = !last && !(a.mode & 4);
    if (cin && cout) tile<true, true>(a, l, by, bx, smem);
    else if (cin) tile<true, false>(a, l, by, bx, smem);
    else if (cout) tile<false, true>(a, l, by, bx, smem);
    else tile<false, false>(a, l, by, bx, smem);
    if (!last) {
        if (!(a.mode & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its sc1 stores ...
            __syncthreads();                                   // ... before ONE lane signals for the workgroup
        }
        if (threadIdx.x == 0 && (a.mode & 32))
            __hip_atomic_store(a.done + (l * 4096 + by) * 64 + bx, a.epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (threadIdx.x == 0 && !(a.mode & 1))
            __hip_atomic_fetch_add(a.done + (l * 4096 + by) * ((a.mode & 16) ? 32 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Placement-independent variant: the grid is the level-1 tiles only.  A finished tile adds 1 to the arrival counters of
// the (up to) four tiles of the next level that read it; the workgroup whose add completes a counter (16 arrivals)
// runs that tile itself, right away ("last arriver continues") -- no polling, no spinning, no assumption about the
// dispatch order, so it cannot deadlock.  Counters are monotonic over launches (16 arrivals per launch each).
__global__ void __launch_bounds__(NT) cont_kernel(QArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned s_new[4];
    unsigned stack[12];
    int sp = 0;
    int l = 1, by = blockIdx.x / a.tx[1], bx = blockIdx.x - by * a.tx[1];
    const int stride = (a.mode & 16) ? 32 : 1;  // counters on lines of their own
    for (;;) {
        const bool last = l == a.nlevels;
        if (l > 1 && !last) tile<true, true>(a, l, by, bx, smem);
        else if (l > 1) tile<true, false>(a, l, by, bx, smem);
        else tile<false, true>(a, l, by, bx, smem);
        if (!last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x < 4) {
                const int dby = wrapi(((by - 1) >> 1) + (int)(threadIdx.x >> 1), a.ty[l + 1]);
                const int dbx = wrapi(((bx - 1) >> 1) + (int)(threadIdx.x & 1), a.tx[l + 1]);
                const unsigned old = __hip_atomic_fetch_add(a.done + ((size_t)((l + 1) * 4096 + dby) * 64 + dbx) * stride, 1u,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_new[threadIdx.x] = ((old + 1) & 15) == 0 ? ((unsigned)(l + 1) << 28 | (unsigned)dby << 14 | (unsigned)dbx) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned e = s_new[i];
                if (e && sp < 12) stack[sp++] = e;
            }
        }
        if (sp == 0) break;
        const unsigned e = stack[--sp];
        l = e >> 28; by = (e >> 14) & 0x3fff; bx = e & 0x3fff;
        __syncthreads();  // s_new and the LDS tile are reused
    }
}

__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i ^ seed;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        p[i] = (float)(x >> 24);
    }
}
__global__ void compare(const float* a, const float* b, size_t n, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

int main(int argc, char** argv) {
    const int lag = argc > 1 ? atoi(argv[1]) : 8;     // tile rows of level l a dependent row of level l+1 trails by
    const int L = argc > 2 ? atoi(argv[2]) : 4;
    const int N = argc > 3 ? atoi(argv[3]) : 4096;
    const int rot = argc > 4 ? atoi(argv[4]) : 8;     // level-1 tile rows the walk starts before row 0 (periodic wrap)
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    QArgs a{}, ref{};
    a.nlevels = ref.nlevels = L;
    size_t n0 = (size_t)N * N;
    float* in;
    CK(hipMalloc(&in, n0 * 4));
    a.in = ref.in = in;
    int total = 0;
    for (int l = 1; l <= L; l++) {
        a.R[l] = ref.R[l] = N >> (l - 1);
        a.Cn[l] = ref.Cn[l] = N >> (l - 1);
        a.ty[l] = ref.ty[l] = (a.R[l] / 2) / TY;
        a.tx[l] = ref.tx[l] = (a.Cn[l] / 2) / TX;
        size_t plane = (size_t)(a.R[l] / 2) * (a.Cn[l] / 2);
        CK(hipMalloc(&a.A[l], plane * 4)); CK(hipMalloc(&a.det[l], 3 * plane * 4));
        CK(hipMalloc(&ref.A[l], plane * 4)); CK(hipMalloc(&ref.det[l], 3 * plane * 4));
        total += a.tx[l] * a.ty[l];
        if (a.tx[l] < 1 || a.ty[l] < 4) { printf("level %d too small\n", l); return 1; }
    }
    // ---- schedule: level-1 tile rows in rotated order; a row of level l+1 becomes ready when its four producer rows
    // have been emitted and is emitted `lag` level-1 rows (scaled by the level) later
    std::vector<unsigned> sched;
    {
        std::vector<std::vector<int>> emitted_at(L + 2);  // emission time (in emitted tiles) of every row, -1 = not yet
        for (int l = 1; l <= L; l++) emitted_at[l].assign(a.ty[l], -1);
        struct Pend { int l, row; long long when; };
        std::vector<Pend> pend;
        auto emit_row = [&](int l, int row) {
            for (int x = 0; x < a.tx[l]; x++) sched.push_back((unsigned)l << 28 | (unsigned)row << 14 | (unsigned)x);
            emitted_at[l][row] = (int)sched.size();
        };
        auto ready_rows = [&](int l) {  // rows of level l+1 whose producers are all emitted and that are not queued yet
            if (l >= L) return;
            for (int r = 0; r < a.ty[l + 1]; r++) {
                if (emitted_at[l + 1][r] != -1) continue;
                bool ok = true;
                for (int k = -1; k <= 2 && ok; k++) ok = emitted_at[l][((2 * r + k) % a.ty[l] + a.ty[l]) % a.ty[l]] >= 0;
                if (ok) {
                    emitted_at[l + 1][r] = -2;  // queued
                    pend.push_back({l + 1, r, (long long)sched.size() + (long long)lag * a.tx[1]});
                }
            }
        };
        auto flush = [&](bool all) {
            bool again = true;
            while (again) {
                again = false;
                for (size_t i = 0; i < pend.size(); i++)
                    if (all || pend[i].when <= (long long)sched.size()) {
                        Pend p = pend[i];
                        pend.erase(pend.begin() + i);
                        emit_row(p.l, p.row);
                        ready_rows(p.l);
                        again = true;
                        break;
                    }
            }
        };
        for (int k = 0; k < a.ty[1]; k++) {
            emit_row(1, ((k - rot) % a.ty[1] + a.ty[1]) % a.ty[1]);
            ready_rows(1);
            flush(false);
        }
        flush(true);
        if ((int)sched.size() != total) { printf("schedule %zu != %d tiles\n", sched.size(), total); return 1; }
        // check: every producer row of every tile is complete at a smaller block id
        std::vector<std::vector<int>> last_of(L + 2);
        for (int l = 1; l <= L; l++) last_of[l].assign(a.ty[l], -1);
        for (int i = 0; i < total; i++) last_of[sched[i] >> 28][(sched[i] >> 14) & 0x3fff] = i;
        for (int i = 0; i < total; i++) {
            const int l = sched[i] >> 28, r = (sched[i] >> 14) & 0x3fff;
            if (l > 1)
                for (int k = -1; k <= 2; k++)
                    if (last_of[l - 1][((2 * r + k) % a.ty[l - 1] + a.ty[l - 1]) % a.ty[l - 1]] >= i) { printf("schedule order violated\n"); return 1; }
        }
    }
    unsigned* d_sched; unsigned* d_done; int* d_err; unsigned long long* d_bad;
    CK(hipMalloc(&d_sched, total * 4)); CK(hipMemcpy(d_sched, sched.data(), total * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_done, (size_t)(kMaxL + 1) * 4096 * 4 * 64 * 32)); CK(hipMemset(d_done, 0, (size_t)(kMaxL + 1) * 4096 * 4 * 64 * 32));
    CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    CK(hipMalloc(&d_bad, 8));
    a.sched = d_sched; a.done = d_done; a.err = d_err; a.mode = mode;
    const size_t lds = (size_t)RY * RXA * 4;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned epoch = 0;
    auto run_sep = [&](QArgs& g) {
        for (int l = 1; l <= L; l++) hipLaunchKernelGGL(level_kernel, dim3(g.tx[l] * g.ty[l]), dim3(NT), lds, s, g, l);
    };
    auto run_queue = [&]() {
        a.epoch = epoch++;
        if (mode & 128) hipLaunchKernelGGL(cont_kernel, dim3(a.tx[1] * a.ty[1]), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL(queue_kernel, dim3(total), dim3(NT), lds, s, a);
    };
    // ---- correctness: several epochs with different inputs, every word of every band
    unsigned long long bad_total = 0;
    for (int it = 0; it < 6; it++) {
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, in, n0, 1234u + it);
        run_sep(ref);
        run_queue();
        for (int rep = 0; rep < (it == 3 ? 20 : 1); rep++) run_queue();  // back-to-back replays on warm caches
        CK(hipMemsetAsync(d_bad, 0, 8, s));
        for (int l = 1; l <= L; l++) {
            size_t plane = (size_t)(a.R[l] / 2) * (a.Cn[l] / 2);
            hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, s, a.A[l], ref.A[l], plane, d_bad);
            hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, s, a.det[l], ref.det[l], 3 * plane, d_bad);
        }
        unsigned long long bad = 0; int err = 0;
        CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        bad_total += bad;
        printf("epoch set %d: mismatching words %llu, spin timeouts %d\n", it, bad, err);
    }
    // ---- timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto&& fn) {
        for (int i = 0; i < 200; i++) fn();  // pre-heat
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 50; i++) fn();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 50); sum += ms / 50;
        }
        printf("%-34s %8.2f us (best of 5; mean %.2f)\n", name, best * 1e3, sum / 5 * 1e3);
    };
    printf("N %d, levels %d, tiles %d, lag %d rows, rotation %d rows, mode %d\n", N, L, total, lag, rot, mode);
    time_it("launch per level", [&] { run_sep(ref); });
    time_it("one launch, schedule + counters", [&] { run_queue(); });
    for (int l = 1; l <= L; l++) {
        char nm[64]; snprintf(nm, sizeof nm, "level %d alone", l);
        time_it(nm, [&] { hipLaunchKernelGGL(level_kernel, dim3(ref.tx[l] * ref.ty[l]), dim3(NT), lds, s, ref, l); });
    }
    int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    printf("total mismatches %llu, spin timeouts %d\n", bad_total, err);
    return bad_total || err ? 2 : 0;
}
