// launch_util.hpp -- small helpers shared by the launch_*.hip translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <atomic>

#include "strip_walk.hpp"  // strip_walk_seg

namespace pdwt {

// Environment knobs.  The PRODUCT libraries read seven documented variables with plain getenv (the pool
// limits, the RCCL library path and four "no fused launches" switches that tests and bug hunts use).  Everything else -- tile
// shapes, thresholds, segment lengths of the A/B measurements -- goes through lab_env and exists only in the measurement
// build libpypwt_amd_lab.so (-DPDWT_LAB_KERNELS): the product returns "unset" without looking, its dispatch is what the
// sources say.  (Dispatch thresholds a caller may want to move at run time are pdwt_set_tuning keys, pypwt_amd_bench.h.)
#ifdef PDWT_LAB_KERNELS
static inline const char* lab_env(const char* name) { return getenv(name); }
#else
static inline const char* lab_env(const char*) { return nullptr; }
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// resident workgroups of a kernel on the whole chip (queried once per instantiation and device)
template <typename K>
static inline int resident_slots(K kernel, int nt, size_t lds_bytes, std::atomic<int>* cache) {
    int v = cache->load(std::memory_order_relaxed);
    if (v > 0) return v;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nt, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    v = per_cu * (cus > 0 ? cus : 256);
    cache->store(v, std::memory_order_relaxed);
    return v;
}

// Kernels that stage more than 64 KiB of dynamic LDS (long filters) must opt in once per
// device; gfx950 has 160 KiB per CU.  The per-device flags are atomics: two host threads driving two
// devices may race to set one (setting the attribute twice is harmless).
template <typename K>
static inline hipError_t allow_big_lds(K kernel, size_t bytes, std::atomic<bool>* done_per_device) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) dev = 0;
    if (done_per_device[dev].load(std::memory_order_relaxed)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
    if (e == hipSuccess) done_per_device[dev].store(true, std::memory_order_relaxed);
    return e;
}

// grid for a grid-stride streaming kernel: enough workgroups to fill 256 CUs x 8
static inline int stream_grid(long long work_items, int block) {
    long long g = cdivll(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace pdwt
