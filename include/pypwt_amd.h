/*
 * pypwt_amd.h -- C ABI of the MI355X-native wavelet transform library
 *                (libpypwt_amd.so, gfx950 HIP kernels).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference's
 * Cython module binds a C++ class directly (`cdef extern from "../pdwt/src/wt.h"`,
 * reference src/pypwt.pyx:8-61; class Wavelets, reference pdwt/src/wt.h:20-76).
 * Every method the Cython shim declares has exactly one entry point here, taking
 * an opaque handle instead of `this`; plain pointers and sizes only, no C++ or
 * torch types.  INTEGRATION.md shows the `cdef extern` block a pypwt maintainer
 * would put in place of the C++ one.
 *
 * Conventions
 *   - every function returns an int status: 0 (PDWT_OK) or a negative pdwt_status,
 *     except the getters documented as returning an element count (like the
 *     reference's get_image/get_coeff, pdwt/src/wt.cu:419-422, :473-506) and the
 *     raw-pointer getters.  pdwt_last_error() gives the message of the last
 *     failure on the calling thread.
 *   - the plan owns all device memory; host buffers belong to the caller; device
 *     pointers returned by pdwt_image_ptr / pdwt_coeff_ptr are borrowed and valid
 *     while the plan lives (pdwt/src/wt.cu:658-665).
 *   - kernels are enqueued asynchronously on the plan's HIP stream; device->host
 *     getters synchronise that stream (the reference relies on blocking
 *     cudaMemcpy on the default stream for the same effect).
 *   - plans are independent: filter taps are per plan (the reference keeps them
 *     in process-global __constant__ memory, pdwt/src/common.h:28-36), so plans
 *     with different wavelets, devices and streams can coexist.  One plan must
 *     not be used from two threads at once.
 *   - this header is the CONTRACT: what a binding of the reference's class needs.  Measurement and test hooks (per-launch
 *     timing, the level / copy micro-benchmarks, the dispatch knobs, the synthetic input) are in pypwt_amd_bench.h.
 *   - coefficient index `num` (pdwt/src/wt.cu:479-502):
 *       2D: 0 = A_L, 1 + 3(l-1) + {0,1,2} = H_l, V_l, D_l   (level 1 = finest)
 *       1D: 0 = A_L, l = D_l
 *     A batched plan (batch > 1) stores every band as [batch][rows][cols].
 */
#ifndef PYPWT_AMD_H
#define PYPWT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdwt_plan* pdwt_handle;

/* Sample / coefficient / scalar type of the library.  libpypwt_amd.so is the fp32 build (the reference's
 * default, and the only type its Python class accepts); libpypwt_amd_f64.so is the same source built with
 * -DPDWT_DOUBLE (the reference's DOUBLEPRECISION make switch, pdwt/src/filters.h:16-30) and exports the
 * same symbols with pdwt_real = double.  Compile a client of the fp64 library with -DPDWT_DOUBLE. */
#ifdef PDWT_DOUBLE
typedef double pdwt_real;
#else
typedef float pdwt_real;
#endif

/* mirrors struct w_info (reference pdwt/src/utils.h:9-19) */
typedef struct pdwt_info {
    int ndims;   /* 1 or 2 (a 2D array with ndims == 1 is a batched 1D transform) */
    int Nr;      /* rows (1 for plain 1D) */
    int Nc;      /* columns */
    int nlevels; /* after clamping, pdwt/src/wt.cu:155-165 */
    int do_swt;  /* stationary (undecimated) transform */
    int hlen;    /* filter length */
} pdwt_info;

/* mirrors enum w_state (reference pdwt/src/wt.h:8-17) */
typedef enum pdwt_state {
    PDWT_INIT = 0,
    PDWT_FORWARD = 1,
    PDWT_INVERSE = 2,
    PDWT_THRESHOLD = 3,
    PDWT_CREATION_ERROR = 4,
    PDWT_FORWARD_ERROR = 5,
    PDWT_INVERSE_ERROR = 6,
    PDWT_THRESHOLD_ERROR = 7
} pdwt_state;

typedef enum pdwt_status {
    PDWT_OK = 0,
    PDWT_ERR_ARG = -1,          /* bad argument (null pointer, bad index, size) */
    PDWT_ERR_WAVELET = -2,      /* unknown wavelet name (reference returns -2, separable.cu:42-45) */
    PDWT_ERR_HIP = -3,          /* a HIP runtime call failed (reference -3, separable.cu:57-73) */
    PDWT_ERR_STATE = -4,        /* refused by the state machine (e.g. coefficients after inverse) */
    PDWT_ERR_FILTER_LEN = -5,   /* custom filter longer than PDWT_MAX_FILTER_WIDTH (reference -1) */
    PDWT_ERR_MISMATCH = -6,     /* add_wavelet operands differ */
    PDWT_ERR_UNSUPPORTED = -7,
    PDWT_ERR_NOMEM = -8
} pdwt_status;

#define PDWT_MAX_FILTER_WIDTH 40 /* reference pdwt/src/common.h:15 */

/* ---- construction (replaces Wavelets::Wavelets, pdwt/src/wt.cu:84-185, wt.h:42) ----
 * img: Nr*Nc pdwt_real, row-major; host memory if mem_is_on_host else device memory;
 * NULL = zero image.  Unknown wname -> PDWT_ERR_WAVELET (the reference stores -2 and
 * later hangs in w_ilog2).  levels is clamped as the reference does.
 * Uses the current HIP device and a private stream. */
int pdwt_create(const pdwt_real* img, int Nr, int Nc, const char* wname, int levels, int mem_is_on_host,
                int do_separable, int do_cycle_spinning, int do_swt, int ndim, pdwt_handle* out);

/* NEW (not in the reference): explicit device, caller stream (NULL = private stream)
 * and a batch of `batch` independent images [batch][Nr][Nc] transformed by every call. */
int pdwt_create_batched(const pdwt_real* img, int batch, int Nr, int Nc, const char* wname, int levels,
                        int mem_is_on_host, int do_separable, int do_cycle_spinning, int do_swt, int ndim,
                        int device_id, void* hip_stream, pdwt_handle* out);

/* deep copy (replaces the copy constructor, pdwt/src/wt.cu:191-222) */
int pdwt_clone(pdwt_handle src, pdwt_handle* out);
int pdwt_destroy(pdwt_handle h); /* ~Wavelets, pdwt/src/wt.cu:226-233 */

/* ---- transforms (Wavelets::forward / inverse, pdwt/src/wt.cu:236-305) ---- */
int pdwt_forward(pdwt_handle h);
int pdwt_inverse(pdwt_handle h); /* second call in a row: PDWT_ERR_STATE, nothing done */

/* ---- coefficient operators (pdwt/src/wt.cu:308-356, common.cu:219-371) ---- */
int pdwt_soft_threshold(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs, int normalize);
int pdwt_hard_threshold(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs, int normalize);
int pdwt_group_soft_threshold(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs, int normalize);
int pdwt_shrink(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs);
int pdwt_proj_linf(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs);
int pdwt_circshift(pdwt_handle h, int sr, int sc, int inplace); /* wt.cu:364-366 */
int pdwt_norm1(pdwt_handle h, pdwt_real* out);   /* wt.cu:396-416 */
int pdwt_norm2sq(pdwt_handle h, pdwt_real* out); /* wt.cu:368-393 (1D bug at :387 not reproduced) */
/* NEW (round 6): the same two sums WITHOUT the round trip.  The results -- {sum |c|, sum c^2} as two doubles -- are written
 * to DEVICE memory on the plan's stream (d_out2, or the plan's own slot when NULL: pdwt_norms_slot) and nothing is
 * synchronised; pdwt_norm1 / pdwt_norm2sq above stay blocking host getters like the reference's (wt.cu:368-416).
 * pdwt_soft_threshold_norms_async = pdwt_soft_threshold followed by pdwt_norms_async in ONE sweep over the coefficients
 * (the inner loop of iterative shrinkage, pdwt/README.md:6-7): same arguments and state rules as pdwt_soft_threshold. */
int pdwt_norms_async(pdwt_handle h, double* d_out2);
int pdwt_norms_slot(pdwt_handle h, double** d_ptr); /* device address of the plan's own result slot (valid while the plan lives) */
int pdwt_soft_threshold_norms_async(pdwt_handle h, pdwt_real beta, int do_thresh_appcoeffs, int normalize, double* d_out2);
/* dst += alpha * src ; returns 0, or the reference's codes -1..-4 / +1 (wt.cu:622-655) */
int pdwt_add_wavelet(pdwt_handle dst, pdwt_handle src, pdwt_real alpha);

/* ---- data movement (pdwt/src/wt.cu:419-506) ---- */
long long pdwt_get_image(pdwt_handle h, pdwt_real* dst);          /* returns element count, <0 on error */
long long pdwt_get_coeff(pdwt_handle h, pdwt_real* dst, int num); /* 0 if refused after inverse */
/* In place: with mem_is_on_device = 1 and src == pdwt_image_ptr(h) (pdwt_coeff_ptr(h, num)) nothing is copied and the
 * host does not block -- the caller has written the buffer itself (on the plan's stream, or ordered with it by
 * pdwt_wait_for_stream) and the call only makes the image (the coefficients) current. */
int pdwt_set_image(pdwt_handle h, const pdwt_real* src, int mem_is_on_device);
int pdwt_set_coeff(pdwt_handle h, const pdwt_real* src, int num, int mem_is_on_device);
long long pdwt_coeff_count(pdwt_handle h, int num, int* rows, int* cols); /* elements incl. batch */
/* NEW: ALL coefficient bands in ONE device-to-host copy.  The reference's `coeffs` is 3 L + 1 blocking cudaMemcpy calls
 * (pypwt.pyx:287-305, wt.cu:473-506): 28 round trips for a 512^2 haar transform, which dominate its own benchmark method
 * (test/benchmark.py:141-162).  The plan keeps its bands back to back in `num` order, each padded to a multiple of 64
 * elements: pdwt_coeff_region writes the offset of every band (elements from the start of the region) into band_offsets
 * (capacity entries) and returns the region's length; pdwt_get_coeff_region copies the whole region into dst (that many
 * elements) and returns the count, 0 when refused after inverse() like pdwt_get_coeff. */
long long pdwt_coeff_region(pdwt_handle h, long long* band_offsets, int capacity);
long long pdwt_get_coeff_region(pdwt_handle h, pdwt_real* dst);
/* NEW (batched plans): one image / one image's sub-band of a batch, so that a 128-image shard can be
 * inspected without a host buffer for the whole batch; same refusal rule as pdwt_get_coeff (wt.cu:473-477) */
long long pdwt_get_image_at(pdwt_handle h, pdwt_real* dst, int image_index);
long long pdwt_get_coeff_at(pdwt_handle h, pdwt_real* dst, int num, int image_index);
intptr_t pdwt_image_ptr(pdwt_handle h);                                   /* wt.cu:658-660 */
intptr_t pdwt_coeff_ptr(pdwt_handle h, int num);                          /* wt.cu:663-665 */

/* ---- custom filter banks (pdwt/src/wt.cu:558-600) ----
 * separable plan: filter1 = low-pass, filter2 = high-pass (filter3/4 ignored)
 * non-separable : filter1..4 = LL, LH, HL, HH, each len*len row-major */
int pdwt_set_filters_forward(pdwt_handle h, const char* name, unsigned int len, const pdwt_real* filter1,
                             const pdwt_real* filter2, const pdwt_real* filter3, const pdwt_real* filter4);
int pdwt_set_filters_inverse(pdwt_handle h, const pdwt_real* filter1, const pdwt_real* filter2, const pdwt_real* filter3,
                             const pdwt_real* filter4);

/* ---- introspection ---- */
int pdwt_get_info(pdwt_handle h, pdwt_info* info, int* do_separable, int* do_cycle_spinning, int* state,
                  int* batch);
int pdwt_print_info(pdwt_handle h);                       /* print_informations, wt.cu:511-550 */
/* NEW: destroyed plans hand their device memory and stream to a small process-wide pool that new plans draw from:
 * creating + destroying a plan costs 3.4 ms of hipMalloc / hipFree / stream calls otherwise, twenty times the transform of
 * a 512^2 image (the reference's tests and tutorials build one Wavelets object per image).  Only small blocks are kept: at
 * most PDWT_POOL_BLOCK_MB (default 64 MiB) each and PDWT_POOL_MB (default 256 MiB) in total, 8 streams -- a large plan's
 * memory goes back to the driver when it is destroyed.  The library releases the pool by itself when one of its own
 * allocations fails; pdwt_trim_pool() releases everything the pool holds (call it before another allocator in the process
 * needs the memory); returns the number of blocks freed. */
int pdwt_trim_pool(void);
int pdwt_info_string(pdwt_handle h, char* buf, size_t n); /* same text into a buffer */
int pdwt_current_shift(pdwt_handle h, int* sr, int* sc);
const char* pdwt_last_error(void);
const char* pdwt_version(void);

/* ---- wavelet table ---- */
int pdwt_wavelet_count(void);
const char* pdwt_wavelet_name(int index);
/* banks: 4*hlen floats = dec_lo, dec_hi, rec_lo, rec_hi ; returns hlen or PDWT_ERR_WAVELET */
int pdwt_wavelet_filters(const char* wname, pdwt_real* banks, int capacity);

/* ---- stream / device plumbing (NEW) ---- */
int pdwt_synchronize(pdwt_handle h);
int pdwt_set_stream(pdwt_handle h, void* hip_stream); /* borrow a caller stream (e.g. torch's) */
void* pdwt_get_stream(pdwt_handle h);
int pdwt_device(pdwt_handle h);
int pdwt_device_count(void); /* HIP devices visible to the process (0: none -- every pdwt_create will fail, there is no CPU path) */
/* Ordering a DEVICE-memory source with the code that produced it (the reference runs everything on the legacy
 * default stream, so its cudaMemcpy DtoD in wt.cu:117-126,425-466 is ordered for free; a plan here owns a
 * non-blocking stream).  pdwt_set_image / pdwt_set_coeff with mem_is_on_device = 1 and pdwt_create with
 * mem_is_on_host = 0 copy on the plan's stream and return when the copy has finished (the source may then be
 * reused or freed); what they cannot know is which stream WROTE the source:
 *   pdwt_wait_for_stream(h, s)   everything submitted to the plan's stream from now on starts after the work
 *                                already submitted to stream s (NULL = the legacy default stream): an event
 *                                recorded on s and waited for on the plan's stream, the host does not block;
 *   pdwt_sync_producer(dev, s, whole_device)   host-blocking form for use before a plan exists:
 *                                hipStreamSynchronize(s), or hipDeviceSynchronize() when whole_device != 0
 *                                (the producer's stream is unknown: a __cuda_array_interface__ without "stream").
 *   pdwt_device_of_pointer(p)    the device that owns a device allocation (hipPointerGetAttributes), so that a source
 *                                with an unknown producer stream is ordered by synchronising ITS device, which need
 *                                not be the current one or the plan's; negative status for host / unknown addresses. */
int pdwt_wait_for_stream(pdwt_handle h, void* producer_stream);
int pdwt_device_of_pointer(const void* device_ptr);
int pdwt_sync_producer(int device_id, void* producer_stream, int whole_device);
/* Bind the plan's IMAGE to device memory the caller owns: from now on forward() reads its input there and inverse()
 * writes its reconstruction there (batch x Nr x Nc elements, row-major, on the plan's device; 16-byte aligned -- PDWT_ERR_ARG
 * otherwise: the tuned kernels stage the image with 16-byte accesses); pdwt_image_ptr returns it; the plan's own image buffer is
 * unused.  NULL unbinds.  pdwt_clone of a bound plan copies the image: the clone owns its copy and is not bound.  The memory must outlive the binding; nothing is copied.  Use: chaining plans without copies -- one level's
 * approximation band (pdwt_coeff_ptr(prev, 0)) IS the next level plan's image (pypwt_amd/tiled.py keeps the row slabs of an
 * image tiled over several GPUs that way).  No reference counterpart (the reference owns all its buffers, wt.cu:527-539).
 * Synchronises the plan's stream. */
int pdwt_bind_image(pdwt_handle h, void* device_ptr);
/* Copy `count` elements (of pdwt_real) between two buffers ON THE PLAN'S STREAM, ordered with its launches.  kind 0: device ->
 * device, enqueued -- the call returns at once; 1: host -> device and 2: device -> host return when the copy has finished.
 * For ROW RANGES of a plan's buffers (pdwt_image_ptr / pdwt_coeff_ptr + an offset): the slab of a rank, the halo rows of a
 * tiled image whose ring closes on the rank itself, the gathered approximation (pypwt_amd/tiled.py needs nothing else from a
 * device runtime -- no torch, no cupy).  The reference copies whole buffers only (cudaMemcpy in pdwt/src/wt.cu:425-466,470-520). */
int pdwt_copy(pdwt_handle h, void* dst, const void* src, long long count, int kind);

/* ---- NEW: neighbour exchange for ONE image tiled over several GPUs (the reference has no multi-GPU code,
 * pdwt/TODO.txt:15).  A communicator wraps an RCCL communicator (librccl is dlopen'ed on first use: single-GPU callers never
 * touch it).  One process per GPU: rank 0 calls pdwt_comm_unique_id and hands the PDWT_COMM_ID_BYTES bytes to the other
 * ranks by any means (a socket, a file, MPI: pypwt_amd/comm.py uses TCP), then every rank calls pdwt_comm_create.  All
 * transfers are enqueued on the given HIP stream -- the plan's, so that they are ordered with its level kernels -- and do
 * not block the host.  Counts are in pdwt_real values.  pdwt_comm_last_error() has the message of the last failure on the
 * calling thread; PDWT_ERR_UNSUPPORTED: librccl could not be loaded. */
typedef struct pdwt_comm* pdwt_comm_handle;
#define PDWT_COMM_ID_BYTES 128
int pdwt_comm_unique_id(void* id);
int pdwt_comm_create(const void* id, int nranks, int rank, int device_id, pdwt_comm_handle* out);
int pdwt_comm_destroy(pdwt_comm_handle c);
int pdwt_comm_rank(pdwt_comm_handle c);
int pdwt_comm_size(pdwt_comm_handle c);
/* ONE grouped point-to-point exchange (ncclGroupStart / Send / Recv / GroupEnd): message i sends send_count[i] values from
 * send_ptr[i] to rank send_peer[i] and receives recv_count[i] values into recv_ptr[i] from rank recv_peer[i]; a null
 * pointer or a zero count skips that half.  The halo rows of every band of a level are one call: neighbour traffic only
 * (xGMI links), no collective.  A rank may name itself (a ring of one: the periodic image). */
int pdwt_comm_exchange(pdwt_comm_handle c, int n, const void* const* send_ptr, const long long* send_count, const int* send_peer,
                       void* const* recv_ptr, const long long* recv_count, const int* recv_peer, void* hip_stream);
/* the two collectives of the gather step (the approximation that has become thinner than the halo goes to rank 0 and comes
 * back): recv holds size x count_per_rank values; broadcast in place from `root` */
int pdwt_comm_all_gather(pdwt_comm_handle c, const void* send, void* recv, long long count_per_rank, void* hip_stream);
int pdwt_comm_broadcast(pdwt_comm_handle c, void* buf, long long count, int root, void* hip_stream);
const char* pdwt_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PYPWT_AMD_H */
