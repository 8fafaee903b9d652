from libc.stdint cimport intptr_t, uint32_t

cdef extern from "pypwt_amd.h":
    ctypedef struct pdwt_plan
    ctypedef pdwt_plan* pdwt_handle

    ctypedef struct pdwt_info:          # == w_info (pdwt/src/utils.h:9-19)
        int ndims
        int Nr
        int Nc
        int nlevels
        int do_swt
        int hlen

    # construction / destruction            replaces
    int pdwt_create(const float* img, int Nr, int Nc, const char* wname, int levels,
                    int mem_is_on_host, int do_separable, int do_cycle_spinning,
                    int do_swt, int ndim, pdwt_handle* out)      # C_Wavelets(float*, int, ...)  pypwt.pyx:42
    int pdwt_clone(pdwt_handle src, pdwt_handle* out)            # C_Wavelets(C_Wavelets)        pypwt.pyx:43
    int pdwt_destroy(pdwt_handle h)                              # del self.w                    pypwt.pyx:604
    # transforms
    int pdwt_forward(pdwt_handle h)                              # forward()                     pypwt.pyx:44
    int pdwt_inverse(pdwt_handle h)                              # inverse()                     pypwt.pyx:49
    # coefficient operators
    int pdwt_soft_threshold(pdwt_handle h, float beta, int do_app, int normalize)   # pypwt.pyx:45
    int pdwt_hard_threshold(pdwt_handle h, float beta, int do_app, int normalize)   # pypwt.pyx:46
    int pdwt_group_soft_threshold(pdwt_handle h, float beta, int do_app, int normalize)   # C++ only in the reference (wt.cu:329-336)
    int pdwt_shrink(pdwt_handle h, float beta, int do_app)                          # pypwt.pyx:47
    int pdwt_proj_linf(pdwt_handle h, float beta, int do_app)                       # C++ only in the reference (wt.cu:349-356)
    int pdwt_circshift(pdwt_handle h, int sr, int sc, int inplace)                  # pypwt.pyx:48
    int pdwt_norm2sq(pdwt_handle h, float* out)                                     # pypwt.pyx:50
    int pdwt_norm1(pdwt_handle h, float* out)                                       # pypwt.pyx:51
    int pdwt_add_wavelet(pdwt_handle dst, pdwt_handle src, float alpha)             # pypwt.pyx:57
    # round 6: the norms without the round trip -- two doubles {sum |c|, sum c^2} left in device memory, nothing synchronised
    int pdwt_norms_async(pdwt_handle h, double* d_out2)
    int pdwt_norms_slot(pdwt_handle h, double** d_ptr)
    int pdwt_soft_threshold_norms_async(pdwt_handle h, float beta, int do_app, int normalize, double* d_out2)
    # data movement
    long long pdwt_get_image(pdwt_handle h, float* dst)                             # pypwt.pyx:52
    long long pdwt_get_coeff(pdwt_handle h, float* dst, int num)                    # pypwt.pyx:54
    int pdwt_set_image(pdwt_handle h, const float* src, int mem_is_on_device)       # pypwt.pyx:55
    int pdwt_set_coeff(pdwt_handle h, const float* src, int num, int on_device)     # pypwt.pyx:56
    intptr_t pdwt_image_ptr(pdwt_handle h)                                          # pypwt.pyx:58
    intptr_t pdwt_coeff_ptr(pdwt_handle h, int num)                                 # pypwt.pyx:59
    long long pdwt_coeff_count(pdwt_handle h, int num, int* rows, int* cols)        # elements of band `num` (the reference copies blindly, wt.cu:435)
    # all bands in one device-to-host copy (the bands sit back to back in `num` order, each padded to 64 elements)
    long long pdwt_coeff_region(pdwt_handle h, long long* band_offsets, int capacity)   # layout; returns the region length
    long long pdwt_get_coeff_region(pdwt_handle h, float* dst)                          # replaces the loop of pypwt.pyx:287-305
    # custom filters
    int pdwt_set_filters_forward(pdwt_handle h, const char* name, unsigned int len,
                                 const float* f1, const float* f2,
                                 const float* f3, const float* f4)                  # pypwt.pyx:60
    int pdwt_set_filters_inverse(pdwt_handle h, const float* f1, const float* f2,
                                 const float* f3, const float* f4)                  # pypwt.pyx:61
    # introspection (replaces direct reads of w.winfos / w.do_separable / w.state, pypwt.pyx:33-36)
    int pdwt_get_info(pdwt_handle h, pdwt_info* info, int* do_separable,
                      int* do_cycle_spinning, int* state, int* batch)
    int pdwt_print_info(pdwt_handle h)                                              # pypwt.pyx:53
    int pdwt_info_string(pdwt_handle h, char* buf, size_t n)                        # the same text into a buffer
    int pdwt_current_shift(pdwt_handle h, int* sr, int* sc)                         # Wavelets::current_shift_r/c (wt.h:27-28)
    const char* pdwt_version()
    const char* pdwt_last_error()
    int pdwt_synchronize(pdwt_handle h)
    void* pdwt_get_stream(pdwt_handle h)
    int pdwt_device(pdwt_handle h)
    # device-memory sources: order the copy after the stream that produced them (no reference counterpart:
    # the reference runs everything on the legacy default stream, wt.cu:117-126,425-466)
    int pdwt_wait_for_stream(pdwt_handle h, void* producer_stream)
    int pdwt_sync_producer(int device_id, void* producer_stream, int whole_device)
    # round 4: which device holds a pointer / how many devices (one process over all GPUs of a node: pypwt_amd.ShardedBatch)
    int pdwt_bind_image(pdwt_handle h, void* device_ptr)      # round 4: the plan's image lives in caller-owned device memory (NULL unbinds)
    int pdwt_copy(pdwt_handle h, void* dst, const void* src, long long count, int kind)  # round 5: row ranges of plan buffers, on the plan's stream (0 d2d, 1 h2d, 2 d2h)
    int pdwt_device_count()
    int pdwt_device_of_pointer(const void* p)

    # round 4: neighbour exchange for ONE image tiled over several GPUs, straight on RCCL (librccl is dlopen'ed on first
    # use).  No reference counterpart: the reference is single-GPU (pdwt/TODO.txt:15).
    ctypedef struct pdwt_comm
    ctypedef pdwt_comm* pdwt_comm_handle
    int pdwt_comm_unique_id(void* id128)                                  # rank 0; hand the 128 bytes to the other ranks
    int pdwt_comm_create(const void* id128, int nranks, int rank, int device_id, pdwt_comm_handle* out)
    int pdwt_comm_destroy(pdwt_comm_handle c)
    int pdwt_comm_rank(pdwt_comm_handle c)
    int pdwt_comm_size(pdwt_comm_handle c)
    int pdwt_comm_exchange(pdwt_comm_handle c, int n,
                           const void* const* send_ptr, const long long* send_count, const int* send_peer,
                           void* const* recv_ptr, const long long* recv_count, const int* recv_peer,
                           void* hip_stream)                              # ONE ncclGroupStart .. End on that stream
    int pdwt_comm_all_gather(pdwt_comm_handle c, const void* send, void* recv, long long count_per_rank, void* hip_stream)
    int pdwt_comm_broadcast(pdwt_comm_handle c, void* buf, long long count, int root, void* hip_stream)
    const char* pdwt_comm_last_error()
