// runtime.hip -- status strings, device queries, memory, streams and events of the C-ABI.
// This is what cusp::device_memory containers sit on (replaces thrust::device_malloc_allocator,
// reference cusp/detail/memory.inl:28-35, and the cudaEvent timer of performance/timer.h:23-54).
#include "common.h"
#include <cstdarg>

namespace cmi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace cmi

using namespace cmi;

CMI_API const char *cmi_status_string(int status)
{
    switch (status) {
    case CMI_SUCCESS: return "CMI_SUCCESS";
    case CMI_ERROR_INVALID_VALUE: return "CMI_ERROR_INVALID_VALUE";
    case CMI_ERROR_HIP: return "CMI_ERROR_HIP";
    case CMI_ERROR_NOT_SUPPORTED: return "CMI_ERROR_NOT_SUPPORTED";
    case CMI_ERROR_NO_DEVICE: return "CMI_ERROR_NO_DEVICE";
    case CMI_ERROR_ALLOC: return "CMI_ERROR_ALLOC";
    case CMI_ERROR_IO: return "CMI_ERROR_IO";
    case CMI_ERROR_COMM: return "CMI_ERROR_COMM";
    default: return "CMI_ERROR_UNKNOWN";
    }
}

CMI_API const char *cmi_last_error(void) { return g_err; }
CMI_API int cmi_version(void) { return CMI_VERSION; }

CMI_API int cmi_device_count(int *count)
{
    if (!count) return fail(CMI_ERROR_INVALID_VALUE, "cmi_device_count: null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return CMI_SUCCESS;
}

CMI_API int cmi_set_device(int device) { CMI_HIP(hipSetDevice(device)); return CMI_SUCCESS; }
CMI_API int cmi_get_device(int *device)
{
    if (!device) return fail(CMI_ERROR_INVALID_VALUE, "cmi_get_device: null");
    CMI_HIP(hipGetDevice(device));
    return CMI_SUCCESS;
}

CMI_API int cmi_device_info(int device, char *name, size_t name_len, int *cus, int64_t *hbm_bytes)
{
    hipDeviceProp_t p;
    CMI_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cus) *cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return CMI_SUCCESS;
}

CMI_API int cmi_malloc(void **ptr, size_t bytes)
{
    if (!ptr) return fail(CMI_ERROR_INVALID_VALUE, "cmi_malloc: null out pointer");
    *ptr = nullptr;
    if (bytes == 0) return CMI_SUCCESS;
    CMI_HIP(hipMalloc(ptr, bytes));
    return CMI_SUCCESS;
}

CMI_API int cmi_free(void *ptr)
{
    if (ptr) CMI_HIP(hipFree(ptr));
    return CMI_SUCCESS;
}

// page-locked host memory: the target of cmi_memcpy_d2h_async (a pageable target would make the
// "async" copy synchronous)
CMI_API int cmi_malloc_host(void **ptr, size_t bytes)
{
    if (!ptr) return fail(CMI_ERROR_INVALID_VALUE, "cmi_malloc_host: null out pointer");
    *ptr = nullptr;
    if (bytes == 0) return CMI_SUCCESS;
    CMI_HIP(hipHostMalloc(ptr, bytes, hipHostMallocMapped | hipHostMallocCoherent)); // device-writable, fine-grained
    return CMI_SUCCESS;
}

CMI_API int cmi_free_host(void *ptr)
{
    if (ptr) CMI_HIP(hipHostFree(ptr));
    return CMI_SUCCESS;
}

static int copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, void *stream)
{
    if (bytes == 0) return CMI_SUCCESS;
    if (!dst || !src) return fail(CMI_ERROR_INVALID_VALUE, "cmi_memcpy: null pointer");
    CMI_HIP(hipMemcpyAsync(dst, src, bytes, kind, as_stream(stream)));
    // host-side copies are synchronous from the caller's point of view, as Thrust's are
    if (kind != hipMemcpyDeviceToDevice) CMI_HIP(hipStreamSynchronize(as_stream(stream)));
    return CMI_SUCCESS;
}

CMI_API int cmi_memcpy_h2d(void *d, const void *s, size_t b, void *st) { return copy(d, s, b, hipMemcpyHostToDevice, st); }
CMI_API int cmi_memcpy_d2h(void *d, const void *s, size_t b, void *st) { return copy(d, s, b, hipMemcpyDeviceToHost, st); }
CMI_API int cmi_memcpy_d2d(void *d, const void *s, size_t b, void *st) { return copy(d, s, b, hipMemcpyDeviceToDevice, st); }

// device -> page-locked host, ordered on `stream`, NOT waited for: pair it with cmi_event_record +
// cmi_event_synchronize (the CG convergence read: the next SpMV is queued before the host waits)
CMI_API int cmi_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes == 0) return CMI_SUCCESS;
    if (!dst || !src) return fail(CMI_ERROR_INVALID_VALUE, "cmi_memcpy_d2h_async: null pointer");
    CMI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return CMI_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// Peer mapping of an HBM allocation of another process (one process per GPU) + a multi-range copy:
// what the one-sided halo exchange of the row-block sharded SpMV is made of.  xGMI is a load/store
// fabric: a rank maps its neighbours' x buffers once and then PULLS the few boundary values it
// needs with one small kernel on its own stream -- no collective, no second stream, no proxy.
// ---------------------------------------------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) <= CMI_IPC_HANDLE_BYTES, "cmi_ipc handle buffer too small");

CMI_API int cmi_ipc_get_handle(void *dev_ptr, void *handle_out)
{
    if (!dev_ptr || !handle_out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_ipc_get_handle: null pointer");
    hipIpcMemHandle_t h;
    CMI_HIP(hipIpcGetMemHandle(&h, dev_ptr));
    memset(handle_out, 0, CMI_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    return CMI_SUCCESS;
}

CMI_API int cmi_ipc_open_handle(const void *handle, void **ptr)
{
    if (!handle || !ptr) return fail(CMI_ERROR_INVALID_VALUE, "cmi_ipc_open_handle: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    *ptr = nullptr;
    CMI_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return CMI_SUCCESS;
}

// can kernels on `device` load / store memory that lives on `peer_device`? (the question RCCL asks before it
// picks its direct peer-to-peer transport); the one-sided exchange is only set up between such pairs
CMI_API int cmi_device_can_access_peer(int device, int peer_device, int *can_access)
{
    if (!can_access) return fail(CMI_ERROR_INVALID_VALUE, "cmi_device_can_access_peer: null result");
    *can_access = 0;
    if (device == peer_device) { *can_access = 1; return CMI_SUCCESS; }
    CMI_HIP(hipDeviceCanAccessPeer(can_access, device, peer_device));
    return CMI_SUCCESS;
}

CMI_API int cmi_ipc_close_handle(void *ptr)
{
    if (ptr) CMI_HIP(hipIpcCloseMemHandle(ptr));
    return CMI_SUCCESS;
}

namespace cmi {
struct copy_ranges_args {
    const unsigned char *src[CMI_MAX_COPY_RANGES];
    unsigned char *dst[CMI_MAX_COPY_RANGES];
    long long bytes[CMI_MAX_COPY_RANGES];
};

// blockIdx.y = range; 16-byte accesses when src, dst and the length allow, bytes otherwise
__global__ void __launch_bounds__(256) copy_ranges_kernel(copy_ranges_args a)
{
    const int r = blockIdx.y;
    const unsigned char *src = a.src[r];
    unsigned char *dst = a.dst[r];
    const long long n = a.bytes[r];
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const long long nv = n / 16;
        for (long long i = t; i < nv; i += stride) reinterpret_cast<int4v *>(dst)[i] = reinterpret_cast<const int4v *>(src)[i];
        for (long long i = nv * 16 + t; i < n; i += stride) dst[i] = src[i];
    } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)n) & 7) == 0) {
        const long long nv = n / 8;
        for (long long i = t; i < nv; i += stride) reinterpret_cast<long long *>(dst)[i] = reinterpret_cast<const long long *>(src)[i];
    } else {
        for (long long i = t; i < n; i += stride) dst[i] = src[i];
    }
}
} // namespace cmi

CMI_API int cmi_copy_ranges(int count, const void *const *src, void *const *dst, const int64_t *bytes, void *stream)
{
    if (count < 0 || count > CMI_MAX_COPY_RANGES) return fail(CMI_ERROR_INVALID_VALUE, "cmi_copy_ranges: count must be in [0, CMI_MAX_COPY_RANGES]");
    if (count == 0) return CMI_SUCCESS;
    if (!src || !dst || !bytes) return fail(CMI_ERROR_INVALID_VALUE, "cmi_copy_ranges: null array");
    copy_ranges_args a;
    int64_t longest = 0;
    int used = 0;
    for (int i = 0; i < count; i++) {
        if (bytes[i] < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_copy_ranges: negative length");
        if (bytes[i] == 0) continue;
        if (!src[i] || !dst[i]) return fail(CMI_ERROR_INVALID_VALUE, "cmi_copy_ranges: null range");
        a.src[used] = static_cast<const unsigned char *>(src[i]);
        a.dst[used] = static_cast<unsigned char *>(dst[i]);
        a.bytes[used] = bytes[i];
        if (bytes[i] > longest) longest = bytes[i];
        used++;
    }
    if (used == 0) return CMI_SUCCESS;
    int64_t blocks = ceil_div(longest, (int64_t)256 * 16);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_ranges_kernel, dim3((unsigned)blocks, (unsigned)used), dim3(256), 0, as_stream(stream), a);
    CMI_LAUNCH_CHECK("copy_ranges");
    return CMI_SUCCESS;
}

CMI_API int cmi_memset(void *dst, int byte_value, size_t bytes, void *stream)
{
    if (bytes == 0) return CMI_SUCCESS;
    if (!dst) return fail(CMI_ERROR_INVALID_VALUE, "cmi_memset: null pointer");
    CMI_HIP(hipMemsetAsync(dst, byte_value, bytes, as_stream(stream)));
    return CMI_SUCCESS;
}

CMI_API int cmi_stream_create(void **stream)
{
    if (!stream) return fail(CMI_ERROR_INVALID_VALUE, "cmi_stream_create: null");
    hipStream_t s;
    CMI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return CMI_SUCCESS;
}
CMI_API int cmi_stream_destroy(void *stream) { CMI_HIP(hipStreamDestroy(as_stream(stream))); return CMI_SUCCESS; }
CMI_API int cmi_stream_synchronize(void *stream) { CMI_HIP(hipStreamSynchronize(as_stream(stream))); return CMI_SUCCESS; }
CMI_API int cmi_device_synchronize(void) { CMI_HIP(hipDeviceSynchronize()); return CMI_SUCCESS; }

CMI_API int cmi_event_create(void **event)
{
    if (!event) return fail(CMI_ERROR_INVALID_VALUE, "cmi_event_create: null");
    hipEvent_t e;
    CMI_HIP(hipEventCreate(&e));
    *event = e;
    return CMI_SUCCESS;
}
CMI_API int cmi_event_destroy(void *event) { CMI_HIP(hipEventDestroy((hipEvent_t)event)); return CMI_SUCCESS; }
CMI_API int cmi_event_record(void *event, void *stream)
{
    CMI_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return CMI_SUCCESS;
}
// everything enqueued on `stream` after this call waits for `event` (recorded on another stream): the fork / join of the sharded
// multiply's interior rows on a side stream (cusp/distributed/csr_matrix.h); the host does not block
CMI_API int cmi_stream_wait_event(void *stream, void *event)
{
    if (!event) return fail(CMI_ERROR_INVALID_VALUE, "cmi_stream_wait_event: null event");
    CMI_HIP(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
    return CMI_SUCCESS;
}
CMI_API int cmi_event_synchronize(void *event)
{
    CMI_HIP(hipEventSynchronize((hipEvent_t)event));
    return CMI_SUCCESS;
}
CMI_API int cmi_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) return fail(CMI_ERROR_INVALID_VALUE, "cmi_event_elapsed_ms: null");
    CMI_HIP(hipEventSynchronize((hipEvent_t)stop));
    CMI_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return CMI_SUCCESS;
}
