// sort.hip -- the COO container's ordering on the device: cmi_coo_sort_by_row_* and cmi_coo_is_sorted.
//
// Replaces (reference): cusp/sort.h:231 sort_by_row, :302 sort_by_row_and_column as called by
// coo_matrix::sort_by_row[_and_column] (cusp/detail/coo_matrix.inl) on device_memory -- there a thrust sort of a
// permutation and three gathers.  The reference's device multiply REQUIRES row-sorted entries
// (cusp/system/cuda/detail/multiply/coo_flat_spmv.h:139-145): sorting ONCE is the caller's step, not the multiply's, so the
// sort lives here and not inside a plan (a plan that owned sorted copies of the values would go stale the first time the
// caller refreshed the values in place).
//
// Method: a STABLE least-significant-digit radix sort (rocPRIM's device radix sort: a library primitive of ROCm like RCCL,
// not a hot-path kernel) of the key (row, or row:column) carrying the entry's position; then one gather per array through
// the permutation and a copy back into the caller's arrays.  Stable = entries of one row keep their storage order, so the
// row sums of the sorted matrix are the chains the reference's host loop (sequential/multiply/coo_spmv.h) forms on the
// unsorted one: bit-identical products AND sums.  Only the key bits that can differ are sorted (ceil(log2(rows)) [+ columns]).
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace cmi {

__global__ void __launch_bounds__(256) pack_row_col_kernel(int64_t n, const int *__restrict__ Ai, const int *__restrict__ Aj, uint64_t *__restrict__ keys)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        keys[e] = ((uint64_t)(uint32_t)Ai[e] << 32) | (uint64_t)(uint32_t)Aj[e];
}

__global__ void __launch_bounds__(256) unpack_row_col_kernel(int64_t n, const uint64_t *__restrict__ keys, int *__restrict__ Ai, int *__restrict__ Aj)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const uint64_t k = keys[e];
        Ai[e] = (int)(uint32_t)(k >> 32);
        Aj[e] = (int)(uint32_t)k;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gather_kernel(int64_t n, const uint32_t *__restrict__ perm, const T *__restrict__ in, T *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) out[e] = in[perm[e]];
}

// flags[0] |= 1: a row index smaller than its predecessor; |= 2: same row, column smaller than its predecessor;
// |= 4: a row index outside [0, num_rows)
__global__ void __launch_bounds__(256)
coo_order_kernel(int64_t num_rows, int64_t n, const int *__restrict__ Ai, const int *__restrict__ Aj, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int found = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int r = Ai[e];
        if (r < 0 || (int64_t)r >= num_rows) found |= 4;
        if (e > 0) {
            const int p = Ai[e - 1];
            if (p > r) found |= 1;
            else if (Aj && p == r && Aj[e - 1] > Aj[e]) found |= 2;
        }
    }
    if (found) atomicOr(flags, found);
}

static int grid_1d(int64_t n)
{
    int64_t b = ceil_div(n, 256);
    if (b > kCus * 16) b = kCus * 16;
    return b < 1 ? 1 : (int)b;
}

static int bits_for(int64_t count) // how many low bits hold every value in [0, count)
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < count) b++;
    return b;
}

struct scratch { // device allocations of one call, released on every path out
    void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    hipError_t get(void **out, size_t bytes)
    {
        hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) p[n++] = *out;
        return e;
    }
    ~scratch() { for (int i = 0; i < n; i++) (void)hipFree(p[i]); }
};

static int coo_order(int64_t num_rows, int64_t n, const int *Ai, const int *Aj, hipStream_t s, int *flags_host)
{
    *flags_host = 0;
    if (n == 0) return CMI_SUCCESS;
    scratch mem;
    int *flags = nullptr;
    hipError_t e = mem.get((void **)&flags, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(flags, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(coo_order_kernel, dim3(grid_1d(n)), dim3(256), 0, s, num_rows, n, Ai, Aj, flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(flags_host, flags, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return e == hipSuccess ? CMI_SUCCESS : hip_fail(e, "cmi_coo_is_sorted");
}

template <typename T>
static int coo_sort(int64_t num_rows, int64_t num_cols, int64_t n, int *Ai, int *Aj, T *Ax, int and_column, void *stream)
{
    if (num_rows < 0 || num_cols < 0 || n < 0 || n > (int64_t)INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_sort_by_row: bad size");
    if (n > 0 && (!Ai || !Aj || !Ax)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_sort_by_row: null array");
    if (n <= 1) return CMI_SUCCESS;
    hipStream_t s = as_stream(stream);
    int flags = 0;
    int st = coo_order(num_rows, n, Ai, and_column ? Aj : nullptr, s, &flags);
    if (st != CMI_SUCCESS) return st;
    if (flags & 4) return fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_sort_by_row: a row index lies outside [0, num_rows)");
    if (!(flags & 3)) return CMI_SUCCESS; // already in the order asked for: nothing moves

    scratch mem;
    uint32_t *perm = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = mem.get((void **)&perm, (size_t)n * sizeof(uint32_t));
    if (e != hipSuccess) return hip_fail(e, "cmi_coo_sort_by_row: scratch");
    rocprim::counting_iterator<uint32_t> position(0);
    const unsigned size = (unsigned)n;
    if (!and_column) {
        int *rows_sorted = nullptr, *cols_sorted = nullptr;
        T *vals_sorted = nullptr;
        const unsigned end_bit = (unsigned)bits_for(num_rows);
        e = mem.get((void **)&rows_sorted, (size_t)n * sizeof(int));
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint32_t *)Ai, (uint32_t *)rows_sorted, position, perm, size, 0u, end_bit, s);
        if (e == hipSuccess) e = mem.get(&temp, temp_bytes);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)Ai, (uint32_t *)rows_sorted, position, perm, size, 0u, end_bit, s);
        if (e == hipSuccess) e = hipMemcpyAsync(Ai, rows_sorted, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, s);
        // the row scratch is free again once its copy has been enqueued behind it on the stream: reuse it for the columns
        cols_sorted = rows_sorted;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_kernel<int>, dim3(grid_1d(n)), dim3(256), 0, s, n, perm, (const int *)Aj, cols_sorted);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(Aj, cols_sorted, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = mem.get((void **)&vals_sorted, (size_t)n * sizeof(T));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_kernel<T>, dim3(grid_1d(n)), dim3(256), 0, s, n, perm, (const T *)Ax, vals_sorted);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(Ax, vals_sorted, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, s);
    } else {
        uint64_t *keys = nullptr, *keys_sorted = nullptr;
        T *vals_sorted = nullptr;
        const unsigned end_bit = 32u + (unsigned)bits_for(num_rows);
        e = mem.get((void **)&keys, (size_t)n * sizeof(uint64_t));
        if (e == hipSuccess) e = mem.get((void **)&keys_sorted, (size_t)n * sizeof(uint64_t));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pack_row_col_kernel, dim3(grid_1d(n)), dim3(256), 0, s, n, (const int *)Ai, (const int *)Aj, keys);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint64_t *)keys, keys_sorted, position, perm, size, 0u, end_bit, s);
        if (e == hipSuccess) e = mem.get(&temp, temp_bytes);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t *)keys, keys_sorted, position, perm, size, 0u, end_bit, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(unpack_row_col_kernel, dim3(grid_1d(n)), dim3(256), 0, s, n, (const uint64_t *)keys_sorted, Ai, Aj);
            e = hipGetLastError();
        }
        vals_sorted = (T *)keys; // (8 bytes per entry, no longer read)
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_kernel<T>, dim3(grid_1d(n)), dim3(256), 0, s, n, perm, (const T *)Ax, vals_sorted);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(Ax, vals_sorted, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, s);
    }
    // the scratch is released when this returns: everything enqueued above must have finished with it
    hipError_t e2 = hipStreamSynchronize(s);
    if (e == hipSuccess) e = e2;
    return e == hipSuccess ? CMI_SUCCESS : hip_fail(e, "cmi_coo_sort_by_row");
}

} // namespace cmi

CMI_API int cmi_coo_sort_by_row_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, int32_t *Ai, int32_t *Aj, double *Ax, int and_column, void *stream)
{ return cmi::coo_sort<double>(num_rows, num_cols, num_entries, Ai, Aj, Ax, and_column, stream); }
CMI_API int cmi_coo_sort_by_row_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, int32_t *Ai, int32_t *Aj, float *Ax, int and_column, void *stream)
{ return cmi::coo_sort<float>(num_rows, num_cols, num_entries, Ai, Aj, Ax, and_column, stream); }

CMI_API int cmi_coo_is_sorted(int64_t num_rows, int64_t num_entries, const int32_t *Ai, const int32_t *Aj, int and_column, int *sorted_host, void *stream)
{
    if (num_rows < 0 || num_entries < 0) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_is_sorted: negative size");
    if (!sorted_host || (num_entries > 0 && (!Ai || (and_column && !Aj)))) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_is_sorted: null array");
    int flags = 0;
    const int st = cmi::coo_order(num_rows, num_entries, Ai, and_column ? Aj : nullptr, cmi::as_stream(stream), &flags);
    if (st != CMI_SUCCESS) return st;
    *sorted_host = (flags & 3) == 0; // (an out-of-range row index is the multiply's to refuse; order is all that is asked here)
    return CMI_SUCCESS;
}
