// builders.hip -- on-device construction of the benchmark inputs and the CSR -> {ELL, COO} layout
// transforms, so 1e7..1e8-row matrices never pass through the host.
//
// Behaviour restated from the reference (integer outputs must match the host path bit-for-bit):
//   cusp::gallery::poisson5pt            cusp/gallery/detail/poisson.inl:29-47
//   generate_matrix_from_stencil (DIA)   cusp/gallery/detail/stencil.inl:118-134,143-188
//   DIA -> CSR (drop zeros, row-major)   cusp/system/detail/generic/conversions/dia_to_other.h:109-163
//   CSR -> ELL                           cusp/system/detail/generic/conversions/csr_to_other.h:155-227
//   CSR -> COO row indices               csr_to_other.h:56-70 (offsets_to_indices)
//   ELLR row lengths                     cusp/ktt/detail/ellr_matrix.inl:16-53
//
// poisson5pt(m, n): grid point (ix, iy), row r = iy*m + ix, stencil (0,-1),(-1,0),(0,0),(1,0),(0,1)
// -> diagonal offsets [-m,-1,0,1,m], values -1,-1,4,-1,-1, a neighbour counts only if it lies inside
// the grid.  The CSR row therefore holds, in ascending column order, {r-m | iy>0}, {r-1 | ix>0}, r,
// {r+1 | ix<m-1}, {r+m | iy<n-1}; its offset has the closed form used below, so every lane writes
// its row independently (no scan).
#include "common.h"
#include <vector>

namespace cmi {

// number of entries in rows [0, r) of poisson5pt(m, n)
__host__ __device__ inline int64_t poisson_prefix(int64_t m, int64_t n, int64_t r)
{
    const int64_t iy = r / m, ix = r % m; // r may equal m*n (iy == n, ix == 0)
    int64_t c = 5 * r;
    c -= iy + (ix > 0 ? 1 : 0);              // rows before r with ix == 0      (no left neighbour)
    c -= iy;                                 // rows before r with ix == m-1    (no right neighbour)
    c -= iy > 0 ? m : ix;                    // rows before r with iy == 0      (no lower neighbour)
    c -= iy >= n ? m : (iy == n - 1 ? ix : 0); // rows before r with iy == n-1  (no upper neighbour)
    return c;
}

template <typename T>
__global__ void __launch_bounds__(256)
poisson_csr_kernel(int64_t m, int64_t n, int64_t row_begin, int64_t row_end, int *__restrict__ Ap,
                   int *__restrict__ Aj, T *__restrict__ Ax)
{
    const int64_t nrows = row_end - row_begin;
    const int64_t origin = poisson_prefix(m, n, row_begin);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrows; i += stride) {
        const int64_t r = row_begin + i;
        int64_t p = poisson_prefix(m, n, r) - origin;
        Ap[i] = (int)p;
        if (i == nrows) break;
        const int64_t iy = r / m, ix = r % m;
        if (iy > 0)     { Aj[p] = (int)(r - m); Ax[p] = T(-1); p++; }
        if (ix > 0)     { Aj[p] = (int)(r - 1); Ax[p] = T(-1); p++; }
        Aj[p] = (int)r; Ax[p] = T(4); p++;
        if (ix < m - 1) { Aj[p] = (int)(r + 1); Ax[p] = T(-1); p++; }
        if (iy < n - 1) { Aj[p] = (int)(r + m); Ax[p] = T(-1); p++; }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
poisson_dia_kernel(int64_t m, int64_t n, int64_t pitch, int *__restrict__ offsets, T *__restrict__ vals)
{
    const int64_t N = m * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        offsets[0] = (int)-m; offsets[1] = -1; offsets[2] = 0; offsets[3] = 1; offsets[4] = (int)m;
    }
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < pitch; r += stride) {
        const bool in = r < N;
        const int64_t iy = r / m, ix = r % m;
        vals[0 * pitch + r] = (in && iy > 0) ? T(-1) : T(0);
        vals[1 * pitch + r] = (in && ix > 0) ? T(-1) : T(0);
        vals[2 * pitch + r] = in ? T(4) : T(0);
        vals[3 * pitch + r] = (in && ix < m - 1) ? T(-1) : T(0);
        vals[4 * pitch + r] = (in && iy < n - 1) ? T(-1) : T(0);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
csr_to_ell_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj,
                  const T *__restrict__ Ax, int width, int64_t pitch, int *__restrict__ ell_Aj, T *__restrict__ ell_Ax)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pitch; i += stride) {
        int s = 0, e = 0;
        if (i < num_rows) { s = Ap[i]; e = Ap[i + 1]; }
        for (int k = 0; k < width; k++) {
            const bool have = s + k < e;
            ell_Aj[(int64_t)k * pitch + i] = have ? Aj[s + k] : -1;
            ell_Ax[(int64_t)k * pitch + i] = have ? Ax[s + k] : T(0);
        }
    }
}

// CSR -> HYB, COO part: the entries at within-row index >= width, in CSR order.  coo_offsets[i] =
// number of such entries in rows [0, i) (an exclusive scan of max(0, len_i - width), supplied by the
// caller: it is a function of the row offsets alone).  A 64-lane wave per row keeps the copies coalesced.
template <typename T>
__global__ void __launch_bounds__(256)
csr_to_hyb_coo_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, const T *__restrict__ Ax,
                      int width, const int *__restrict__ coo_offsets, int *__restrict__ coo_Ai, int *__restrict__ coo_Aj,
                      T *__restrict__ coo_Ax)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / kWave;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave; row < num_rows; row += nwaves) {
        const int s = Ap[row] + width, e = Ap[row + 1];
        const int dst = coo_offsets[row];
        for (int jj = s + lane; jj < e; jj += kWave) {
            const int o = dst + (jj - s);
            coo_Ai[o] = (int)row;
            coo_Aj[o] = Aj[jj];
            coo_Ax[o] = Ax[jj];
        }
    }
}

__global__ void __launch_bounds__(256)
csr_row_indices_kernel(int64_t num_rows, const int *__restrict__ Ap, int *__restrict__ Ai)
{
    // a 64-lane wave per row keeps the writes coalesced for long rows; short rows cost one pass
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / kWave;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave; row < num_rows; row += nwaves) {
        const int s = Ap[row], e = Ap[row + 1];
        for (int jj = s + lane; jj < e; jj += kWave) Ai[jj] = (int)row;
    }
}

__global__ void __launch_bounds__(256)
ell_row_lengths_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ ell_Aj,
                       int *__restrict__ row_lengths)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int len = 0;
        while (len < width && ell_Aj[(int64_t)len * pitch + i] >= 0) len++;
        row_lengths[i] = len;
    }
}

// ---- CSR -> DIA (reference conversions/csr_to_other.h:73-153: count the occupied diagonals, fill) ----
// slot_map has one int per possible diagonal, index = (col - row) + num_rows.  Pass 1 flags the occupied
// ones (plain stores of 1: racing writers agree).  Pass 2 appends each flagged offset to an UNORDERED list
// through one atomic counter (D atomics in all); the caller sorts the D offsets (ascending, as the
// reference stores them) and hands them back.  Pass 3 turns the flags into slot numbers, pass 4 scatters.
__global__ void __launch_bounds__(256)
dia_flag_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, int *__restrict__ slot_map)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / kWave;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave; row < num_rows; row += nwaves) {
        const int s = Ap[row], e = Ap[row + 1];
        for (int jj = s + lane; jj < e; jj += kWave) slot_map[(int64_t)Aj[jj] - row + num_rows] = 1;
    }
}

__global__ void __launch_bounds__(256)
dia_list_kernel(int64_t num_rows, int64_t map_len, const int *__restrict__ slot_map, int *__restrict__ list, int64_t capacity,
                unsigned long long *__restrict__ counter)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < map_len; s += stride) {
        if (slot_map[s]) {
            const unsigned long long k = atomicAdd(counter, 1ull);
            if ((int64_t)k < capacity) list[k] = (int)(s - num_rows);
        }
    }
}

__global__ void __launch_bounds__(256)
dia_slot_kernel(int64_t num_rows, int64_t num_diagonals, const int *__restrict__ offsets, int *__restrict__ slot_map)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < num_diagonals) slot_map[(int64_t)offsets[k] + num_rows] = (int)k;
}

template <typename T>
__global__ void __launch_bounds__(256)
dia_scatter_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, const T *__restrict__ Ax,
                   const int *__restrict__ slot_map, int64_t pitch, T *__restrict__ values)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / kWave;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave; row < num_rows; row += nwaves) {
        const int s = Ap[row], e = Ap[row + 1];
        for (int jj = s + lane; jj < e; jj += kWave)
            values[(int64_t)slot_map[(int64_t)Aj[jj] - row + num_rows] * pitch + row] = Ax[jj];
    }
}

static int grid_1d(int64_t n)
{
    int64_t b = ceil_div(n, 256);
    if (b > kCus * 16) b = kCus * 16;
    return b < 1 ? 1 : (int)b;
}

template <typename T>
static int poisson_csr(int64_t m, int64_t n, int64_t rb, int64_t re, int *Ap, int *Aj, T *Ax, void *stream)
{
    if (m < 1 || n < 1 || rb < 0 || re < rb || re > m * n) return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_csr: bad grid or row range");
    if (m * n > INT32_MAX || cmi_poisson5pt_shard_entries(m, n, rb, re) > INT32_MAX)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_csr: sizes exceed the int32 index type");
    if (!Ap || (re > rb && (!Aj || !Ax))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_csr: null array");
    hipLaunchKernelGGL((poisson_csr_kernel<T>), dim3(grid_1d(re - rb + 1)), dim3(256), 0, as_stream(stream), m, n, rb, re, Ap, Aj, Ax);
    CMI_LAUNCH_CHECK("poisson5pt csr");
    return CMI_SUCCESS;
}

template <typename T>
static int poisson_dia(int64_t m, int64_t n, int64_t pitch, int *offsets, T *vals, void *stream)
{
    if (m < 1 || n < 1 || pitch < m * n) return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_dia: bad grid or pitch");
    if (m * n > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_dia: sizes exceed the int32 index type");
    if (!offsets || !vals) return fail(CMI_ERROR_INVALID_VALUE, "cmi_poisson5pt_dia: null array");
    hipLaunchKernelGGL((poisson_dia_kernel<T>), dim3(grid_1d(pitch)), dim3(256), 0, as_stream(stream), m, n, pitch, offsets, vals);
    CMI_LAUNCH_CHECK("poisson5pt dia");
    return CMI_SUCCESS;
}

template <typename T>
static int csr_to_hyb_coo(int64_t rows, const int *Ap, const int *Aj, const T *Ax, int64_t width, const int *coo_offsets,
                          int *coo_Ai, int *coo_Aj, T *coo_Ax, void *stream)
{
    if (rows < 0 || width < 0 || width > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_hyb_coo: bad size");
    if (rows == 0) return CMI_SUCCESS;
    if (!Ap || !coo_offsets) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_hyb_coo: null array");
    hipLaunchKernelGGL((csr_to_hyb_coo_kernel<T>), dim3(grid_1d(rows * 8)), dim3(256), 0, as_stream(stream), rows, Ap, Aj, Ax, (int)width,
                       coo_offsets, coo_Ai, coo_Aj, coo_Ax);
    CMI_LAUNCH_CHECK("csr_to_hyb_coo");
    return CMI_SUCCESS;
}

template <typename T>
static int csr_to_ell(int64_t rows, const int *Ap, const int *Aj, const T *Ax, int64_t width, int64_t pitch,
                      int *ell_Aj, T *ell_Ax, void *stream)
{
    if (rows < 0 || width < 0 || width > INT32_MAX || pitch < rows) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_ell: bad size");
    if (width == 0 || pitch == 0) return CMI_SUCCESS;
    if (!Ap || !ell_Aj || !ell_Ax) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_ell: null array");
    hipLaunchKernelGGL((csr_to_ell_kernel<T>), dim3(grid_1d(pitch)), dim3(256), 0, as_stream(stream), rows, Ap, Aj, Ax, (int)width, pitch, ell_Aj, ell_Ax);
    CMI_LAUNCH_CHECK("csr_to_ell");
    return CMI_SUCCESS;
}

template <typename T>
static int csr_to_dia(int64_t rows, int64_t cols, const int *Ap, const int *Aj, const T *Ax, int64_t ndiag, int64_t pitch,
                      const int *offsets, int *slot_map, T *values, void *stream)
{
    if (rows < 0 || cols < 0 || ndiag < 0 || pitch < rows) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_dia: bad size");
    if (rows == 0 || ndiag == 0) return CMI_SUCCESS;
    if (!Ap || !Aj || !Ax || !offsets || !slot_map || !values) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_to_dia: null array");
    hipStream_t s = as_stream(stream);
    CMI_HIP(hipMemsetAsync(values, 0, (size_t)pitch * ndiag * sizeof(T), s));
    hipLaunchKernelGGL(dia_slot_kernel, dim3((unsigned)ceil_div(ndiag, 256)), dim3(256), 0, s, rows, ndiag, offsets, slot_map);
    hipLaunchKernelGGL((dia_scatter_kernel<T>), dim3(grid_1d(rows * 8)), dim3(256), 0, s, rows, Ap, Aj, Ax, slot_map, pitch, values);
    CMI_LAUNCH_CHECK("csr_to_dia");
    return CMI_SUCCESS;
}

} // namespace cmi

using namespace cmi;

// Occupied diagonals of a CSR matrix.  slot_map: num_rows + num_cols ints (scratch, kept for
// cmi_csr_to_dia_*); diag_list: `capacity` ints, receives the offsets (col - row) in NO particular
// order; *num_diagonals_host: how many there are (may exceed capacity: then the list is truncated and
// the caller should give up, as the reference does when the fill-in is too large).  Synchronises.
CMI_API int cmi_csr_diagonals(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, int32_t *slot_map,
                              int32_t *diag_list, int64_t capacity, int64_t *num_diagonals_host, void *stream)
{
    if (num_rows < 0 || num_cols < 0 || capacity < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_diagonals: bad size");
    if (!num_diagonals_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_diagonals: null result");
    *num_diagonals_host = 0;
    if (num_rows == 0 || num_cols == 0) return CMI_SUCCESS;
    if (!Ap || !slot_map || (capacity > 0 && !diag_list)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_diagonals: null array");
    hipStream_t s = as_stream(stream);
    const int64_t map_len = num_rows + num_cols;
    unsigned long long *counter = nullptr;
    CMI_HIP(hipMalloc((void **)&counter, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemsetAsync(slot_map, 0, (size_t)map_len * sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dia_flag_kernel, dim3(grid_1d(num_rows * 8)), dim3(256), 0, s, num_rows, Ap, Aj, slot_map);
        hipLaunchKernelGGL(dia_list_kernel, dim3(grid_1d(map_len)), dim3(256), 0, s, num_rows, map_len, slot_map, diag_list, capacity, counter);
        e = hipGetLastError();
    }
    unsigned long long n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n, counter, sizeof(n), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(counter);
    if (e != hipSuccess) return hip_fail(e, "cmi_csr_diagonals");
    *num_diagonals_host = (int64_t)n;
    return CMI_SUCCESS;
}

// values[k*pitch + i] <- A(i, i + offsets[k]) for the SORTED offsets the caller derived from
// cmi_csr_diagonals (same slot_map, still flagged); everything else in `values` is zeroed.
CMI_API int cmi_csr_to_dia_f64(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                               int64_t num_diagonals, int64_t pitch, const int32_t *offsets, int32_t *slot_map, double *values, void *stream)
{ return csr_to_dia<double>(num_rows, num_cols, Ap, Aj, Ax, num_diagonals, pitch, offsets, slot_map, values, stream); }
CMI_API int cmi_csr_to_dia_f32(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                               int64_t num_diagonals, int64_t pitch, const int32_t *offsets, int32_t *slot_map, float *values, void *stream)
{ return csr_to_dia<float>(num_rows, num_cols, Ap, Aj, Ax, num_diagonals, pitch, offsets, slot_map, values, stream); }


CMI_API int64_t cmi_poisson5pt_num_entries(int64_t m, int64_t n)
{
    if (m < 1 || n < 1) return 0;
    return poisson_prefix(m, n, m * n);
}

CMI_API int64_t cmi_poisson5pt_shard_entries(int64_t m, int64_t n, int64_t row_begin, int64_t row_end)
{
    if (m < 1 || n < 1 || row_begin < 0 || row_end < row_begin || row_end > m * n) return 0;
    return poisson_prefix(m, n, row_end) - poisson_prefix(m, n, row_begin);
}

CMI_API int cmi_poisson5pt_csr_f64(int64_t m, int64_t n, int64_t rb, int64_t re, int32_t *Ap, int32_t *Aj, double *Ax, void *stream)
{ return poisson_csr<double>(m, n, rb, re, Ap, Aj, Ax, stream); }
CMI_API int cmi_poisson5pt_csr_f32(int64_t m, int64_t n, int64_t rb, int64_t re, int32_t *Ap, int32_t *Aj, float *Ax, void *stream)
{ return poisson_csr<float>(m, n, rb, re, Ap, Aj, Ax, stream); }
CMI_API int cmi_poisson5pt_dia_f64(int64_t m, int64_t n, int64_t pitch, int32_t *offsets, double *values, void *stream)
{ return poisson_dia<double>(m, n, pitch, offsets, values, stream); }
CMI_API int cmi_poisson5pt_dia_f32(int64_t m, int64_t n, int64_t pitch, int32_t *offsets, float *values, void *stream)
{ return poisson_dia<float>(m, n, pitch, offsets, values, stream); }

CMI_API int cmi_csr_to_ell_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax, int64_t width,
                               int64_t pitch, int32_t *ell_Aj, double *ell_Ax, void *stream)
{ return csr_to_ell<double>(num_rows, Ap, Aj, Ax, width, pitch, ell_Aj, ell_Ax, stream); }
CMI_API int cmi_csr_to_ell_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax, int64_t width,
                               int64_t pitch, int32_t *ell_Aj, float *ell_Ax, void *stream)
{ return csr_to_ell<float>(num_rows, Ap, Aj, Ax, width, pitch, ell_Aj, ell_Ax, stream); }

CMI_API int cmi_csr_to_hyb_coo_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax, int64_t width,
                                   const int32_t *coo_offsets, int32_t *coo_Ai, int32_t *coo_Aj, double *coo_Ax, void *stream)
{ return csr_to_hyb_coo<double>(num_rows, Ap, Aj, Ax, width, coo_offsets, coo_Ai, coo_Aj, coo_Ax, stream); }
CMI_API int cmi_csr_to_hyb_coo_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax, int64_t width,
                                   const int32_t *coo_offsets, int32_t *coo_Ai, int32_t *coo_Aj, float *coo_Ax, void *stream)
{ return csr_to_hyb_coo<float>(num_rows, Ap, Aj, Ax, width, coo_offsets, coo_Ai, coo_Aj, coo_Ax, stream); }

CMI_API int cmi_csr_row_indices(int64_t num_rows, const int32_t *Ap, int32_t *Ai, void *stream)
{
    if (num_rows < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_row_indices: negative size");
    if (num_rows == 0) return CMI_SUCCESS;
    if (!Ap) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_row_indices: null array");
    hipLaunchKernelGGL(csr_row_indices_kernel, dim3(grid_1d(num_rows * 8)), dim3(256), 0, as_stream(stream), num_rows, Ap, Ai);
    CMI_LAUNCH_CHECK("csr_row_indices");
    return CMI_SUCCESS;
}

// Smallest and largest column index of `num_entries` entries (a row block's gather window: what the sharded operator's exchange
// plan is made from, cusp/distributed/csr_matrix.h).  Set-up call: allocates 8 bytes of scratch and synchronises the stream.
// No entries: *min_host = 0, *max_host = -1.
namespace cmi {
__global__ void __launch_bounds__(256) column_span_kernel(int64_t n, const int *__restrict__ Aj, int *__restrict__ out)
{
    int lo = INT32_MAX, hi = INT32_MIN;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = Aj[i];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const int a = __shfl_down(lo, o), b = __shfl_down(hi, o);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicMin(out, lo); atomicMax(out + 1, hi); }
}
} // namespace cmi

CMI_API int cmi_csr_column_span(int64_t num_entries, const int32_t *Aj, int32_t *min_host, int32_t *max_host, void *stream)
{
    if (num_entries < 0 || !min_host || !max_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_column_span: bad argument");
    *min_host = 0;
    *max_host = -1;
    if (num_entries == 0) return CMI_SUCCESS;
    if (!Aj) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_column_span: null column indices");
    hipStream_t s = as_stream(stream);
    int *dev = nullptr, host[2] = {INT32_MAX, INT32_MIN};
    CMI_HIP(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(num_entries, 256 * 8);
        if (blocks > kCus * 8) blocks = kCus * 8;
        hipLaunchKernelGGL(column_span_kernel, dim3((unsigned)blocks), dim3(256), 0, s, num_entries, Aj, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "cmi_csr_column_span");
    *min_host = host[0];
    *max_host = host[1];
    return CMI_SUCCESS;
}

// Row offsets of a row block cut out of a larger CSR matrix: out[i] = Ap[i] - Ap[0] for i <= num_rows (in place allowed).
namespace cmi {
// (lane per row: set-up work, once per sharded operator)
__global__ void __launch_bounds__(256)
interior_rows_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, int64_t lo, int64_t hi, int64_t mid, int *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < num_rows; r += stride) {
        bool outside = false;
        for (int jj = Ap[r]; jj < Ap[r + 1] && !outside; jj++) outside = Aj[jj] < lo || Aj[jj] >= hi;
        if (outside) {
            if (r < mid) atomicMax(out, (int)r); else atomicMin(out + 1, (int)r);
        }
    }
}
} // namespace cmi

CMI_API int cmi_csr_interior_rows(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, int64_t col_lo, int64_t col_hi, int64_t *first_host,
                                  int64_t *last_host, void *stream)
{
    if (num_rows < 0 || num_rows > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_interior_rows: bad size");
    if (!first_host || !last_host || (num_rows > 0 && !Ap)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_interior_rows: null array");
    *first_host = 0;
    *last_host = num_rows;
    if (num_rows == 0) return CMI_SUCCESS;
    hipStream_t s = as_stream(stream);
    int *dev = nullptr, host[2] = {-1, (int)num_rows};
    CMI_HIP(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(interior_rows_kernel, dim3(grid_1d(num_rows)), dim3(256), 0, s, num_rows, Ap, Aj, col_lo, col_hi, num_rows / 2, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "cmi_csr_interior_rows");
    *first_host = (int64_t)host[0] + 1;
    *last_host = host[1];
    return CMI_SUCCESS;
}

namespace cmi {
__global__ void __launch_bounds__(256) rebase_offsets_kernel(int64_t n, const int *__restrict__ Ap, int base, int *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Ap[i] - base;
}
} // namespace cmi
CMI_API int cmi_csr_rebase_offsets(int64_t num_rows, const int32_t *Ap, int32_t base, int32_t *out, void *stream)
{
    if (num_rows < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_rebase_offsets: negative size");
    if (!Ap || !out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_rebase_offsets: null array");
    hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)ceil_div(num_rows + 1, 256)), dim3(256), 0, as_stream(stream), num_rows + 1, Ap, base, out);
    CMI_LAUNCH_CHECK("rebase_offsets");
    return CMI_SUCCESS;
}

// Row offsets of a row-sorted COO matrix: entry e with row r closes the offsets of every row in (row of entry e-1, r]
// at e; the thread past the last entry closes the rest at num_entries.  Also the order check: *unsorted != 0 afterwards
// means some row index was smaller than its predecessor (or out of range) and Ap is not to be used.
namespace cmi {
__global__ void __launch_bounds__(256)
coo_row_offsets_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ai, int *__restrict__ Ap, int *__restrict__ unsorted)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= num_entries; e += stride) {
        const int64_t prev = e > 0 ? Ai[e - 1] : -1;
        const int64_t cur = e < num_entries ? Ai[e] : num_rows;
        // prev is validated HERE too (thread e-1 also flags it, but this thread's fill loop starts at prev + 1: a negative
        // row index in front of a valid one would otherwise write in front of Ap)
        if (prev < -1 || (e > 0 && prev < 0) || prev >= num_rows || cur < prev || cur < 0 || cur > num_rows ||
            (e < num_entries && cur >= num_rows)) { atomicOr(unsorted, 1); continue; }
        const int64_t hi = e < num_entries ? cur : num_rows; // rows (prev, hi] start at e (the last thread: rows beyond the last entry, and Ap[num_rows])
        for (int64_t r = prev + 1; r <= hi; r++) Ap[r] = (int)e;
    }
}
} // namespace cmi

CMI_API int cmi_coo_row_offsets(int64_t num_rows, int64_t num_entries, const int32_t *Ai, int32_t *Ap, int *sorted_host, void *stream)
{
    if (num_rows < 0 || num_entries < 0 || num_entries > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_row_offsets: bad size");
    if (!Ap || !sorted_host || (num_entries > 0 && !Ai)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_coo_row_offsets: null array");
    *sorted_host = 0;
    hipStream_t s = as_stream(stream);
    int *flag = nullptr;
    CMI_HIP(hipMalloc((void **)&flag, sizeof(int)));
    int host = 1;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(coo_row_offsets_kernel, dim3(grid_1d(num_entries + 1)), dim3(256), 0, s, num_rows, num_entries, Ai, Ap, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(flag);
    if (e != hipSuccess) return hip_fail(e, "cmi_coo_row_offsets");
    *sorted_host = host == 0;
    return CMI_SUCCESS;
}

CMI_API int cmi_ell_row_lengths(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj,
                                int32_t *row_lengths, void *stream)
{
    if (num_rows < 0 || width < 0 || width > INT32_MAX || (width > 0 && pitch < num_rows))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_ell_row_lengths: bad size");
    if (num_rows == 0) return CMI_SUCCESS;
    if (!row_lengths || (width > 0 && !ell_Aj)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_ell_row_lengths: null array");
    hipLaunchKernelGGL(ell_row_lengths_kernel, dim3(grid_1d(num_rows)), dim3(256), 0, as_stream(stream), num_rows, (int)width, pitch, ell_Aj, row_lengths);
    CMI_LAUNCH_CHECK("ell_row_lengths");
    return CMI_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// ELL -> CSR and DIA -> CSR on the device (reference conversions/ell_to_other.h, dia_to_other.h:107-160: keep the
// entries with a valid column -- DIA: and a non-zero value -- in row-major order): count per row, exclusive scan,
// scatter.  Setup-time code: the scan is the plain three-pass one (tile sums, scan of the sums, apply).
// ---------------------------------------------------------------------------------------------
namespace cmi {

constexpr int kScanTile = 2048; // 256 lanes x 8
constexpr int64_t kScanMaxCount = INT32_MAX / kScanTile; // per-row counts up to this keep a tile's int32 sum exact

__global__ void __launch_bounds__(256) scan_tile_sums_kernel(int64_t n, const int *__restrict__ in, int *__restrict__ sums)
{
    __shared__ int slots[256 / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    int acc = 0;
    for (int k = 0; k < 8; k++) {
        const int64_t i = base + threadIdx.x * 8 + k;
        if (i < n) acc += in[i];
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & (kWave - 1)) == 0) slots[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = slots[0] + slots[1] + slots[2] + slots[3];
}

// one workgroup: sums[] -> exclusive scan in place, the grand total to *total.  The running total is kept in 64 bits:
// a total beyond INT32_MAX (rows*width or rows*diagonals can exceed it) is reported as *total = -1, never wrapped.
__global__ void __launch_bounds__(256) scan_sums_kernel(int64_t ntiles, int *__restrict__ sums, int *__restrict__ total)
{
    __shared__ long long buf[256];
    long long carry = 0;
    for (int64_t base = 0; base < ntiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const long long v = i < ntiles ? sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { // Hillis-Steele inclusive scan of 256 values
            const long long t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < ntiles) sums[i] = (int)(carry + buf[threadIdx.x] - v); // meaningless past an overflow; the caller stops on total < 0
        const long long chunk = buf[255];
        __syncthreads();
        carry += chunk;
    }
    if (threadIdx.x == 0) *total = carry > (long long)INT32_MAX ? -1 : (int)carry;
}

__global__ void __launch_bounds__(256) scan_apply_kernel(int64_t n, const int *__restrict__ in, const int *__restrict__ sums, int *__restrict__ out)
{
    __shared__ int wave_tot[256 / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
    int v[8], local = 0;
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : 0; local += v[k]; }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    int incl = local; // inclusive scan of the lanes' totals inside the wave
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    __syncthreads();
    int off = sums[blockIdx.x] + incl - local;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    for (int k = 0; k < 8; k++) { if (base + k < n) out[base + k] = off; off += v[k]; }
}

__global__ void __launch_bounds__(256)
ell_valid_counts_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ ell_Aj, int *__restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int c = 0;
        for (int k = 0; k < width; k++) c += ell_Aj[(int64_t)k * pitch + i] >= 0;
        counts[i] = c;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
ell_to_csr_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ ell_Aj, const T *__restrict__ ell_Ax,
                  const int *__restrict__ Ap, int *__restrict__ Aj, T *__restrict__ Ax)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int p = Ap[i];
        for (int k = 0; k < width; k++) {
            const int c = ell_Aj[(int64_t)k * pitch + i];
            if (c >= 0) { Aj[p] = c; Ax[p] = ell_Ax[(int64_t)k * pitch + i]; p++; }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
dia_valid_counts_kernel(int64_t num_rows, int64_t num_cols, int nd, int64_t pitch, const int *__restrict__ offsets,
                        const T *__restrict__ vals, int *__restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int c = 0;
        for (int d = 0; d < nd; d++) {
            const int64_t j = i + offsets[d];
            c += j >= 0 && j < num_cols && vals[(int64_t)d * pitch + i] != T(0);
        }
        counts[i] = c;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
dia_to_csr_kernel(int64_t num_rows, int64_t num_cols, int nd, int64_t pitch, const int *__restrict__ offsets, const T *__restrict__ vals,
                  const int *__restrict__ Ap, int *__restrict__ Aj, T *__restrict__ Ax)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int p = Ap[i];
        for (int d = 0; d < nd; d++) {
            const int64_t j = i + offsets[d];
            const T v = vals[(int64_t)d * pitch + i];
            if (j >= 0 && j < num_cols && v != T(0)) { Aj[p] = (int)j; Ax[p] = v; p++; }
        }
    }
}

// out[0..n] <- exclusive scan of in[0..n) (out[n] = the total); returns the total to the host.  Synchronises the stream.
static int exclusive_scan_i32(int64_t n, const int *in, int *out, int64_t *total_host, hipStream_t s)
{
    const int64_t ntiles = ceil_div(n > 0 ? n : 1, (int64_t)kScanTile);
    int *sums = nullptr;
    CMI_HIP(hipMalloc((void **)&sums, (size_t)ntiles * sizeof(int)));
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, n, in, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, s, ntiles, sums, out + n);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, n, in, sums, out);
    hipError_t e = hipGetLastError();
    int total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, out + n, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(sums);
    if (e != hipSuccess) return hip_fail(e, "exclusive scan");
    if (total < 0) return fail(CMI_ERROR_NOT_SUPPORTED, "conversion to CSR: more than INT32_MAX entries (int32 row offsets cannot address them)");
    if (total_host) *total_host = total;
    return CMI_SUCCESS;
}

template <typename T>
static int ell_to_csr(int64_t rows, int64_t width, int64_t pitch, const int *eAj, const T *eAx, int *Ap, int *Aj, T *Ax,
                      int64_t capacity, int64_t *nnz_host, void *stream)
{
    if (rows < 0 || width < 0 || width > kScanMaxCount || (width > 0 && pitch < rows)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_ell_to_csr: bad size");
    if (!Ap || !nnz_host || (rows > 0 && width > 0 && (!eAj || !eAx))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_ell_to_csr: null array");
    hipStream_t s = as_stream(stream);
    int *counts = nullptr;
    CMI_HIP(hipMalloc((void **)&counts, (size_t)(rows > 0 ? rows : 1) * sizeof(int)));
    if (rows > 0) hipLaunchKernelGGL(ell_valid_counts_kernel, dim3(grid_1d(rows)), dim3(256), 0, s, rows, (int)width, pitch, eAj, counts);
    int st = exclusive_scan_i32(rows, counts, Ap, nnz_host, s);
    if (st == CMI_SUCCESS && Aj && Ax && *nnz_host > capacity) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_ell_to_csr: capacity is smaller than the entry count");
    if (st == CMI_SUCCESS && *nnz_host > 0 && Aj && Ax) { // Aj == NULL: a sizing call
        hipLaunchKernelGGL((ell_to_csr_kernel<T>), dim3(grid_1d(rows)), dim3(256), 0, s, rows, (int)width, pitch, eAj, eAx, Ap, Aj, Ax);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = fail(CMI_ERROR_HIP, "cmi_ell_to_csr: scatter failed");
    }
    (void)hipFree(counts);
    return st;
}

template <typename T>
static int dia_to_csr(int64_t rows, int64_t cols, int64_t nd, int64_t pitch, const int *offsets, const T *vals, int *Ap, int *Aj, T *Ax,
                      int64_t capacity, int64_t *nnz_host, void *stream)
{
    if (rows < 0 || cols < 0 || nd < 0 || nd > kScanMaxCount || (nd > 0 && pitch < rows)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_dia_to_csr: bad size");
    if (!Ap || !nnz_host || (rows > 0 && nd > 0 && (!offsets || !vals))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_dia_to_csr: null array");
    hipStream_t s = as_stream(stream);
    int *counts = nullptr;
    CMI_HIP(hipMalloc((void **)&counts, (size_t)(rows > 0 ? rows : 1) * sizeof(int)));
    if (rows > 0) hipLaunchKernelGGL((dia_valid_counts_kernel<T>), dim3(grid_1d(rows)), dim3(256), 0, s, rows, cols, (int)nd, pitch, offsets, vals, counts);
    int st = exclusive_scan_i32(rows, counts, Ap, nnz_host, s);
    if (st == CMI_SUCCESS && Aj && Ax && *nnz_host > capacity) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_dia_to_csr: capacity is smaller than the entry count");
    if (st == CMI_SUCCESS && *nnz_host > 0 && Aj && Ax) {
        hipLaunchKernelGGL((dia_to_csr_kernel<T>), dim3(grid_1d(rows)), dim3(256), 0, s, rows, cols, (int)nd, pitch, offsets, vals, Ap, Aj, Ax);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = fail(CMI_ERROR_HIP, "cmi_dia_to_csr: scatter failed");
    }
    (void)hipFree(counts);
    return st;
}

// HYB -> CSR: a row's ELL entries, then its COO entries (the order hyb_matrix keeps: the first K entries of a row live in
// the ELL part, the rest in the COO part; reference conversions/hyb_to_other.h goes through COO and sorts by row)
__global__ void __launch_bounds__(256)
hyb_counts_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ ell_Aj, const int *__restrict__ coo_off, int *__restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int c = coo_off[i + 1] - coo_off[i];
        for (int k = 0; k < width; k++) c += ell_Aj[(int64_t)k * pitch + i] >= 0;
        counts[i] = c;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
hyb_to_csr_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ ell_Aj, const T *__restrict__ ell_Ax,
                  const int *__restrict__ coo_off, const int *__restrict__ coo_Aj, const T *__restrict__ coo_Ax,
                  const int *__restrict__ Ap, int *__restrict__ Aj, T *__restrict__ Ax)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int p = Ap[i];
        for (int k = 0; k < width; k++) {
            const int c = ell_Aj[(int64_t)k * pitch + i];
            if (c >= 0) { Aj[p] = c; Ax[p] = ell_Ax[(int64_t)k * pitch + i]; p++; }
        }
        for (int q = coo_off[i]; q < coo_off[i + 1]; q++, p++) { Aj[p] = coo_Aj[q]; Ax[p] = coo_Ax[q]; }
    }
}

template <typename T>
static int hyb_to_csr(int64_t rows, int64_t width, int64_t pitch, const int *eAj, const T *eAx, int64_t ncoo, const int *cAi,
                      const int *cAj, const T *cAx, int *Ap, int *Aj, T *Ax, int64_t capacity, int64_t *nnz_host, void *stream)
{
    if (rows < 0 || width < 0 || width > kScanMaxCount / 2 || ncoo < 0 || ncoo > INT32_MAX / 2 || (width > 0 && pitch < rows))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_to_csr: bad size");
    if (!Ap || !nnz_host || (rows > 0 && width > 0 && (!eAj || !eAx)) || (ncoo > 0 && (!cAi || !cAj || !cAx)))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_to_csr: null array");
    hipStream_t s = as_stream(stream);
    int *scratch = nullptr; // [rows + 1] COO row offsets | [rows] counts | order flag
    const size_t n1 = (size_t)rows + 1;
    CMI_HIP(hipMalloc((void **)&scratch, (2 * n1 + 1) * sizeof(int)));
    int *coo_off = scratch, *counts = scratch + n1, *flag = scratch + 2 * n1;
    int st = CMI_SUCCESS, unsorted = 0;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(coo_row_offsets_kernel, dim3(grid_1d(ncoo + 1)), dim3(256), 0, s, rows, ncoo, cAi, coo_off, flag);
        e = hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) st = hip_fail(e, "cmi_hyb_to_csr");
    else if (unsorted) st = fail(CMI_ERROR_NOT_SUPPORTED, "cmi_hyb_to_csr: the COO part is not sorted by row");
    if (st == CMI_SUCCESS) {
        if (rows > 0) hipLaunchKernelGGL(hyb_counts_kernel, dim3(grid_1d(rows)), dim3(256), 0, s, rows, (int)width, pitch, eAj, coo_off, counts);
        st = exclusive_scan_i32(rows, counts, Ap, nnz_host, s);
    }
    if (st == CMI_SUCCESS && Aj && Ax && *nnz_host > capacity) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_to_csr: capacity is smaller than the entry count");
    if (st == CMI_SUCCESS && *nnz_host > 0 && Aj && Ax) {
        hipLaunchKernelGGL((hyb_to_csr_kernel<T>), dim3(grid_1d(rows)), dim3(256), 0, s, rows, (int)width, pitch, eAj, eAx, coo_off, cAj, cAx, Ap, Aj, Ax);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = fail(CMI_ERROR_HIP, "cmi_hyb_to_csr: scatter failed");
    }
    (void)hipFree(scratch);
    return st;
}

} // namespace cmi

// Two-call protocol: first with Aj == Ax == NULL (Ap and *num_entries_host are filled: size the arrays), then with the arrays
// and their capacity.  (The second call recomputes the offsets: setup-time code, simplicity over speed.)
CMI_API int cmi_ell_to_csr_f64(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj, const double *ell_Ax, int32_t *Ap,
                               int32_t *Aj, double *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::ell_to_csr<double>(num_rows, width, pitch, ell_Aj, ell_Ax, Ap, Aj, Ax, capacity, num_entries_host, stream); }
CMI_API int cmi_ell_to_csr_f32(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj, const float *ell_Ax, int32_t *Ap,
                               int32_t *Aj, float *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::ell_to_csr<float>(num_rows, width, pitch, ell_Aj, ell_Ax, Ap, Aj, Ax, capacity, num_entries_host, stream); }
CMI_API int cmi_dia_to_csr_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch, const int32_t *offsets,
                               const double *values, int32_t *Ap, int32_t *Aj, double *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::dia_to_csr<double>(num_rows, num_cols, num_diagonals, pitch, offsets, values, Ap, Aj, Ax, capacity, num_entries_host, stream); }
CMI_API int cmi_dia_to_csr_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch, const int32_t *offsets,
                               const float *values, int32_t *Ap, int32_t *Aj, float *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::dia_to_csr<float>(num_rows, num_cols, num_diagonals, pitch, offsets, values, Ap, Aj, Ax, capacity, num_entries_host, stream); }
CMI_API int cmi_hyb_to_csr_f64(int64_t num_rows, int64_t ell_width, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                               int64_t coo_entries, const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, int32_t *Ap,
                               int32_t *Aj, double *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::hyb_to_csr<double>(num_rows, ell_width, ell_pitch, ell_Aj, ell_Ax, coo_entries, coo_Ai, coo_Aj, coo_Ax, Ap, Aj, Ax, capacity, num_entries_host, stream); }
CMI_API int cmi_hyb_to_csr_f32(int64_t num_rows, int64_t ell_width, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                               int64_t coo_entries, const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, int32_t *Ap,
                               int32_t *Aj, float *Ax, int64_t capacity, int64_t *num_entries_host, void *stream)
{ return cmi::hyb_to_csr<float>(num_rows, ell_width, ell_pitch, ell_Aj, ell_Ax, coo_entries, coo_Ai, coo_Aj, coo_Ax, Ap, Aj, Ax, capacity, num_entries_host, stream); }

// ---------------------------------------------------------------------------------------------
// HYB ELL-width cutoff under the tuned rule (cmi_tuning_hyb_rule): histogram of the row lengths on the device,
// the reference's threshold search (format_utils.inl:281-325: cumulative histogram + find_if with
// speed_threshold_functor, functional.inl:114-132) on the host.  Row lengths of kHybCap or more share the last bin: a
// width beyond that is never returned (an ELL part 4096 slots wide is no ELL part).
// ---------------------------------------------------------------------------------------------
namespace cmi {
constexpr int kHybCap = 4096;
__global__ void __launch_bounds__(256) row_length_histogram_kernel(int64_t num_rows, const int *__restrict__ Ap, unsigned int *__restrict__ hist)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        int len = Ap[i + 1] - Ap[i];
        len = len < 0 ? 0 : (len > kHybCap ? kHybCap : len);
        atomicAdd(hist + len, 1u);
    }
}
} // namespace cmi

CMI_API int cmi_hyb_entries_per_row(int dtype, int64_t num_rows, const int32_t *Ap, int kind, double relative_speed,
                                    int64_t threshold, int64_t *width_host, void *stream)
{
    if (dtype < 0 || dtype > 1 || num_rows < 0 || !width_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_entries_per_row: bad argument");
    *width_host = 0;
    if (num_rows == 0) return CMI_SUCCESS;
    if (!Ap) return fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_entries_per_row: null row offsets");
    if (kind < 0) {
        const int st = cmi_tuning_hyb_rule(dtype, &kind, &relative_speed, &threshold);
        if (st) return st;
    }
    if (kind < CMI_HYB_RULE_REFERENCE || kind > CMI_HYB_RULE_COST2 || !(relative_speed > 0.0) || threshold < 0)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_hyb_entries_per_row: bad rule");
    double light_speed = relative_speed;
    if (kind == CMI_HYB_RULE_COST2) {
        const int st = cmi_tuning_hyb_light_speed(dtype, &light_speed);
        if (st) return st;
    }
    hipStream_t s = as_stream(stream);
    unsigned int *dev = nullptr;
    const size_t bytes = (size_t)(kHybCap + 1) * sizeof(unsigned int);
    CMI_HIP(hipMalloc((void **)&dev, bytes));
    std::vector<unsigned int> hist(kHybCap + 1);
    hipError_t e = hipMemsetAsync(dev, 0, bytes, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(row_length_histogram_kernel, dim3(grid_1d(num_rows)), dim3(256), 0, s, num_rows, Ap, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hist.data(), dev, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "cmi_hyb_entries_per_row");
    int max_len = 0;
    for (int k = 0; k <= kHybCap; k++) if (hist[k]) max_len = k;
    int64_t K = max_len;
    if (kind == CMI_HYB_RULE_REFERENCE) {
        // the smallest k with  relative_speed * #{rows longer than k} < num_rows  or  #{rows longer than k} < threshold
        // (float arithmetic as the reference's functor, so that (3.0, 4096) reproduces its widths exactly)
        int64_t cum = 0;
        for (int k = 0; k < max_len; k++) {
            cum += hist[k];
            const int64_t longer = num_rows - cum;
            if ((float)relative_speed * (float)longer < (float)num_rows || longer < threshold) { K = k; break; }
        }
    } else {
        // argmin over k of  num_rows * k + [coo(k) > 0] * (threshold + relative_speed * coo(k)),  coo(k) = sum over rows of
        // max(0, len - k): walked from the longest row down (coo(max_len) = 0; coo(k) = coo(k+1) + #{rows longer than k})
        double best = (double)num_rows * (double)max_len;
        int64_t longer = 0;
        double coo = 0.0;
        for (int k = max_len - 1; k >= 0; k--) {
            longer += hist[k + 1]; // rows of length > k
            coo += (double)longer;
            // COST2: while the COO part is light enough for the one-launch kernel (the plan's own limit, kHybFusedMaxPerRow entries
            // per row) an entry costs light_speed slots and there is no second launch
            const bool light = kind == CMI_HYB_RULE_COST2 && coo <= kHybFusedMaxPerRow * (double)num_rows;
            const double cost = (double)num_rows * (double)k + (light ? light_speed * coo : (double)threshold + relative_speed * coo);
            if (cost < best) { best = cost; K = k; } // strict: ties go to the wider ELL part (fewer launches)
        }
    }
    *width_host = K;
    return CMI_SUCCESS;
}

// explicit zeros among n values (reference csr_to_other.h:188)
namespace cmi {
template <typename T>
__global__ void __launch_bounds__(256) count_zeros_kernel(int64_t n, const T *__restrict__ v, unsigned long long *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += v[i] == T(0);
    const unsigned long long m = __ballot(1);
    (void)m;
    for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(out, c);
}
template <typename T> static int count_zeros(int64_t n, const T *v, int64_t *count_host, void *stream)
{
    if (n < 0 || !count_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_count_zeros: bad argument");
    *count_host = 0;
    if (n == 0) return CMI_SUCCESS;
    if (!v) return fail(CMI_ERROR_INVALID_VALUE, "cmi_count_zeros: null array");
    hipStream_t s = as_stream(stream);
    unsigned long long *dev = nullptr, host = 0;
    CMI_HIP(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(host), s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(n, 256 * 8);
        if (blocks > kCus * 16) blocks = kCus * 16;
        hipLaunchKernelGGL((count_zeros_kernel<T>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, n, v, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "cmi_count_zeros");
    *count_host = (int64_t)host;
    return CMI_SUCCESS;
}
} // namespace cmi
CMI_API int cmi_count_zeros_f64(int64_t n, const double *values, int64_t *count_host, void *stream) { return cmi::count_zeros<double>(n, values, count_host, stream); }
CMI_API int cmi_count_zeros_f32(int64_t n, const float *values, int64_t *count_host, void *stream) { return cmi::count_zeros<float>(n, values, count_host, stream); }
