// spmv_csr.hip -- CSR SpMV for gfx950 (MI355X): y = A*x or y += A*x, int32 indices, f64/f32 values.
//
// Replaces (reference tree): cusp/system/cuda/detail/multiply/csr_scalar.h:51-109,
// csr_vector_spmv.h:71-258 and the KTT csr_spmv kernel family
// (cusp/system/cuda/ktt/kernels/csr_kernel.h:160-410).  Arithmetic contract: the host loop of
// cusp/system/detail/sequential/multiply/csr_spmv.h:42-74.
//
// Three hand-written variants, chosen per matrix shape by the persisted tuning table:
//   csr_scalar : one lane per row.
//   csr_vector : TPR = 2..64 lanes per row, lane-strided accumulate, DPP/shuffle reduction inside the
//                64-wide wave (no LDS, no implicit warp-synchronous code).
//   csr_stream : the bandwidth kernel for short rows (5-pt Poisson: 5 entries/row).  A workgroup owns
//                a contiguous run of rows; their column-index and value streams are read from HBM as
//                fully coalesced 16-byte-per-lane vectors, multiplied with the gathered x entries and
//                parked in LDS; then one lane per row adds its products in STORAGE ORDER.  Built with
//                -ffp-contract=off the per-row arithmetic (init, then multiply and add, in order) is
//                the reference host loop's, so csr_scalar and csr_stream are bit-identical to it.
//
// SpMV is HBM-bound (0.125 flop/byte): no MFMA.  Algorithmic bytes per call:
//   12*nnz + 20*num_rows + 4   (Ap once, Aj once, Ax once, x once, y once; f64).
#include "common.h"

namespace cmi {

// ---------------------------------------------------------------------------------------------
// csr_scalar
// ---------------------------------------------------------------------------------------------
template <typename T, bool NT>
__global__ void __launch_bounds__(1024)
csr_scalar_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj,
                  const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int accumulate)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < num_rows; row += stride) {
        const int s = Ap[row], e = Ap[row + 1];
        T acc = accumulate ? y[row] : T(0);
        for (int jj = s; jj < e; jj++) acc = acc + ld<NT>(Ax + jj) * x[ld<NT>(Aj + jj)];
        y[row] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// csr_vector<TPR>
// ---------------------------------------------------------------------------------------------
template <typename T, int TPR, bool NT>
__global__ void __launch_bounds__(1024)
csr_vector_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj,
                  const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int accumulate)
{
    const int lane = threadIdx.x & (TPR - 1);
    const int64_t nvec = (int64_t)gridDim.x * blockDim.x / TPR;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TPR; row < num_rows; row += nvec) {
        const int s = Ap[row], e = Ap[row + 1];
        T sum = T(0);
        for (int jj = s + lane; jj < e; jj += TPR) sum = sum + ld<NT>(Ax + jj) * x[ld<NT>(Aj + jj)];
        // butterfly inside the TPR-lane group; every lane of a group runs the same trip count
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sum = sum + __shfl_down(sum, o, TPR);
        if (lane == 0) y[row] = accumulate ? y[row] + sum : sum;
    }
}

// ---------------------------------------------------------------------------------------------
// csr_stream
// ---------------------------------------------------------------------------------------------
// Tile = blockDim.x * IPT * 4 entries.  LDS: T prod[tile] then int rowptr[rows_per_block + 1].
// VEC: Aj and Ax are 16-byte aligned, so entry index e with e % 4 == 0 is a 16-byte boundary in Aj
// and a 32-byte boundary in Ax (f64) / 16-byte (f32): one int4 + two double2 (or one float4) per lane.
template <typename T, int IPT, bool VEC, bool NT>
__global__ void __launch_bounds__(1024)
csr_stream_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap,
                  const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x,
                  T *__restrict__ y, int rows_per_block, int64_t num_tiles, int64_t tiles_per_xcd,
                  int swizzle, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int block = blockDim.x;
    const int tid = threadIdx.x;
    const int tile_entries = block * IPT * 4;
    T *prod = reinterpret_cast<T *>(smem);
    int *rowptr = reinterpret_cast<int *>(smem + (size_t)tile_entries * sizeof(T));

    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle != 0);
    if (tile >= num_tiles) return; // whole workgroup leaves together: no barrier is skipped by a part
    const int64_t r0 = tile * rows_per_block;
    const int nr = (int)((num_rows - r0) < rows_per_block ? (num_rows - r0) : rows_per_block);

    for (int i = tid; i <= nr; i += block) rowptr[i] = Ap[r0 + i];
    __syncthreads();
    const int nz0 = rowptr[0], nz1 = rowptr[nr];

    // a lane owns rows tid, tid+block, ... (at most 4: rows_per_block <= 4*block, host-checked)
    T acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = tid + q * block;
        acc[q] = (accumulate && r < nr) ? y[r0 + r] : T(0);
    }

    for (int base = VEC ? (nz0 & ~3) : nz0; base < nz1; base += tile_entries) {
        // ---- phase 1: stream Aj/Ax, gather x, park products in LDS ----
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            if constexpr (VEC) {
                const int slot = (k * block + tid) * 4;
                const int e = base + slot;
                if (e < nz1) {
                    T p0, p1, p2, p3;
                    if ((int64_t)e + 4 <= num_entries) {
                        const int4v c = ld<NT>(reinterpret_cast<const int4v *>(Aj + e));
                        if constexpr (sizeof(T) == 8) {
                            const double2v v01 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                            const double2v v23 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
                            p0 = v01.x * x[c.x]; p1 = v01.y * x[c.y];
                            p2 = v23.x * x[c.z]; p3 = v23.y * x[c.w];
                        } else {
                            const float4v v = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
                            p0 = v.x * x[c.x]; p1 = v.y * x[c.y];
                            p2 = v.z * x[c.z]; p3 = v.w * x[c.w];
                        }
                    } else { // last (partial) vector of the arrays
                        p0 = (int64_t)e + 0 < num_entries ? Ax[e + 0] * x[Aj[e + 0]] : T(0);
                        p1 = (int64_t)e + 1 < num_entries ? Ax[e + 1] * x[Aj[e + 1]] : T(0);
                        p2 = (int64_t)e + 2 < num_entries ? Ax[e + 2] * x[Aj[e + 2]] : T(0);
                        p3 = (int64_t)e + 3 < num_entries ? Ax[e + 3] * x[Aj[e + 3]] : T(0);
                    }
                    prod[slot + 0] = p0; prod[slot + 1] = p1; prod[slot + 2] = p2; prod[slot + 3] = p3;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int slot = (k * 4 + i) * block + tid;
                    const int e = base + slot;
                    if (e < nz1) prod[slot] = ld<NT>(Ax + e) * x[ld<NT>(Aj + e)];
                }
            }
        }
        __syncthreads();
        // ---- phase 2: one lane per row, products added in storage order ----
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = tid + q * block;
            if (r < nr) {
                int a = rowptr[r], b = rowptr[r + 1];
                a = a > base ? a : base;
                b = b < base + tile_entries ? b : base + tile_entries;
                T s = acc[q];
                for (int j = a; j < b; j++) s = s + prod[j - base];
                acc[q] = s;
            }
        }
        if (base + tile_entries < nz1) __syncthreads(); // another pass reuses prod (uniform condition)
    }

#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = tid + q * block;
        if (r < nr) y[r0 + r] = acc[q];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(int64_t work_items, int block, int items_per_block_thread = 1)
{
    // memory-bound grid-stride kernels: cap at 8 workgroups of 256 threads per CU (guide G11)
    int64_t blocks = ceil_div(work_items, (int64_t)block * items_per_block_thread);
    const int64_t cap = (int64_t)kCus * 8 * 256 / block;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <typename T, bool NT>
static int launch_vector(int tpr, int grid, int block, hipStream_t s, int64_t rows, const int *Ap, const int *Aj,
                         const T *Ax, const T *x, T *y, int acc)
{
    switch (tpr) {
    case 2:  hipLaunchKernelGGL((csr_vector_kernel<T, 2, NT>),  dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    case 4:  hipLaunchKernelGGL((csr_vector_kernel<T, 4, NT>),  dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    case 8:  hipLaunchKernelGGL((csr_vector_kernel<T, 8, NT>),  dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    case 16: hipLaunchKernelGGL((csr_vector_kernel<T, 16, NT>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    case 32: hipLaunchKernelGGL((csr_vector_kernel<T, 32, NT>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    case 64: hipLaunchKernelGGL((csr_vector_kernel<T, 64, NT>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, acc); break;
    default: return fail(CMI_ERROR_NOT_SUPPORTED, "csr_vector: threads_per_row must be 2,4,8,16,32 or 64");
    }
    return CMI_SUCCESS;
}

template <typename T, int IPT, bool VEC, bool NT>
static void launch_stream_one(int grid, int block, size_t lds, hipStream_t s, int64_t rows, int64_t nnz,
                              const int *Ap, const int *Aj, const T *Ax, const T *x, T *y, int rpb, int64_t tiles,
                              int64_t tpx, int swz, int acc)
{
    hipLaunchKernelGGL((csr_stream_kernel<T, IPT, VEC, NT>), dim3(grid), dim3(block), lds, s, rows, nnz, Ap, Aj, Ax,
                       x, y, rpb, tiles, tpx, swz, acc);
}

template <typename T, bool VEC, bool NT>
static int launch_stream_ipt(int ipt, int grid, int block, size_t lds, hipStream_t s, int64_t rows, int64_t nnz,
                             const int *Ap, const int *Aj, const T *Ax, const T *x, T *y, int rpb, int64_t tiles,
                             int64_t tpx, int swz, int acc)
{
    switch (ipt) {
    case 1: launch_stream_one<T, 1, VEC, NT>(grid, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, acc); break;
    case 2: launch_stream_one<T, 2, VEC, NT>(grid, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, acc); break;
    case 4: launch_stream_one<T, 4, VEC, NT>(grid, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, acc); break;
    default: return fail(CMI_ERROR_NOT_SUPPORTED, "csr_stream: items_per_thread must be 1, 2 or 4");
    }
    return CMI_SUCCESS;
}

template <typename T>
static int spmv_csr(int dtype, int64_t rows, int64_t cols, int64_t nnz, const int *Ap, const int *Aj, const T *Ax,
                    const T *x, T *y, int accumulate, const cmi_config *user, void *stream)
{
    if (rows < 0 || cols < 0 || nnz < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX || nnz > INT32_MAX)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: sizes exceed the int32 index type");
    if (rows == 0) return CMI_SUCCESS;
    if (!Ap || !y || (nnz > 0 && (!Aj || !Ax || !x)))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: null array");
    cmi_config c;
    select_config(CMI_FORMAT_CSR, dtype, rows, cols, nnz, user, &c);
    hipStream_t s = as_stream(stream);
    const int block = c.block_size;
    const bool nt = c.nontemporal != 0;

    switch (c.kernel) {
    case CMI_CSR_SCALAR: {
        const int grid = grid_for(rows, block);
        if (nt) hipLaunchKernelGGL((csr_scalar_kernel<T, true>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, accumulate);
        else    hipLaunchKernelGGL((csr_scalar_kernel<T, false>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, accumulate);
        break;
    }
    case CMI_CSR_VECTOR: {
        const int tpr = c.threads_per_row;
        const int grid = grid_for(rows * tpr, block);
        int st = nt ? launch_vector<T, true>(tpr, grid, block, s, rows, Ap, Aj, Ax, x, y, accumulate)
                    : launch_vector<T, false>(tpr, grid, block, s, rows, Ap, Aj, Ax, x, y, accumulate);
        if (st) return st;
        break;
    }
    case CMI_CSR_STREAM: {
        const int rpb = c.rows_per_block;
        const int ipt = c.items_per_thread;
        if (rpb < 1 || rpb > 4 * block) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: rows_per_block must be in [1, 4*block_size]");
        const int64_t tiles = ceil_div(rows, rpb);
        const int64_t tpx = ceil_div(tiles, kXcds);
        const int swz = c.xcd_swizzle != 0;
        const int64_t grid64 = swz ? tpx * kXcds : tiles;
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: grid too large");
        const size_t lds = (size_t)block * ipt * 4 * sizeof(T) + (size_t)(rpb + 1) * sizeof(int);
        if (lds > 160 * 1024) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: tile does not fit 160 KiB of LDS");
        const bool vec = (reinterpret_cast<uintptr_t>(Aj) % 16 == 0) && (reinterpret_cast<uintptr_t>(Ax) % 16 == 0);
        int st;
        if (vec) st = nt ? launch_stream_ipt<T, true, true>(ipt, (int)grid64, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, accumulate)
                         : launch_stream_ipt<T, true, false>(ipt, (int)grid64, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, accumulate);
        else     st = nt ? launch_stream_ipt<T, false, true>(ipt, (int)grid64, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, accumulate)
                         : launch_stream_ipt<T, false, false>(ipt, (int)grid64, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, accumulate);
        if (st) return st;
        break;
    }
    default: return fail(CMI_ERROR_NOT_SUPPORTED, "cmi_spmv_csr: config.kernel is not a CSR kernel");
    }
    CMI_LAUNCH_CHECK("csr spmv");
    return CMI_SUCCESS;
}

} // namespace cmi

CMI_API int cmi_spmv_csr_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                             const int32_t *Aj, const double *Ax, const double *x, double *y, int accumulate,
                             const cmi_config *cfg, void *stream)
{
    return cmi::spmv_csr<double>(CMI_F64, num_rows, num_cols, num_entries, Ap, Aj, Ax, x, y, accumulate, cfg, stream);
}

CMI_API int cmi_spmv_csr_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                             const int32_t *Aj, const float *Ax, const float *x, float *y, int accumulate,
                             const cmi_config *cfg, void *stream)
{
    return cmi::spmv_csr<float>(CMI_F32, num_rows, num_cols, num_entries, Ap, Aj, Ax, x, y, accumulate, cfg, stream);
}
