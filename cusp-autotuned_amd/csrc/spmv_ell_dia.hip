// spmv_ell_dia.hip -- ELL (+ELLR) and DIA SpMV for gfx950.
//
// Replaces (reference tree): cusp/system/cuda/detail/multiply/ell_spmv.h:55-155,
// dia_spmv.h:69-188 and the KTT kernels cusp/system/cuda/ktt/kernels/ell_kernel.h:86-213,
// dia_kernel.h:129-252.  Arithmetic contracts: sequential/multiply/ell_spmv.h:41-76 and
// dia_spmv.h:43-82 (y = init(y); then slot-major / diagonal-major accumulation: per row that is
// "start from init, add slot 0, slot 1, ..." -- exactly what one lane per row does here, so with
// -ffp-contract=off both kernels are bit-identical to the host loops).
//
// Both formats are column-major with leading dimension `pitch`: lane i of a wave reads element
// (row0+i, n), i.e. 64 consecutive addresses -- fully coalesced 256/512-byte wave loads.  RPL = rows
// per lane (1 or 2): with 2, a lane reads int2 / double2 so each wave instruction moves 512 B / 1 KiB.
//
// Algorithmic bytes per call (f64):
//   ELL: width*pitch*(4+8) + 16*num_rows        DIA: ndiag*pitch*8 + 4*ndiag + 16*num_rows
#include "common.h"
#include <cstdlib>

namespace cmi {

// $CMI_DOT_SWIZZLE: XCD dealing of the fused <y, w> instances with a table shape (measurements; unset: the table's dealing)
static int dot_swizzle_env(int table_swizzle)
{
    static const int env = [] { const char *e = std::getenv("CMI_DOT_SWIZZLE"); return e ? std::atoi(e) : -1; }();
    return env >= 0 ? env : table_swizzle;
}

// ---------------------------------------------------------------------------------------------
// ELL: one lane per row (RPL rows per lane)
// ---------------------------------------------------------------------------------------------
// Memory-level parallelism: slots are processed U at a time -- all U column loads, all U value loads
// and all U x gathers are ISSUED before the first product is added (padding slots gather x[0] and are
// masked out of the sum), so a lane has up to 3U loads in flight instead of walking a
// load -> wait -> branch -> gather -> wait chain per slot.  The adds then run in slot order, skipping
// padding exactly as the host loop does: same bits.

// DOT: the workgroup also leaves sum_r y[r] * w[r] over its rows in dot_partial[blockIdx.x] (see
// spmv_csr.hip csr_stream DOT: the CG step <A p, p> without a second pass over y).
template <typename T, int RPL, bool ELLR, int POL, bool DOT = false>
__global__ void __launch_bounds__(1024)
ell_row_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ Aj, const T *__restrict__ Ax,
               const int *__restrict__ row_lengths, const T *__restrict__ x, T *__restrict__ y, int accumulate,
               int64_t tiles, int64_t tiles_per_xcd, int swizzle,
               const T *__restrict__ w = nullptr, double *__restrict__ dot_partial = nullptr)
{
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    double dsum = 0.0;
    // A tile = the blockDim.x * RPL consecutive rows of one workgroup (one-shot grid).  Tiles are dealt to the XCDs in
    // chunks (`swizzle`: common.h tile_of_block) so that the x window a chunk gathers -- its rows +- the matrix
    // bandwidth -- is fetched into ONE L2; in launch order x is fetched ~2.5x on the headline matrix
    // (archive/profiles/r02_formats_pmc.json).  The grid is padded to whole chunk rounds: a workgroup whose tile lies past the
    // end leaves as a whole.
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= tiles) return;
    const int64_t row = (tile * blockDim.x + threadIdx.x) * RPL;
    if (row < num_rows) {
        if constexpr (RPL == 1) {
            T acc = accumulate ? y[row] : T(0);
            const int len = ELLR ? row_lengths[row] : width; // ELLR: valid leading slots of THIS row
            // K slots at a time: all column / value loads, then all gathers, then the adds in slot order
            auto chunk = [&](auto Kc, int n0) {
                constexpr int K = decltype(Kc)::value;
                int col[K];
                T val[K], xv[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    col[k] = ld<NT>(Aj + (n0 + k) * pitch + row);
                    val[k] = ld<NT>(Ax + (n0 + k) * pitch + row);
                }
#pragma unroll
                for (int k = 0; k < K; k++) xv[k] = x[col[k] < 0 ? 0 : col[k]];
                __builtin_amdgcn_sched_barrier(0); // keep every load above issued before the first add
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (ELLR ? (n0 + k < len) : (col[k] != -1)) acc = acc + val[k] * xv[k];
            };
            int n0 = 0; // width is wave-uniform: full chunks of 8, then a 4 / 2 / 1 tail -- no wasted loads
            for (; n0 + 8 <= width; n0 += 8) chunk(std::integral_constant<int, 8>(), n0);
            if (n0 + 4 <= width) { chunk(std::integral_constant<int, 4>(), n0); n0 += 4; }
            if (n0 + 2 <= width) { chunk(std::integral_constant<int, 2>(), n0); n0 += 2; }
            if (n0 + 1 <= width) { chunk(std::integral_constant<int, 1>(), n0); }
            st<NTS>(y + row, acc);
            if constexpr (DOT) dsum += (double)acc * (double)w[row];
        } else {
            // rows row, row+1 (pitch even and arrays 16-byte aligned: checked by the launcher)
            typedef typename vec2<T>::type T2;
            const bool has1 = row + 1 < num_rows;
            T acc0 = accumulate ? y[row] : T(0);
            T acc1 = (accumulate && has1) ? y[row + 1] : T(0);
            int len0 = width, len1 = width;
            if constexpr (ELLR) { len0 = row_lengths[row]; len1 = has1 ? row_lengths[row + 1] : 0; }
            auto chunk = [&](auto Kc, int n0) {
                constexpr int K = decltype(Kc)::value;
                int2v c[K];
                T2 v[K];
                T x0[K], x1[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    c[k] = ld<NT>(reinterpret_cast<const int2v *>(Aj + (n0 + k) * pitch + row));
                    v[k] = ld<NT>(reinterpret_cast<const T2 *>(Ax + (n0 + k) * pitch + row));
                }
#pragma unroll
                for (int k = 0; k < K; k++) {
                    x0[k] = x[c[k].x < 0 ? 0 : c[k].x];
                    x1[k] = x[c[k].y < 0 ? 0 : c[k].y];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (ELLR ? (n0 + k < len0) : (c[k].x != -1)) acc0 = acc0 + v[k].x * x0[k];
                    if (ELLR ? (n0 + k < len1) : (c[k].y != -1 && has1)) acc1 = acc1 + v[k].y * x1[k];
                }
            };
            int n0 = 0;
            for (; n0 + 4 <= width; n0 += 4) chunk(std::integral_constant<int, 4>(), n0);
            if (n0 + 2 <= width) { chunk(std::integral_constant<int, 2>(), n0); n0 += 2; }
            if (n0 + 1 <= width) { chunk(std::integral_constant<int, 1>(), n0); }
            if (has1 && (reinterpret_cast<uintptr_t>(y + row) % (2 * sizeof(T)) == 0)) {
                T2 o; o.x = acc0; o.y = acc1;
                st<NTS>(reinterpret_cast<T2 *>(y + row), o); // one 16-byte (f64) store for the row pair
            } else {
                st<NTS>(y + row, acc0);
                if (has1) st<NTS>(y + row + 1, acc1);
            }
            if constexpr (DOT) {
                dsum += (double)acc0 * (double)w[row];
                if (has1) dsum += (double)acc1 * (double)w[row + 1];
            }
        }
    }
    if constexpr (DOT) {
        tile_dot_store(dsum, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---------------------------------------------------------------------------------------------
// ELL: S lanes per row ("slices") -- few rows, wide rows
// ---------------------------------------------------------------------------------------------
// One lane per row leaves the chip idle when the matrix has fewer rows than the chip has lanes (256 CUs x 8 waves x 64 =
// 131 072) and makes every lane walk `width` dependent chunks.  Here a workgroup owns R = blockDim.x / S consecutive rows
// and thread (ri = tid % R, s = tid / R) takes the slots s, s + S, s + 2S, ... of row ri: lanes that are neighbours in a
// wave are neighbours in a COLUMN of the column-major arrays (whole waves for R >= 64, 128-byte runs for R = 16), the
// partial sums meet in LDS and lane s = 0 adds them in slice order.  Replaces the THREADS_PER_ROW parameter of
// cusp/system/cuda/ktt/kernels/ell_kernel.h:102-109,165-173 (there: threadIdx.y and an LDS atomicAdd, so a row's sum order
// changes from run to run; here it is fixed, but it is not the storage order: 1e-6 class like csr_vector).
template <typename T, int S, bool ELLR, int POL>
__global__ void __launch_bounds__(1024)
ell_slices_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ Aj, const T *__restrict__ Ax,
                  const int *__restrict__ row_lengths, const T *__restrict__ x, T *__restrict__ y, int accumulate)
{
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    __shared__ T part[1024];
    const int R = blockDim.x / S;
    const int ri = threadIdx.x % R, s = threadIdx.x / R;
    const int64_t row = (int64_t)blockIdx.x * R + ri;
    const bool live = row < num_rows;
    T acc = T(0);
    if (live) {
        const int len = ELLR ? row_lengths[row] : width;
        auto chunk = [&](auto Kc, int n0) { // slots n0, n0 + S, ...: every load issued before the first add
            constexpr int K = decltype(Kc)::value;
            int col[K];
            T val[K], xv[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                col[k] = ld<NT>(Aj + (int64_t)(n0 + k * S) * pitch + row);
                val[k] = ld<NT>(Ax + (int64_t)(n0 + k * S) * pitch + row);
            }
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[col[k] < 0 ? 0 : col[k]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++)
                if (ELLR ? (n0 + k * S < len) : (col[k] != -1)) acc = acc + val[k] * xv[k];
        };
        const int bound = ELLR ? len : width; // ELLR: slots past the row's length are never read
        int n = s;
        for (; n + 3 * S < bound; n += 4 * S) chunk(std::integral_constant<int, 4>(), n);
        for (; n < bound; n += S) chunk(std::integral_constant<int, 1>(), n);
    }
    part[s * R + ri] = acc;
    __syncthreads();
    if (s == 0 && live) {
        T sum = accumulate ? y[row] : T(0);
#pragma unroll
        for (int k = 0; k < S; k++) sum = sum + part[k * R + ri];
        st<NTS>(y + row, sum);
    }
}

// lanes per row for an ELL multiply: an explicit config.threads_per_row (1 = the row kernel, storage-order sums), else
// slices only where one lane per row cannot fill the chip and rows are wide enough to split
int ell_lanes_per_row(const cmi_config &c, int64_t rows, int64_t width)
{
    int t = c.threads_per_row;
    if (t <= 0) {
        t = 1;
        while (t < 16 && (int64_t)t * 2 * kEllSliceMinSlots <= width) t *= 2;
        if (t == 1 || rows >= (t == 2 ? kEllSliceMaxRows2 : kEllSliceMaxRows)) return 1;
    }
    int p = 1;
    while (p < t && p < 16) p <<= 1;
    while (p > 1 && p > width) p >>= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------
// DIA: one lane per row, diagonal offsets staged through LDS in chunks
// ---------------------------------------------------------------------------------------------
// Same memory-level-parallelism scheme as ELL: U diagonals at a time, all value loads and all x
// loads issued first (column clamped into range, out-of-range products masked), adds in diagonal
// order.
constexpr int kDiaChunk = 256;

template <typename T, int POL, bool DOT = false>
__global__ void __launch_bounds__(1024)
dia_row_kernel(int64_t num_rows, int64_t num_cols, int num_diagonals, int64_t pitch,
               const int *__restrict__ offsets, const T *__restrict__ vals, const T *__restrict__ x,
               T *__restrict__ y, int accumulate, int64_t tiles, int64_t tiles_per_xcd, int swizzle,
               const T *__restrict__ w = nullptr, double *__restrict__ dot_partial = nullptr)
{
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    __shared__ int soff[kDiaChunk];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    // tiles (one workgroup's rows) dealt to the XCDs in chunks: see ell_row_kernel.  Measured on the headline matrix
    // (archive/tools/r2_probe.hip, archive/profiles/r02_probe_*): launch order 98-101 us with x fetched 2.9x (651 MB read), chunks of 32
    // tiles 85 us with 498 MB read (compulsory: 480 MB).
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= tiles) return;
    const int64_t row = tile * blockDim.x + threadIdx.x;
    const bool live = row < num_rows;
    const int64_t lrow = live ? row : num_rows - 1; // dead lanes shadow the last row (loads stay valid)
    T acc = (accumulate && live) ? y[row] : T(0);
    for (int base = 0; base < num_diagonals; base += kDiaChunk) {
        const int nchunk = num_diagonals - base < kDiaChunk ? num_diagonals - base : kDiaChunk;
        if (base > 0) __syncthreads();
        for (int i = threadIdx.x; i < nchunk; i += blockDim.x) soff[i] = offsets[base + i];
        __syncthreads();
        auto chunk = [&](auto Kc, int d0) {
            constexpr int K = decltype(Kc)::value;
            T v[K], xv[K];
            bool ok[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t col = lrow + soff[d0 + k];
                ok[k] = col >= 0 && col < num_cols;
                v[k] = ld<NT>(vals + (int64_t)(base + d0 + k) * pitch + lrow);
                xv[k] = x[col < 0 ? 0 : (col >= num_cols ? num_cols - 1 : col)];
            }
            __builtin_amdgcn_sched_barrier(0); // keep every load above issued before the first add
#pragma unroll
            for (int k = 0; k < K; k++)
                if (ok[k]) acc = acc + v[k] * xv[k];
        };
        int d0 = 0; // nchunk is wave-uniform: full chunks of 8, then a 4 / 2 / 1 tail
        for (; d0 + 8 <= nchunk; d0 += 8) chunk(std::integral_constant<int, 8>(), d0);
        if (d0 + 4 <= nchunk) { chunk(std::integral_constant<int, 4>(), d0); d0 += 4; }
        if (d0 + 2 <= nchunk) { chunk(std::integral_constant<int, 2>(), d0); d0 += 2; }
        if (d0 + 1 <= nchunk) { chunk(std::integral_constant<int, 1>(), d0); }
    }
    if (live) st<NTS>(y + row, acc);
    if constexpr (DOT) {
        tile_dot_store(live ? (double)acc * (double)w[row] : 0.0, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// two rows per lane: values as T2 vectors; x for the second row is the neighbouring element
template <typename T, int POL, bool DOT = false>
__global__ void __launch_bounds__(1024)
dia_row2_kernel(int64_t num_rows, int64_t num_cols, int num_diagonals, int64_t pitch,
                const int *__restrict__ offsets, const T *__restrict__ vals, const T *__restrict__ x,
                T *__restrict__ y, int accumulate, int64_t tiles, int64_t tiles_per_xcd, int swizzle,
                const T *__restrict__ w = nullptr, double *__restrict__ dot_partial = nullptr)
{
    typedef typename vec2<T>::type T2;
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    __shared__ int soff[kDiaChunk];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= tiles) return;
    const int64_t row = (tile * blockDim.x + threadIdx.x) * 2;
    const bool live0 = row < num_rows, live1 = row + 1 < num_rows;
    // dead lanes shadow the last even row pair that is fully inside the pitch (loads stay valid)
    const int64_t lrow = live0 ? row : ((num_rows - 1) & ~(int64_t)1);
    T acc0 = (accumulate && live0) ? y[row] : T(0);
    T acc1 = (accumulate && live1) ? y[row + 1] : T(0);
    for (int base = 0; base < num_diagonals; base += kDiaChunk) {
        const int nchunk = num_diagonals - base < kDiaChunk ? num_diagonals - base : kDiaChunk;
        if (base > 0) __syncthreads();
        for (int i = threadIdx.x; i < nchunk; i += blockDim.x) soff[i] = offsets[base + i];
        __syncthreads();
        auto chunk = [&](auto Kc, int d0) {
            constexpr int K = decltype(Kc)::value;
            T2 v[K];
            T x0[K], x1[K];
            bool ok0[K], ok1[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t c0 = lrow + soff[d0 + k], c1 = c0 + 1;
                ok0[k] = live0 && c0 >= 0 && c0 < num_cols;
                ok1[k] = live1 && c1 >= 0 && c1 < num_cols;
                v[k] = ld<NT>(reinterpret_cast<const T2 *>(vals + (int64_t)(base + d0 + k) * pitch + lrow));
                x0[k] = x[c0 < 0 ? 0 : (c0 >= num_cols ? num_cols - 1 : c0)];
                x1[k] = x[c1 < 0 ? 0 : (c1 >= num_cols ? num_cols - 1 : c1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (ok0[k]) acc0 = acc0 + v[k].x * x0[k];
                if (ok1[k]) acc1 = acc1 + v[k].y * x1[k];
            }
        };
        int d0 = 0;
        for (; d0 + 4 <= nchunk; d0 += 4) chunk(std::integral_constant<int, 4>(), d0);
        if (d0 + 2 <= nchunk) { chunk(std::integral_constant<int, 2>(), d0); d0 += 2; }
        if (d0 + 1 <= nchunk) { chunk(std::integral_constant<int, 1>(), d0); }
    }
    if (live1 && (reinterpret_cast<uintptr_t>(y + row) % (2 * sizeof(T)) == 0)) {
        T2 o; o.x = acc0; o.y = acc1;
        st<NTS>(reinterpret_cast<T2 *>(y + row), o);
    } else {
        if (live0) st<NTS>(y + row, acc0);
        if (live1) st<NTS>(y + row + 1, acc1);
    }
    if constexpr (DOT) {
        double d = live0 ? (double)acc0 * (double)w[row] : 0.0;
        if (live1) d += (double)acc1 * (double)w[row + 1];
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename T>
static int spmv_ell(int dtype, int64_t rows, int64_t cols, int64_t width, int64_t pitch, const int *Aj, const T *Ax,
                    const int *row_lengths, const T *x, T *y, int accumulate, const cmi_config *user, void *stream,
                    const T *wdot = nullptr, double *dot_partial = nullptr, int *dot_partials = nullptr)
{
    if (dot_partials) *dot_partials = 0; // > 0: the kernel left that many partials of <y, wdot> in dot_partial
    if (rows < 0 || cols < 0 || width < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX || width > INT32_MAX)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: sizes exceed the int32 index type");
    // reference: throws when pitch < minor dimension (cusp/detail/array2d.inl:43-44)
    if (width > 0 && pitch < rows) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: pitch < num_rows");
    if (rows == 0) return CMI_SUCCESS;
    if (!y || (width > 0 && (!Aj || !Ax || !x))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: null array");
    cmi_config c;
    select_config(CMI_FORMAT_ELL, dtype, rows, cols, rows * width, user, &c);
    if (c.kernel != CMI_ELL_ROW) return fail(CMI_ERROR_NOT_SUPPORTED, "cmi_spmv_ell: config.kernel is not an ELL kernel");
    hipStream_t s = as_stream(stream);
    int block = c.block_size;
    const int pol = c.nontemporal & 3;
    const bool ellr = row_lengths != nullptr;
    int rpl = c.items_per_thread >= 2 ? 2 : 1;
    // two rows per lane needs 8-byte aligned int2 and 2*sizeof(T)-aligned value pairs in every slot
    if (rpl == 2 && !(pitch % 2 == 0 && reinterpret_cast<uintptr_t>(Aj) % 8 == 0 &&
                      reinterpret_cast<uintptr_t>(Ax) % (2 * sizeof(T)) == 0))
        rpl = 1;
    // a fused dot leaves one partial per workgroup: widen the workgroups until they fit the workspace
    if (wdot && dot_partial)
        while (block < 1024 && ceil_div(rows, (int64_t)block * rpl) > kPartialCapacity) block *= 2;
    const int lanes = ell_lanes_per_row(c, rows, width);
    if (lanes > 1) { // (a fused dot falls back to the separate dot: *dot_partials stays 0)
        const int R = block / lanes;
        const int64_t g64 = ceil_div(rows, (int64_t)R);
        if (g64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: grid too large");
        const int g = (int)g64, wd = (int)width;
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value;
                if (ellr) hipLaunchKernelGGL((ell_slices_kernel<T, S, true, POL>), dim3(g), dim3(block), 0, s, rows, wd, pitch, Aj, Ax, row_lengths, x, y, accumulate);
                else      hipLaunchKernelGGL((ell_slices_kernel<T, S, false, POL>), dim3(g), dim3(block), 0, s, rows, wd, pitch, Aj, Ax, row_lengths, x, y, accumulate);
            };
            switch (lanes) {
            case 2: go(std::integral_constant<int, 2>()); break;
            case 4: go(std::integral_constant<int, 4>()); break;
            case 8: go(std::integral_constant<int, 8>()); break;
            default: go(std::integral_constant<int, 16>()); break;
            }
        });
        CMI_LAUNCH_CHECK("ell slices spmv");
        return CMI_SUCCESS;
    }
    const int64_t tiles = ceil_div(rows, (int64_t)block * rpl); // one-shot grid (see spmv_csr.hip grid_for), padded to chunk rounds
    int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
    if (wdot && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO)) swz = dot_swizzle_env(swz); // (experiment knob: $CMI_DOT_SWIZZLE)
    const int64_t tpx = ceil_div(tiles, kXcds);
    const int64_t grid64 = padded_grid(tiles, swz);
    if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell: grid too large");
    const int grid = (int)grid64;
    const int w = (int)width;
#define CMI_ELL_LAUNCH(RPL, ELLR)                                                                             \
    with_policy(pol, [&](auto P) {                                                                                \
        hipLaunchKernelGGL((ell_row_kernel<T, RPL, ELLR, decltype(P)::value>), dim3(grid), dim3(block), 0, s, rows, w, \
                           pitch, Aj, Ax, row_lengths, x, y, accumulate, tiles, tpx, swz);                        \
    })
#define CMI_ELL_LAUNCH_DOT(RPL, ELLR)                                                                         \
    with_policy(pol, [&](auto P) {                                                                                \
        hipLaunchKernelGGL((ell_row_kernel<T, RPL, ELLR, decltype(P)::value, true>), dim3(grid), dim3(block), 0, s, rows, w, \
                           pitch, Aj, Ax, row_lengths, x, y, accumulate, tiles, tpx, swz, wdot, dot_partial);     \
    })
    const bool dot = wdot && dot_partial && tiles <= kPartialCapacity; // one partial per tile (a double, whatever T)
    {
        if (dot) {
            if (rpl == 1) { if (ellr) CMI_ELL_LAUNCH_DOT(1, true); else CMI_ELL_LAUNCH_DOT(1, false); }
            else          { if (ellr) CMI_ELL_LAUNCH_DOT(2, true); else CMI_ELL_LAUNCH_DOT(2, false); }
            CMI_LAUNCH_CHECK("ell spmv dot");
            if (dot_partials) *dot_partials = (int)tiles;
            return CMI_SUCCESS;
        }
    }
    if (rpl == 1) { if (ellr) CMI_ELL_LAUNCH(1, true); else CMI_ELL_LAUNCH(1, false); }
    else          { if (ellr) CMI_ELL_LAUNCH(2, true); else CMI_ELL_LAUNCH(2, false); }
#undef CMI_ELL_LAUNCH_DOT
#undef CMI_ELL_LAUNCH
    CMI_LAUNCH_CHECK("ell spmv");
    return CMI_SUCCESS;
}

template <typename T>
static int spmv_dia(int dtype, int64_t rows, int64_t cols, int64_t ndiag, int64_t pitch, const int *offsets,
                    const T *vals, const T *x, T *y, int accumulate, const cmi_config *user, void *stream,
                    const T *wdot = nullptr, double *dot_partial = nullptr, int *dot_partials = nullptr)
{
    if (dot_partials) *dot_partials = 0;
    if (rows < 0 || cols < 0 || ndiag < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX || ndiag > INT32_MAX)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia: sizes exceed the int32 index type");
    if (ndiag > 0 && pitch < rows) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia: pitch < num_rows");
    if (rows == 0) return CMI_SUCCESS;
    if (!y || (ndiag > 0 && (!offsets || !vals || !x))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia: null array");
    cmi_config c;
    select_config(CMI_FORMAT_DIA, dtype, rows, cols, rows * ndiag, user, &c);
    if (c.kernel != CMI_DIA_ROW) return fail(CMI_ERROR_NOT_SUPPORTED, "cmi_spmv_dia: config.kernel is not a DIA kernel");
    hipStream_t s = as_stream(stream);
    int block = c.block_size;
    const int pol = c.nontemporal & 3;
    int rpl = c.items_per_thread >= 2 ? 2 : 1;
    if (rpl == 2 && !(pitch % 2 == 0 && reinterpret_cast<uintptr_t>(vals) % (2 * sizeof(T)) == 0)) rpl = 1;
    if (wdot && dot_partial) // one partial per workgroup: widen the workgroups until they fit the workspace
        while (block < 1024 && ceil_div(rows, (int64_t)block * rpl) > kPartialCapacity) block *= 2;
    const int64_t tiles = ceil_div(rows, (int64_t)block * rpl);
    int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
    if (wdot && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO)) swz = dot_swizzle_env(swz); // (experiment knob: $CMI_DOT_SWIZZLE)
    const int64_t tpx = ceil_div(tiles, kXcds);
    const int64_t grid64 = padded_grid(tiles, swz);
    if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia: grid too large");
    const int grid = (int)grid64, nd = (int)ndiag;
    const bool dot = wdot && dot_partial && tiles <= kPartialCapacity;
    with_policy(pol, [&](auto P) {
        constexpr int POL = decltype(P)::value;
        {
            if (dot) {
                if (rpl == 1) hipLaunchKernelGGL((dia_row_kernel<T, POL, true>), dim3(grid), dim3(block), 0, s, rows, cols, nd, pitch, offsets, vals, x, y, accumulate, tiles, tpx, swz, wdot, dot_partial);
                else          hipLaunchKernelGGL((dia_row2_kernel<T, POL, true>), dim3(grid), dim3(block), 0, s, rows, cols, nd, pitch, offsets, vals, x, y, accumulate, tiles, tpx, swz, wdot, dot_partial);
                return;
            }
        }
        if (rpl == 1) hipLaunchKernelGGL((dia_row_kernel<T, POL>), dim3(grid), dim3(block), 0, s, rows, cols, nd, pitch, offsets, vals, x, y, accumulate, tiles, tpx, swz);
        else          hipLaunchKernelGGL((dia_row2_kernel<T, POL>), dim3(grid), dim3(block), 0, s, rows, cols, nd, pitch, offsets, vals, x, y, accumulate, tiles, tpx, swz);
    });
    CMI_LAUNCH_CHECK("dia spmv");
    if (dot && dot_partials) *dot_partials = (int)tiles;
    return CMI_SUCCESS;
}

} // namespace cmi

CMI_API int cmi_spmv_ell_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                             const int32_t *Aj, const double *Ax, const int32_t *row_lengths, const double *x,
                             double *y, int accumulate, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_ell<double>(CMI_F64, num_rows, num_cols, num_entries_per_row, pitch, Aj, Ax, row_lengths, x, y, accumulate, cfg, stream);
}
CMI_API int cmi_spmv_ell_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                             const int32_t *Aj, const float *Ax, const int32_t *row_lengths, const float *x,
                             float *y, int accumulate, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_ell<float>(CMI_F32, num_rows, num_cols, num_entries_per_row, pitch, Aj, Ax, row_lengths, x, y, accumulate, cfg, stream);
}
CMI_API int cmi_spmv_dia_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                             const int32_t *diagonal_offsets, const double *values, const double *x, double *y,
                             int accumulate, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_dia<double>(CMI_F64, num_rows, num_cols, num_diagonals, pitch, diagonal_offsets, values, x, y, accumulate, cfg, stream);
}
CMI_API int cmi_spmv_dia_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                             const int32_t *diagonal_offsets, const float *values, const float *x, float *y,
                             int accumulate, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_dia<float>(CMI_F32, num_rows, num_cols, num_diagonals, pitch, diagonal_offsets, values, x, y, accumulate, cfg, stream);
}

// y <- A x and *dot_dev <- <y, w> in one pass (see cmi_spmv_csr_dot_f64); when the launch shape cannot fuse the
// dot (more workgroups than the workspace holds partials), the plain SpMV followed by cmi_blas_dot_f64.
CMI_API int cmi_spmv_ell_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                                 const int32_t *Aj, const double *Ax, const int32_t *row_lengths, const double *x,
                                 double *y, const double *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    if ((!w && num_rows > 0) || !dot_dev || !workspace) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell_dot: null w, result or workspace");
    int partials = 0;
    const int st = cmi::spmv_ell<double>(CMI_F64, num_rows, num_cols, num_entries_per_row, pitch, Aj, Ax, row_lengths, x, y, 0, cfg, stream,
                                         w, (double *)workspace, &partials);
    if (st) return st;
    if (partials > 0) return cmi::reduce_partials_f64(partials, (double *)workspace, dot_dev, cmi::as_stream(stream));
    return cmi_blas_dot_f64(num_rows, y, w, dot_dev, workspace, stream);
}

// float matrices: the same, the scalar stays a double (partials are doubles; the fallback is cmi_blas_dotd_f32)
CMI_API int cmi_spmv_ell_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                                 const int32_t *Aj, const float *Ax, const int32_t *row_lengths, const float *x,
                                 float *y, const float *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    if ((!w && num_rows > 0) || !dot_dev || !workspace) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_ell_dot: null w, result or workspace");
    int partials = 0;
    const int st = cmi::spmv_ell<float>(CMI_F32, num_rows, num_cols, num_entries_per_row, pitch, Aj, Ax, row_lengths, x, y, 0, cfg, stream,
                                        w, (double *)workspace, &partials);
    if (st) return st;
    if (partials > 0) return cmi::reduce_partials_f64(partials, (double *)workspace, dot_dev, cmi::as_stream(stream));
    return cmi_blas_dotd_f32(num_rows, y, w, dot_dev, workspace, stream);
}
CMI_API int cmi_spmv_dia_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                                 const int32_t *diagonal_offsets, const float *values, const float *x, float *y,
                                 const float *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    if ((!w && num_rows > 0) || !dot_dev || !workspace) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia_dot: null w, result or workspace");
    int partials = 0;
    const int st = cmi::spmv_dia<float>(CMI_F32, num_rows, num_cols, num_diagonals, pitch, diagonal_offsets, values, x, y, 0, cfg, stream,
                                        w, (double *)workspace, &partials);
    if (st) return st;
    if (partials > 0) return cmi::reduce_partials_f64(partials, (double *)workspace, dot_dev, cmi::as_stream(stream));
    return cmi_blas_dotd_f32(num_rows, y, w, dot_dev, workspace, stream);
}

CMI_API int cmi_spmv_dia_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                                 const int32_t *diagonal_offsets, const double *values, const double *x, double *y,
                                 const double *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    if ((!w && num_rows > 0) || !dot_dev || !workspace) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_dia_dot: null w, result or workspace");
    int partials = 0;
    const int st = cmi::spmv_dia<double>(CMI_F64, num_rows, num_cols, num_diagonals, pitch, diagonal_offsets, values, x, y, 0, cfg, stream,
                                         w, (double *)workspace, &partials);
    if (st) return st;
    if (partials > 0) return cmi::reduce_partials_f64(partials, (double *)workspace, dot_dev, cmi::as_stream(stream));
    return cmi_blas_dot_f64(num_rows, y, w, dot_dev, workspace, stream);
}
