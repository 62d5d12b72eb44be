// spmv_coo_hyb.hip -- COO SpMV (and the HYB = ELL + COO composition) for gfx950.
//
// Replaces (reference tree): the COO flat trio cusp/system/cuda/detail/multiply/coo_flat_spmv.h:231-463 +
// coo_serial.h:38-54 (three launches and two temporary device arrays per call), the Thrust
// reduce_by_key fallback that device COO really runs on Thrust >= 1.9
// (cusp/system/detail/generic/multiply/spmv.h:185-238) and the KTT coo_spmv composite
// (cusp/system/cuda/ktt/kernels/coo_kernel.h:25-41,289-392).  HYB: generic/multiply/spmv.h:275-290.
// Arithmetic contracts: sequential/multiply/coo_spmv.h:42-68 and hyb_spmv.h:42-57.
//
// Design (wave64, no scratch allocation, one launch after an optional zero-fill of y):
//   each wave owns a contiguous interval of entries and walks it 64 entries at a time, fully
//   coalesced (lane i reads entry base+i of Ai, Aj, Ax).  Products are combined by a segmented
//   inclusive scan over equal adjacent row indices done with wave shuffles (6 steps for 64 lanes).
//   The tail lane of every segment adds its sum to y[row] with a hardware float atomic
//   (global_atomic_add_f64 / _f32), except that a segment still open at lane 63 is carried in
//   registers into the next 64 entries, so a row costs one atomic per wave interval it touches
//   (sorted input: ~1 per row).  Entries may come in any order; the result is then still correct
//   (every maximal run of equal rows is one segment) but needs more atomics.
//
//   Summation order inside a row differs from the host loop (tree inside a wave, atomics across
//   waves), so COO agrees with the oracle to rounding (<= 1e-6 relative is the contract), not bitwise.
//
// Algorithmic bytes per call (f64): 16*nnz + 16*num_rows.
#include "common.h"

namespace cmi {

template <typename T> __device__ __forceinline__ void atomic_add(T *p, T v) { unsafeAtomicAdd(p, v); }

template <typename T, bool NT>
__global__ void __launch_bounds__(1024)
coo_segmented_kernel(int64_t num_entries, const int *__restrict__ Ai, const int *__restrict__ Aj,
                     const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t interval)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    int64_t begin = wave * interval;
    int64_t end = begin + interval < num_entries ? begin + interval : num_entries;
    if (begin >= end) return; // whole wave exits together (begin/end are wave-uniform)

    int carry_row = -1; // open segment carried from the previous 64 entries (held by every lane)
    T carry_val = T(0);

    // Four 64-entry steps are LOADED together (row / column / value streams, then the x gathers) so a
    // wave keeps 16 loads in flight; the segmented scans then run step by step on registers.
    constexpr int U = 4;
    for (int64_t base4 = begin; base4 < end; base4 += U * kWave) {
        int rows_[U], cols_[U];
        T vals_[U], xs_[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t e = base4 + u * kWave + lane;
            const int64_t ec = e < end ? e : end - 1; // clamp: dead lanes re-read the last entry
            rows_[u] = ld<NT>(Ai + ec);
            cols_[u] = ld<NT>(Aj + ec);
            vals_[u] = ld<NT>(Ax + ec);
        }
#pragma unroll
        for (int u = 0; u < U; u++) xs_[u] = x[cols_[u]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t base = base4 + u * kWave;
            if (base >= end) break; // wave-uniform
            const bool live = base + lane < end;
            int row = live ? rows_[u] : -2; // -2: never equal to a real row or to the empty carry
            T val = live ? vals_[u] * xs_[u] : T(0);

            // fold the carry into lane 0 if it continues the same row, else flush it
            if (carry_row >= 0) {
                const int row0 = __shfl(row, 0);
                if (row0 == carry_row) { if (lane == 0) val = carry_val + val; }
                else if (lane == 0) atomic_add(y + carry_row, carry_val);
            }

            // segmented inclusive scan over the wave; a segment = a maximal run of equal ADJACENT rows
            // (head flags, so unsorted input -- e.g. rows 5,3,5 -- is never merged across a gap)
            const int prev_row = __shfl_up(row, 1);
            int head = (lane == 0 || prev_row != row) ? 1 : 0;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const T v = __shfl_up(val, o);
                const int h = __shfl_up(head, o);
                if (lane >= o && !head) { val = val + v; head = h; }
            }

            const int next_row = __shfl_down(row, 1);
            const bool tail = live && (lane == kWave - 1 || next_row != row);
            // the segment that reaches the last live lane stays open: carry it
            const int last_lane = (int)((end - base) < kWave ? (end - base) : kWave) - 1;
            if (tail && lane != last_lane) atomic_add(y + row, val);
            carry_row = __shfl(row, last_lane);
            carry_val = __shfl(val, last_lane);
        }
    }
    if (lane == 0 && carry_row >= 0) atomic_add(y + carry_row, carry_val);
}

// ---------------------------------------------------------------------------------------------
// coo_lane4: four consecutive entries per lane
// ---------------------------------------------------------------------------------------------
// Same contract as coo_segmented (any entry order, atomics at run ends), a quarter of the cross-lane
// work: a lane loads FOUR consecutive entries as 16-byte vectors (int4 rows, int4 columns, 2 x double2
// values: fully coalesced), reduces them to runs of equal rows sequentially in registers, and only the
// lane's LAST run enters the wave's segmented scan (one scan per 256 entries instead of four).
//   first run of a lane  : may continue the previous lane's last run -> completed with the scanned
//                          prefix of lane-1 when the lane holds more than one run;
//   middle runs (<= 2)   : complete inside the lane -> added to y directly;
//   last run             : scanned; added to y by the last lane it spans, or carried to the next step.
// Needs 16-byte aligned Ai / Aj / Ax (else the launcher uses coo_segmented).
template <typename T, bool NT>
__global__ void __launch_bounds__(1024)
coo_lane4_kernel(int64_t num_entries, const int *__restrict__ Ai, const int *__restrict__ Aj,
                 const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t interval)
{
    constexpr int K = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const int64_t begin = wave * interval; // multiple of 256
    const int64_t end = begin + interval < num_entries ? begin + interval : num_entries;
    if (begin >= end) return; // wave-uniform

    auto emit = [&](int row, T v) { if (row >= 0) atomic_add(y + row, v); };

    int carry_row = -1; // open run carried from the previous step (same in every lane)
    T carry_val = T(0);
    for (int64_t base = begin; base < end; base += K * kWave) {
        const int64_t e = base + (int64_t)lane * K;
        int r[K];
        T p[K];
        if (e + K <= end) {
            const int4v rv = ld<NT>(reinterpret_cast<const int4v *>(Ai + e));
            const int4v cv = ld<NT>(reinterpret_cast<const int4v *>(Aj + e));
            r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
            if constexpr (sizeof(T) == 8) {
                const double2v a = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                const double2v b = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
                p[0] = a.x * x[cv.x]; p[1] = a.y * x[cv.y]; p[2] = b.x * x[cv.z]; p[3] = b.y * x[cv.w];
            } else {
                const float4v a = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
                p[0] = a.x * x[cv.x]; p[1] = a.y * x[cv.y]; p[2] = a.z * x[cv.z]; p[3] = a.w * x[cv.w];
            }
        } else { // the interval's last, partial vector (or a lane past the end): entries past `end` are dead
#pragma unroll
            for (int k = 0; k < K; k++) {
                const bool live = e + k < end;
                r[k] = live ? Ai[e + k] : -2;
                p[k] = live ? Ax[e + k] * x[Aj[e + k]] : T(0);
            }
        }
        // ---- runs inside the lane ----
        const int frow = r[0];
        int lrow = r[0], nruns = 1;
        T F = T(0), L = p[0];
#pragma unroll
        for (int k = 1; k < K; k++) {
            if (r[k] == lrow) L = L + p[k];
            else {
                if (nruns == 1) F = L; else emit(lrow, L); // a middle run is complete: add it now
                nruns++;
                lrow = r[k];
                L = p[k];
            }
        }
        // ---- chain the lanes' last runs ----
        int prev_lrow = __shfl_up(lrow, 1);
        if (lane == 0) prev_lrow = carry_row;
        const bool cont = frow >= 0 && frow == prev_lrow; // my first run continues the run to my left
        const bool chain = cont && nruns == 1;            // ... and it is also my last run
        T S = L;
        if (lane == 0 && chain) S = carry_val + S;
        int head = chain ? 0 : 1;
        if (lane == 0) head = 1;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const T v = __shfl_up(S, o);
            const int h = __shfl_up(head, o);
            if (lane >= o && !head) { S = S + v; head = h; }
        }
        T S_prev = __shfl_up(S, 1);
        if (lane == 0) S_prev = carry_val;
        // the previous step's open run did not continue into this step: it is complete
        if (lane == 0 && carry_row >= 0 && !cont) emit(carry_row, carry_val);
        // last run: complete if the next lane starts another row; open at the last live lane
        const int next_frow = __shfl_down(frow, 1);
        const int64_t last_entry = (end - base) < (int64_t)K * kWave ? (end - base) : (int64_t)K * kWave;
        const int last_lane = (int)((last_entry - 1) / K);
        const bool ext = lane < last_lane && next_frow != lrow;
        // A CU retires roughly one atomic wave-instruction per 50 ns whatever its lane count
        // (MI355X_MICROARCH.md, global float atomics), so the run ends are packed into as few
        // instructions as possible: round A takes every lane's FIRST completed run (its first run if
        // that ended inside the lane, else its only run when the next lane starts a new row) -- for
        // short uniform rows that is all of them; round B the few lanes that complete two runs.
        const bool multi = nruns > 1;
        if (multi || ext) emit(multi ? frow : lrow, multi ? (cont ? S_prev + F : F) : S);
        if (multi && ext) emit(lrow, S);
        carry_row = __shfl(lrow, last_lane);
        carry_val = __shfl(S, last_lane);
    }
    if (lane == 0) emit(carry_row, carry_val);
}

template <typename T>
static int spmv_coo(int dtype, int64_t rows, int64_t cols, int64_t nnz, const int *Ai, const int *Aj, const T *Ax,
                    const T *x, T *y, int accumulate, const cmi_config *user, void *stream)
{
    if (rows < 0 || cols < 0 || nnz < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: sizes exceed the int32 index type");
    if (rows == 0) return CMI_SUCCESS;
    if (!y || (nnz > 0 && (!Ai || !Aj || !Ax || !x))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: null array");
    cmi_config c;
    select_config(CMI_FORMAT_COO, dtype, rows, cols, nnz, user, &c);
    if (c.kernel != CMI_COO_SEGMENTED && c.kernel != CMI_COO_LANE4) return fail(CMI_ERROR_NOT_SUPPORTED, "cmi_spmv_coo: config.kernel is not a COO kernel");
    hipStream_t s = as_stream(stream);
    // y = initialize(y): zero bytes are +0.0 (sequential/multiply/coo_spmv.h:56-57)
    if (!accumulate) CMI_HIP(hipMemsetAsync(y, 0, (size_t)rows * sizeof(T), s));
    if (nnz == 0) return CMI_SUCCESS;
    const int block = c.block_size;
    const int steps = c.items_per_thread < 1 ? 1 : c.items_per_thread; // steps per wave interval
    const int waves_per_block = block / kWave;
    const bool nt = (c.nontemporal & kPolLoadNT) != 0;
    const bool aligned = reinterpret_cast<uintptr_t>(Ai) % 16 == 0 && reinterpret_cast<uintptr_t>(Aj) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(Ax) % 16 == 0;
    if (c.kernel == CMI_COO_LANE4 && aligned) {
        const int64_t interval = (int64_t)steps * 4 * kWave; // 256 entries per step
        const int64_t grid64 = ceil_div(ceil_div(nnz, interval), waves_per_block);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: grid too large");
        if (nt) hipLaunchKernelGGL((coo_lane4_kernel<T, true>), dim3((int)grid64), dim3(block), 0, s, nnz, Ai, Aj, Ax, x, y, interval);
        else    hipLaunchKernelGGL((coo_lane4_kernel<T, false>), dim3((int)grid64), dim3(block), 0, s, nnz, Ai, Aj, Ax, x, y, interval);
    } else { // coo_segmented (also the unaligned-pointer path of coo_lane4)
        const int64_t interval = (int64_t)steps * kWave;
        const int64_t grid64 = ceil_div(ceil_div(nnz, interval), waves_per_block);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: grid too large");
        if (nt) hipLaunchKernelGGL((coo_segmented_kernel<T, true>), dim3((int)grid64), dim3(block), 0, s, nnz, Ai, Aj, Ax, x, y, interval);
        else    hipLaunchKernelGGL((coo_segmented_kernel<T, false>), dim3((int)grid64), dim3(block), 0, s, nnz, Ai, Aj, Ax, x, y, interval);
    }
    CMI_LAUNCH_CHECK("coo spmv");
    return CMI_SUCCESS;
}

} // namespace cmi

CMI_API int cmi_spmv_coo_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ai,
                             const int32_t *Aj, const double *Ax, const double *x, double *y, int accumulate,
                             const cmi_config *cfg, void *stream)
{
    return cmi::spmv_coo<double>(CMI_F64, num_rows, num_cols, num_entries, Ai, Aj, Ax, x, y, accumulate, cfg, stream);
}
CMI_API int cmi_spmv_coo_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ai,
                             const int32_t *Aj, const float *Ax, const float *x, float *y, int accumulate,
                             const cmi_config *cfg, void *stream)
{
    return cmi::spmv_coo<float>(CMI_F32, num_rows, num_cols, num_entries, Ai, Aj, Ax, x, y, accumulate, cfg, stream);
}

// HYB = ELL part with the caller's accumulate, then the COO part accumulating on top, same stream.
CMI_API int cmi_spmv_hyb_f64(int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row, int64_t ell_pitch,
                             const int32_t *ell_Aj, const double *ell_Ax, int64_t coo_entries,
                             const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x,
                             double *y, int accumulate, const cmi_config *cfg_ell, const cmi_config *cfg_coo,
                             void *stream)
{
    int st = cmi_spmv_ell_f64(num_rows, num_cols, ell_entries_per_row, ell_pitch, ell_Aj, ell_Ax, nullptr, x, y,
                              accumulate, cfg_ell, stream);
    if (st) return st;
    if (coo_entries == 0) return CMI_SUCCESS;
    return cmi_spmv_coo_f64(num_rows, num_cols, coo_entries, coo_Ai, coo_Aj, coo_Ax, x, y, 1, cfg_coo, stream);
}
CMI_API int cmi_spmv_hyb_f32(int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row, int64_t ell_pitch,
                             const int32_t *ell_Aj, const float *ell_Ax, int64_t coo_entries,
                             const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x,
                             float *y, int accumulate, const cmi_config *cfg_ell, const cmi_config *cfg_coo,
                             void *stream)
{
    int st = cmi_spmv_ell_f32(num_rows, num_cols, ell_entries_per_row, ell_pitch, ell_Aj, ell_Ax, nullptr, x, y,
                              accumulate, cfg_ell, stream);
    if (st) return st;
    if (coo_entries == 0) return CMI_SUCCESS;
    return cmi_spmv_coo_f32(num_rows, num_cols, coo_entries, coo_Ai, coo_Aj, coo_Ax, x, y, 1, cfg_coo, stream);
}
