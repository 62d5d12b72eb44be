// spmv_csr_runs.hip -- CSR SpMV on the plan's RUN-COMPRESSED copy of the column indices (CMI_CSR_STREAM_WAVER, round 4; f64 and f32).
//
// FEM and KKT matrices store their columns in short consecutive runs -- three degrees of freedom per node give runs of 3, 6, 9 ...; a
// 27-point coupling is nine runs of 3 (measured on the configs[3] stand-ins, profiles/r04_column_runs.txt: ldoor-like 5.9 entries per
// run, nlpkkt120-like 2.9, thermal2-like 1.7).  For such rows the 4 bytes of column index per entry are mostly redundant, and -- what the
// round-3 ablation found to be the whole gap of this class to CSR's bare streams -- every entry costs its own 8-byte
// gather of x.  A plan made WITH the column indices therefore cuts every row into PIECES of 1..4 consecutive columns and keeps, on the
// device,
//     piece[q]  = (first column << 2) | (length - 1)                         4 bytes per piece = 1.25-1.4 bytes per entry on those matrices
//     start[t]  = {first row, first entry, first piece} of wave tile t       16 bytes per tile of ~950 entries
// and the multiply reads those instead of Aj: the index stream shrinks to a third, and a piece's x values arrive with TWO 16-byte loads
// per piece (x[c], x[c+1] and x[c+len-2], x[c+len-1]: branch-free for every length; f32: ONE load of x[c .. c+3]) instead of one gather
// per entry.
// The VALUES stay the caller's array (a caller may refresh values in place; nothing of them is cached) and are streamed exactly as
// csr_wavev streams them: (double2) pairs, every line requested once.  The x values of a piece meet the values of its entries in the
// wave's LDS region: pieces -> x values written at their entries' slots -> each lane multiplies its own value pairs in place -> the lanes
// that own a row add it in storage order.  Same products, same order of summation as sequential/multiply/csr_spmv.h:56-73: bit-exact.
// Replaces (reference): the fixed T = 32 selector of cusp/system/cuda/detail/multiply/csr_vector_spmv.h:225-258 and the __ldg gathers of
// cusp/system/cuda/ktt/kernels/csr_kernel.h:63-109 for this class of matrices.
//
// PACKED (CMI_CSR_STREAM_PACKED, opt-in, plans made with the VALUES too): per wave tile the pieces and the values are laid side by side
// in ONE plan-owned buffer -- [pieces | pad to 16 | values | pad to 16] -- so that a wave's requests are one contiguous span of HBM
// (one stream reads at 7.1-7.3 TB/s, two arrays side by side at 6.4-6.6).  The plan then owns a copy of the values:
// refresh the values => make a new plan (cmi_plan_validate checks them).
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace cmi {

constexpr int kRunCap = 4;            // entries per piece at most (2 bits of the descriptor)
constexpr int64_t kRunMaxCols = (int64_t)1 << 30;

typedef double __attribute__((ext_vector_type(2), aligned(8))) double2u; // 16 bytes of x from an 8-byte aligned address: one global_load_dwordx4
typedef float __attribute__((ext_vector_type(2), aligned(4))) float2u;   // (f32: a pair of x values / of LDS slots at any 4-byte boundary)
typedef float __attribute__((ext_vector_type(4), aligned(4))) float4u;   // f32: the four x values a piece can span, one 16-byte load
typedef int __attribute__((ext_vector_type(4))) start_t;
template <typename T> struct pair_of;
template <> struct pair_of<double> { typedef double2u type; };
template <> struct pair_of<float> { typedef float2u type; };
// element i (0..3) of a 4-vector, i known at run time only (selects, no scratch)
__device__ __forceinline__ float pick4(const float4u &a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }

// The x values of ONE piece (first column cs, len entries, 1 <= len <= 4), as what the LDS stores need: the first pair, the last pair
// (pieces of 3 and 4) and the single value (pieces of 1).  f64: two 16-byte loads, x[c], x[c+1] and x[c+len-2], x[c+len-1] (a piece of
// one entry in the LAST column loads the pair before it and takes its second element).  f32: ONE 16-byte load of x[c .. c+3] -- clamped
// to the last four columns, the piece's values picked out by the shift.  Branch-free; issues the loads and returns raw registers.
template <typename T> struct piece_x;
template <> struct piece_x<double> {
    double2u a, b;
    int sel;
    __device__ __forceinline__ void load(const double *x, int cs, int len, int num_cols)
    {
        const int ca = cs < num_cols - 2 ? cs : num_cols - 2;
        sel = cs - ca;
        a = *reinterpret_cast<const double2u *>(x + ca);
        b = *reinterpret_cast<const double2u *>(x + ca + (len > 2 ? len - 2 : 0));
    }
    __device__ __forceinline__ double2u first(int) const { return a; }
    __device__ __forceinline__ double2u last(int) const { return b; }
    __device__ __forceinline__ double single() const { return sel ? a.y : a.x; }
};
template <> struct piece_x<float> {
    float4u q;
    int sh;
    __device__ __forceinline__ void load(const float *x, int cs, int, int num_cols)
    {
        const int ca = cs < num_cols - 4 ? cs : num_cols - 4;
        sh = cs - ca;
        q = *reinterpret_cast<const float4u *>(x + ca);
    }
    __device__ __forceinline__ float2u first(int) const { return float2u{pick4(q, sh), pick4(q, sh + 1)}; }
    __device__ __forceinline__ float2u last(int len) const { return float2u{pick4(q, sh + len - 2), pick4(q, sh + len - 1)}; }
    __device__ __forceinline__ float single() const { return pick4(q, sh); }
};

// ---- plan time -------------------------------------------------------------------------------------------------------
// pieces of row r: maximal runs of consecutive columns, cut every `cap` entries.  One lane walks one row -- but over a copy of the block's
// columns in LDS: a workgroup owns `rpb` consecutive rows and first reads their entries [Ap[r0], Ap[r0 + nr]) COALESCED (a lane walking its
// row through global memory touches a line per step, 64 different lines per wave instruction: the first version of these kernels took
// 0.7-1.5 ms per pass on the configs[3] matrices, profiles/r04_plan_cost.txt; staged: one streaming read of Aj per pass).  A block whose rows
// hold more than kRunsTile entries (rows far longer than the mean) walks global memory as before: correct, slower.
constexpr int kRunsTile = 12288; // entries staged per workgroup (48 KiB of LDS)

// FILL == false: count4[r] / count3[r] <- pieces of row r cut at 4 / at 3 (one walk counts both), totals[0 / 1] += their sums.
// FILL == true: pieces[offset[r] ...] <- the row's pieces cut at `cap`.
template <bool FILL>
__global__ void __launch_bounds__(256)
runs_tile_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, int rpb, int cap, int *__restrict__ count4, int *__restrict__ count3,
                 unsigned long long *__restrict__ totals, const int *__restrict__ offset, uint32_t *__restrict__ pieces)
{
    __shared__ int cols[kRunsTile];
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    if (r0 >= num_rows) return;
    const int nr = (int)((num_rows - r0) < rpb ? (num_rows - r0) : rpb);
    const int nz0 = Ap[r0], nz1 = Ap[r0 + nr];
    const bool staged = nz1 - nz0 <= kRunsTile; // (uniform)
    if (staged) {
        for (int i = threadIdx.x; i < nz1 - nz0; i += blockDim.x) cols[i] = Aj[nz0 + i];
        __syncthreads();
    }
    int n4 = 0, n3 = 0;
    if ((int)threadIdx.x < nr) {
        const int64_t r = r0 + threadIdx.x;
        const int a = Ap[r], b = Ap[r + 1];
        int len4 = 0, len3 = 0, prev = 0, first = 0, lenc = 0, q = FILL ? offset[r] - 1 : 0;
        for (int j = a; j < b; j++) {
            const int c = staged ? cols[j - nz0] : Aj[j];
            const bool brk = j == a || c != prev + 1;
            if constexpr (FILL) {
                if (brk || lenc == cap) {
                    if (lenc > 0) pieces[q] = ((uint32_t)first << 2) | (uint32_t)(lenc - 1);
                    q++;
                    lenc = 0;
                    first = c;
                }
                lenc++;
            } else {
                if (brk || len4 == 4) { n4++; len4 = 0; }
                if (brk || len3 == 3) { n3++; len3 = 0; }
                len4++;
                len3++;
            }
            prev = c;
        }
        if constexpr (FILL) { if (lenc > 0) pieces[q] = ((uint32_t)first << 2) | (uint32_t)(lenc - 1); }
        else { count4[r] = n4; count3[r] = n3; }
    }
    if constexpr (!FILL) {
        unsigned long long t4 = (unsigned long long)n4, t3 = (unsigned long long)n3;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) { t4 += __shfl_down(t4, o); t3 += __shfl_down(t3, o); }
        if ((threadIdx.x & (kWave - 1)) == 0) { if (t4) atomicAdd(totals, t4); if (t3) atomicAdd(totals + 1, t3); }
    }
}

// tile t = the rows whose FIRST entry lies in [t q, (t + 1) q) (wave_partition_kernel's rule, spmv_csr.hip); {row, entry, piece, 0} per tile
__global__ void __launch_bounds__(256)
runs_partition_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ piece_offset, int q, int64_t tiles, int32_t *__restrict__ start)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_rows) return;
    const int64_t prev = r == 0 ? -1 : (int64_t)(Ap[r - 1] / q);
    const int e = Ap[r];
    const int64_t mine = r == num_rows ? tiles : (int64_t)(e / q);
    const int pc = piece_offset[r];
    for (int64_t t = prev + 1; t <= mine; t++) { start[4 * t] = (int32_t)r; start[4 * t + 1] = e; start[4 * t + 2] = pc; start[4 * t + 3] = 0; }
}

// PACKED: bytes / 16 of tile t's span = pieces padded to 16 bytes + values padded to 16 bytes
__global__ void __launch_bounds__(256) runs_span_kernel(int64_t tiles, const int32_t *__restrict__ start, int per16, int *__restrict__ span16)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const int np = start[4 * (t + 1) + 2] - start[4 * t + 2], cnt = start[4 * (t + 1) + 1] - start[4 * t + 1];
    span16[t] = (np + 3) / 4 + (cnt + per16 - 1) / per16; // (per16: values per 16 bytes -- 2 for f64, 4 for f32)
}
// one wave per tile copies its pieces and values into the tile's span; start[t].w <- the span's offset in 16-byte units
template <typename T>
__global__ void __launch_bounds__(256)
runs_pack_kernel(int64_t tiles, int32_t *__restrict__ start, const int *__restrict__ off16, const uint32_t *__restrict__ pieces, const T *__restrict__ Ax,
                 unsigned char *__restrict__ packed)
{
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (t > tiles) return;
    if (t == tiles) { if (lane == 0) start[4 * t + 3] = off16[t]; return; } // (the sentinel: where the buffer ends)
    const int p0 = start[4 * t + 2], np = start[4 * (t + 1) + 2] - p0, nz0 = start[4 * t + 1], cnt = start[4 * (t + 1) + 1] - nz0;
    unsigned char *span = packed + (size_t)off16[t] * 16;
    uint32_t *pc = reinterpret_cast<uint32_t *>(span);
    const int np4 = (np + 3) & ~3;
    for (int i = lane; i < np4; i += kWave) pc[i] = i < np ? pieces[p0 + i] : 0u;
    T *vals = reinterpret_cast<T *>(span + (size_t)np4 * 4);
    constexpr int per16 = 16 / (int)sizeof(T);
    const int cntp = (cnt + per16 - 1) / per16 * per16;
    for (int i = lane; i < cntp; i += kWave) vals[i] = i < cnt ? Ax[nz0 + i] : T(0);
    if (lane == 0) start[4 * t + 3] = off16[t];
}

static int device_exclusive_scan(const int *in, int *out, size_t n, hipStream_t s)
{
    size_t bytes = 0;
    void *tmp = nullptr;
    hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s);
    if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 16);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, bytes, in, out, 0, n, rocprim::plus<int>(), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (tmp) (void)hipFree(tmp);
    return e == hipSuccess ? (int)CMI_SUCCESS : hip_fail(e, "cmi_plan_create: run-compressed columns (scan)");
}

// Gives `p` (a CSR plan with a measured row profile) the run-compressed column copy on wave tiles of 256 v slots.  On success
// p->runs_start / runs_pieces are set (and runs_packed when `values` is given).  *mean_piece: entries per piece (the caller decides
// whether that pays).  Nothing is kept when min_mean_piece is not reached.  Synchronises `s`.
int csr_runs_build(cmi_plan *p, const int *Ap, const int *Aj, int v, double min_mean_piece, const void *values, hipStream_t s, double *mean_piece, int cap_asked)
{
    const int64_t rows = p->rows, nnz = p->nnz;
    if (mean_piece) *mean_piece = 0.0;
    if (rows <= 0 || nnz <= 0 || nnz > INT32_MAX - 65536 || p->cols < (p->dtype == CMI_F64 ? 2 : 4) || p->cols >= kRunMaxCols || p->prof.max_len < 1) return CMI_SUCCESS;
    const int q = 256 * v - (int)p->prof.max_len - 3;
    if (q < 1) return CMI_SUCCESS;
    int *count = nullptr, *count3 = nullptr;
    unsigned long long *totals = nullptr, host_totals[2] = {0, 0};
    hipError_t e = hipMalloc((void **)&count, (size_t)(rows + 1) * 2 * sizeof(int)); // [rows + 1] cut at 4, then [rows + 1] cut at 3
    if (e != hipSuccess) return hip_fail(e, "cmi_plan_create: run-compressed columns");
    count3 = count + rows + 1;
    e = hipMalloc((void **)&totals, sizeof(host_totals));
    if (e == hipSuccess) e = hipMemsetAsync(totals, 0, sizeof(host_totals), s);
    if (e == hipSuccess) e = hipMemsetAsync(count + rows, 0, sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(count3 + rows, 0, sizeof(int), s);
    // rows per workgroup of the tile kernels: as many as keep the block's entries inside the LDS tile at the MEAN row length (with a
    // quarter of slack), at most one per lane
    int rpb = (int)((double)kRunsTile * 0.75 / ((double)nnz / (double)rows));
    rpb = rpb > 256 ? 256 : rpb < 1 ? 1 : rpb;
    const unsigned tgrid = (unsigned)ceil_div(rows, rpb);
    // Pieces of at most 4 or at most 3 entries?  Three degrees of freedom per node give runs of 3, 6, 9 ...: cut at 4 they become 4 + 2,
    // 4 + 4 + 1 -- no fewer pieces than cut at 3, and the pieces of 4 park their x values 32 bytes apart in LDS (a 4-way bank conflict
    // where pieces of 3 have none; profiles/r04_long_rows_pmc.json: LDS index unit 71 % busy on ldoor-like).  Cap 3 when it costs at most
    // 3 % more pieces than cap 4 ($CMI_WAVER_CAP=3 / 4 forces).  One walk counts both.
    static const int cap_env_ = [] { const char *ev = std::getenv("CMI_WAVER_CAP"); return ev ? std::atoi(ev) : 0; }();
    const int cap_env = (cap_asked == 3 || cap_asked == 4) ? cap_asked : cap_env_; // (a plan that asks: config.threads_per_row)
    int cap = kRunCap;
    if (e == hipSuccess) {
        hipLaunchKernelGGL((runs_tile_kernel<false>), dim3(tgrid), dim3(256), 0, s, rows, Ap, Aj, rpb, kRunCap, count, count3, totals, (const int *)nullptr, (uint32_t *)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_totals, totals, sizeof(host_totals), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && (cap_env == 3 || (cap_env != 4 && (double)host_totals[1] <= 1.03 * (double)host_totals[0]))) cap = 3;
    if (totals) (void)hipFree(totals);
    int *chosen = cap == 3 ? count3 : count; // the per-row counts that are scanned, in place, into piece offsets
    int st = e == hipSuccess ? (int)CMI_SUCCESS : hip_fail(e, "cmi_plan_create: run-compressed columns");
    if (st == CMI_SUCCESS) st = device_exclusive_scan(chosen, chosen, (size_t)rows + 1, s); // chosen[r] <- pieces before row r
    const int total = (int)(cap == 3 ? host_totals[1] : host_totals[0]); // (= chosen[rows]: the totals were summed by the same walk)
    uint32_t *pieces = nullptr;
    int32_t *start = nullptr;
    const int64_t tiles = nnz / q + 1;
    if (st == CMI_SUCCESS && total > 0) {
        if (mean_piece) *mean_piece = (double)nnz / (double)total;
        if ((double)nnz / (double)total < min_mean_piece) { (void)hipFree(count); return CMI_SUCCESS; }
        e = hipMalloc((void **)&pieces, ((size_t)total + 64) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(pieces + total, 0, 64 * sizeof(uint32_t), s);
        if (e == hipSuccess) e = hipMalloc((void **)&start, (size_t)(tiles + 1) * 4 * sizeof(int32_t));
        if (e == hipSuccess) {
            hipLaunchKernelGGL((runs_tile_kernel<true>), dim3(tgrid), dim3(256), 0, s, rows, Ap, Aj, rpb, cap, (int *)nullptr, (int *)nullptr, (unsigned long long *)nullptr, (const int *)chosen, pieces);
            hipLaunchKernelGGL(runs_partition_kernel, dim3((unsigned)ceil_div(rows + 1, 256)), dim3(256), 0, s, rows, Ap, (const int *)chosen, q, tiles, start);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create: run-compressed columns");
    }
    (void)hipFree(count);
    unsigned char *packed = nullptr;
    int64_t packed_bytes = 0;
    if (st == CMI_SUCCESS && pieces && values) { // PACKED: spans per tile -> offsets -> the copy
        int *span = nullptr;
        e = hipMalloc((void **)&span, (size_t)(tiles + 1) * sizeof(int));
        if (e == hipSuccess) e = hipMemsetAsync(span + tiles, 0, sizeof(int), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(runs_span_kernel, dim3((unsigned)ceil_div(tiles, 256)), dim3(256), 0, s, tiles, start, p->dtype == CMI_F64 ? 2 : 4, span);
            e = hipGetLastError();
        }
        if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create: packed tiles");
        if (st == CMI_SUCCESS) st = device_exclusive_scan(span, span, (size_t)tiles + 1, s);
        int end16 = 0;
        if (st == CMI_SUCCESS) {
            e = hipMemcpy(&end16, span + tiles, sizeof(int), hipMemcpyDeviceToHost);
            if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create: packed tiles");
        }
        if (st == CMI_SUCCESS && end16 < 0) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: the packed copy would exceed 32 GiB");
        if (st == CMI_SUCCESS) {
            packed_bytes = (int64_t)end16 * 16 + 64 * 16; // (+ slack: a wave's clamped request never leaves the buffer)
            e = hipMalloc((void **)&packed, (size_t)packed_bytes);
            if (e == hipSuccess) e = hipMemsetAsync(packed + (size_t)end16 * 16, 0, 64 * 16, s);
            if (e == hipSuccess) {
                const unsigned grid = (unsigned)ceil_div(tiles + 1, 4);
                if (p->dtype == CMI_F64) hipLaunchKernelGGL((runs_pack_kernel<double>), dim3(grid), dim3(256), 0, s, tiles, start, span, pieces, (const double *)values, packed);
                else hipLaunchKernelGGL((runs_pack_kernel<float>), dim3(grid), dim3(256), 0, s, tiles, start, span, pieces, (const float *)values, packed);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create: packed tiles");
        }
        if (span) (void)hipFree(span);
    }
    if (st != CMI_SUCCESS || !pieces) {
        if (pieces) (void)hipFree(pieces);
        if (start) (void)hipFree(start);
        if (packed) (void)hipFree(packed);
        return st;
    }
    p->runs_start = start;
    p->runs_pieces = pieces;
    p->runs_count = total;
    p->runs_cap = cap;
    p->runs_packed = packed;
    p->runs_packed_bytes = packed_bytes;
    p->wave_tiles = tiles;
    p->wave_q = q;
    return CMI_SUCCESS;
}

// ---- the multiply ------------------------------------------------------------------------------------------------------
// One 64-lane wave per tile, four tiles per workgroup, no workgroup barrier (csr_wavev's structure, spmv_csr.hip).  V: 256 V request slots
// per tile.  NPC: piece chunks (of 64) handled by the unrolled, branch-free first pass -- a tile with more pieces (short runs) takes
// further turns of a plain loop.  PACKED: pieces and values come from the plan's one-span-per-tile buffer.
// (Tried and dropped, round 4 session 7: loading the VALUES per piece too and parking products -- no product stage -- was 11-13 % SLOWER,
// 83.0 / 196 us against 73.6 / 178: the value loads of a wave instruction then overlap instead of tiling the stream.  The ablations of the
// same session put the product stage at 1 % and the row sums at 4 % of this kernel: what is left is the load phase, at 6.7 TB/s of
// counted traffic.  profiles/r04_waver_ablation_and_piece_values.txt.)
template <typename T, int V, int POL, bool DOT, bool PACKED>
__global__ void __launch_bounds__(256)
csr_waver_kernel(const int32_t *__restrict__ start, int64_t wave_tiles, int64_t num_entries, int num_cols, const int *Ap /* not restrict: see csr_wave */,
                 const int *__restrict__ Aj, const uint32_t *__restrict__ pieces, const T *__restrict__ Ax, const unsigned char *__restrict__ packed,
                 const T *__restrict__ x, T *__restrict__ y, int64_t num_tiles, int64_t tiles_per_xcd, int swizzle, int accumulate,
                 const T *__restrict__ w, double *__restrict__ dot_partial, int ablate = 0)
{
    // ablate (measurements only, $CMI_WAVER_ABLATE; WRONG results by design): bit 1 -- no product stage (the rows sum the parked x values; the
    // value loads stay needed: added to lane 0's row); bit 2 -- a row's lane reads only its first slot (no sum phase)
    // f64: E = 2 values per 16-byte load, 2 V loads per lane; f32: E = 4, V loads per lane -- 256 V slots per tile either way
    constexpr int E = 16 / (int)sizeof(T), NL = (V * 4) / E, SLOTS = kWave * V * 4, NPC = V == 4 ? 6 : V == 2 ? 3 : 2;
    typedef T __attribute__((ext_vector_type(E))) val_t;
    typedef typename pair_of<T>::type pair_t;
    __shared__ __attribute__((aligned(16))) T prod[4][SLOTS];
    __shared__ double dot_slots[DOT ? 4 : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * 4 + wave;
    double d = 0.0;
    if (wt < wave_tiles) {
        const start_t lo = *reinterpret_cast<const start_t *>(start + 4 * wt), hi = *reinterpret_cast<const start_t *>(start + 4 * wt + 4);
        const int rs = lo.x, nz0 = lo.y, p0 = lo.z, re = hi.x, nz1 = hi.y, p1 = hi.z; // this tile's and the next one's {row, entry, piece}: one scalar hop
        const int nr = re - rs;
        if (nr > 0) { // (uniform per wave)
            const int np = p1 - p0;
            // where the tile's pieces and values lie, and the slot of its first entry
            const uint32_t *pc;
            const T *vbase; // value of slot i = vbase[i]
            int shift, cnt; // first entry's slot; slots that hold entries of this (or, below `shift`, the previous) tile
            bool fits;
            if constexpr (PACKED) {
                const unsigned char *span = packed + (size_t)(unsigned)lo.w * 16;
                pc = reinterpret_cast<const uint32_t *>(span);
                vbase = reinterpret_cast<const T *>(span + (size_t)((np + 3) & ~3) * 4);
                shift = 0;
                cnt = nz1 - nz0;
                fits = cnt > 0 && cnt <= SLOTS;
            } else {
                const int fbase = nz0 & ~(E - 1);
                pc = pieces + p0;
                vbase = Ax + fbase;
                shift = nz0 - fbase;
                cnt = nz1 - fbase;
                fits = nz1 > nz0 && (int64_t)((nz1 + E - 1) & ~(E - 1)) <= num_entries && cnt <= SLOTS; // (the ARRAY's last pair may reach past it)
            }
            const int first_turn_end = Ap[rs + (nr < kWave ? nr : kWave)]; // (scalar) where the 64th row of the tile ends
            int a = Ap[rs + (lane < nr ? lane : nr)], b = 0;
            T *mine = prod[wave];
            if (fits) {
                const int last = (cnt - 1) & ~(E - 1); // the last pair that holds an entry of the tile; lanes past it re-read it
                const int lastp = np - 1;
                // ---- requests: the pieces first (their chain is the longest: piece -> x -> LDS), then the value pairs ----
                uint32_t dsc[NPC];
#pragma unroll
                for (int k = 0; k < NPC; k++) {
                    const int i = k * kWave + lane;
                    dsc[k] = ld<NT>(pc + (i < lastp ? i : lastp));
                }
                val_t v[NL];
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    int e = (k * kWave + lane) * E;
                    e = e < last ? e : last;
                    v[k] = ld<NT>(reinterpret_cast<const val_t *>(vbase + e));
                }
                __builtin_amdgcn_sched_barrier(0); // every stream request is out before the first x address is formed
                // ---- x: two 16-byte loads per piece, whatever its length ----
                int run = shift; // slot of the next piece's first entry (uniform)
                int o[NPC], len[NPC];
                piece_x<T> px[NPC];
#pragma unroll
                for (int k = 0; k < NPC; k++) {
                    const bool valid = k * kWave + lane < np;
                    len[k] = valid ? (int)(dsc[k] & 3u) + 1 : 0;
                    const int incl = wave_inclusive_sum(len[k]);
                    o[k] = run + incl - len[k];
                    run += __builtin_amdgcn_readlane(incl, kWave - 1);
                    px[k].load(x, (int)(dsc[k] >> 2), len[k], num_cols);
                }
                // x values into the slots of their entries: a piece of 2+ stores its first PAIR with one LDS instruction (ds_write2_b64), a
                // piece of 3 or 4 its last pair with another (a piece of 3 rewrites its middle value with itself), a piece of 1 its one value
                // (the branch is skipped where no lane of the chunk holds one: FEM / 27-point rows)
#pragma unroll
                for (int k = 0; k < NPC; k++) {
                    if (len[k] >= 2) *reinterpret_cast<pair_t *>(mine + o[k]) = px[k].first(len[k]);
                    if (len[k] >= 3) *reinterpret_cast<pair_t *>(mine + o[k] + len[k] - 2) = px[k].last(len[k]);
                    if (len[k] == 1) mine[o[k]] = px[k].single();
                }
                for (int base = NPC * kWave; base < np; base += kWave) { // (a tile of short runs: further chunks, one at a time)
                    const int i = base + lane;
                    const uint32_t ds = pc[i < lastp ? i : lastp];
                    const int ln = i < np ? (int)(ds & 3u) + 1 : 0;
                    const int incl = wave_inclusive_sum(ln);
                    const int oo = run + incl - ln;
                    run += __builtin_amdgcn_readlane(incl, kWave - 1);
                    piece_x<T> one;
                    one.load(x, (int)(ds >> 2), ln, num_cols);
                    if (ln >= 2) *reinterpret_cast<pair_t *>(mine + oo) = one.first(ln);
                    if (ln >= 3) *reinterpret_cast<pair_t *>(mine + oo + ln - 2) = one.last(ln);
                    if (ln == 1) mine[oo] = one.single();
                }
                asm volatile("" : "+v"(a)); // the row offset was requested in front of the streams
                __builtin_amdgcn_wave_barrier(); // (compiler only: the hardware runs a wave's LDS instructions in order)
                // ---- products, in place: lane l owns the slot pairs l, l + 64, ... ----
                if (ablate & 1) { // (uniform) the values must stay NEEDED, or the compiler drops their loads
                    T keep = T(0);
#pragma unroll
                    for (int k = 0; k < NL; k++)
#pragma unroll
                        for (int i = 0; i < E; i++) keep = keep + v[k][i];
                    if (keep == T(12345.678)) mine[0] = keep;
                } else {
#pragma unroll
                    for (int k = 0; k < NL; k++) {
                        val_t *slot = reinterpret_cast<val_t *>(mine + (k * kWave + lane) * E);
                        const val_t xs = *slot;
                        val_t pr;
#pragma unroll
                        for (int i = 0; i < E; i++) pr[i] = v[k][i] * xs[i];
                        *slot = pr;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            b = __builtin_amdgcn_update_dpp(first_turn_end, a, 0x130 /* wave_shl:1: the next lane's start; lane 63 keeps the 64th row's end */, 0xf, 0xf, false);
            const int slot0 = nz0 - shift; // entry index of slot 0
            for (int r = lane; r < nr; r += kWave) { // (one turn, except over a stretch of very short rows)
                if (r >= kWave) { a = Ap[rs + r]; b = Ap[rs + r + 1]; }
                T sum = accumulate ? y[rs + r] : T(0);
                if (fits && (ablate & 2)) { if (b > a) sum = sum + mine[a - slot0]; }
                else if (fits) sum = sum_in_order(sum, mine + (a - slot0), b - a);
                else for (int j = a; j < b; j++) sum = sum + Ax[j] * x[Aj[j]]; // (the array's last pair, or an empty tile)
                st<NTS>(y + rs + r, sum);
                if constexpr (DOT) d += (double)sum * (double)w[rs + r];
            }
        }
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

template <typename T>
static int csr_runs_multiply(const cmi_plan *p, const int *Ap, const int *Aj, const T *Ax, const T *x, T *y, int accumulate, hipStream_t s,
                             const T *w, double *dot_partial, int *dot_partials, int pol, int swz_in)
{
    if (!p->runs_start || !p->runs_pieces) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_WAVER / _PACKED run through a plan of cmi_plan_create_csr only");
    const bool packed = p->cfg.kernel == CMI_CSR_STREAM_PACKED;
    if (packed && !p->runs_packed) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_PACKED: the plan holds no packed copy (cmi_plan_create_csr_values)");
    if (!packed && reinterpret_cast<uintptr_t>(Ax) % 16 != 0) return fail(CMI_ERROR_INVALID_VALUE, "csr_waver: Ax must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(x) % sizeof(T) != 0) return fail(CMI_ERROR_INVALID_VALUE, "csr_waver: x must be aligned to its element size");
    const int V = p->cfg.items_per_thread;
    if (V != 1 && V != 2 && V != 4) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_waver: items_per_thread must be 1, 2 or 4");
    const int64_t tiles = ceil_div(p->wave_tiles, (int64_t)4);
    const int64_t tpx = ceil_div(tiles, kXcds);
    const int swz = swz_in < 0 ? 0 : swz_in;
    const int64_t grid64 = padded_grid(tiles, swz);
    if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_waver: grid too large");
    const bool dot = w && dot_partial && tiles <= kPartialCapacity;
    static const int ablate = [] { const char *e = std::getenv("CMI_WAVER_ABLATE"); return e ? std::atoi(e) : 0; }();
    with_policy(pol, [&](auto P) {
        constexpr int POL = decltype(P)::value;
        auto go = [&](auto VV, auto PK) {
            constexpr int VC = decltype(VV)::value;
            constexpr bool PKC = decltype(PK)::value;
            if (dot) hipLaunchKernelGGL((csr_waver_kernel<T, VC, POL, true, PKC>), dim3((unsigned)grid64), dim3(256), 0, s, p->runs_start, p->wave_tiles, p->nnz, (int)p->cols, Ap, Aj, p->runs_pieces, Ax, p->runs_packed, x, y, tiles, tpx, swz, accumulate, w, dot_partial);
            else     hipLaunchKernelGGL((csr_waver_kernel<T, VC, POL, false, PKC>), dim3((unsigned)grid64), dim3(256), 0, s, p->runs_start, p->wave_tiles, p->nnz, (int)p->cols, Ap, Aj, p->runs_pieces, Ax, p->runs_packed, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr, ablate);
        };
        auto by_v = [&](auto PK) {
            switch (V) {
            case 1: go(std::integral_constant<int, 1>(), PK); break;
            case 2: go(std::integral_constant<int, 2>(), PK); break;
            default: go(std::integral_constant<int, 4>(), PK); break;
            }
        };
        if (packed) by_v(std::true_type()); else by_v(std::false_type());
    });
    if (dot && dot_partials) *dot_partials = (int)tiles;
    CMI_LAUNCH_CHECK("csr_waver");
    return CMI_SUCCESS;
}

int csr_runs_multiply_f64(const cmi_plan *p, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, int accumulate, hipStream_t s,
                          const double *w, double *dot_partial, int *dot_partials, int pol, int swz)
{
    return csr_runs_multiply<double>(p, Ap, Aj, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swz);
}
int csr_runs_multiply_f32(const cmi_plan *p, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, int accumulate, hipStream_t s,
                          const float *w, double *dot_partial, int *dot_partials, int pol, int swz)
{
    return csr_runs_multiply<float>(p, Ap, Aj, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swz);
}

} // namespace cmi
