// spmv_csr16.hip -- CSR SpMV with the plan's 16-bit copy of the column indices (CMI_CSR_STREAM_C16, opt-in).
//
// Beside the request shape of the streams the other way to make y = A x faster is to move fewer bytes.  Of CSR's
// 12 nnz + 20 N bytes, 4 nnz are column indices -- and inside one row tile of a banded / FEM-ordered matrix they span a
// few thousand columns, not 2^31.  A plan created with kernel = CMI_CSR_STREAM_C16 therefore keeps, on the device,
//     tile_base[t] = smallest column index among the entries of tile t          (4 bytes per tile)
//     Aj16[e]      = Aj[e] - tile_base[tile of e]   as uint16                     (2 bytes per entry)
// and the multiply reads those instead of Aj: 10 nnz + 20 N bytes (headline matrix: 700 MB instead of 800 MB).  The
// arithmetic is untouched -- same products, same storage-order sums, the bits of sequential/multiply/csr_spmv.h:56-73.
// The reference has nothing comparable (its KTT path only re-blocks the same 32-bit arrays, cuda/ktt/kernels/csr_kernel.h).
//
// All or nothing, decided by cmi_plan_create: every tile must (a) span fewer than 65536 columns and (b) fit the single
// LDS pass of csr_stream's fast path (one lane per row).  If any tile does not, the plan keeps its plain kernel and owns
// nothing.  Two tilings: csr_stream's (rows_per_block rows per tile; csr_stream16_kernel = that fast path and only that) and, for
// stencil-like rows, the wave-tile kernel's (64 rows per tile; csr_wave16_kernel).
#include "common.h"

namespace cmi {

typedef unsigned short __attribute__((ext_vector_type(4))) ushort4v;

// ---- plan time -------------------------------------------------------------------------------------------------------
// one workgroup per tile: min / max column of the tile's entries, the single-pass test; then (second launch) the encoding
__global__ void __launch_bounds__(256)
csr16_scan_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, int rpb, int tile_entries,
                  int32_t *__restrict__ tile_base, int *__restrict__ bad)
{
    __shared__ int smin[256 / kWave], smax[256 / kWave];
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int nr = (int)((num_rows - r0) < rpb ? (num_rows - r0) : rpb);
    const int nz0 = Ap[r0], nz1 = Ap[r0 + nr];
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int e = nz0 + (int)threadIdx.x; e < nz1; e += blockDim.x) {
        const int c = Aj[e];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const int l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { smin[threadIdx.x / kWave] = lo; smax[threadIdx.x / kWave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / kWave); w++) {
            lo = smin[w] < lo ? smin[w] : lo;
            hi = smax[w] > hi ? smax[w] : hi;
        }
        const bool empty = nz1 <= nz0;
        tile_base[blockIdx.x] = empty ? 0 : lo;
        if (!empty && ((int64_t)hi - lo > 65535 || lo < 0)) atomicOr(bad, 1);
        if (nz1 - (nz0 & ~3) > tile_entries) atomicOr(bad, 2); // would not fit one LDS pass
    }
}

__global__ void __launch_bounds__(256)
csr16_encode_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, int rpb,
                    const int32_t *__restrict__ tile_base, uint16_t *__restrict__ Aj16)
{
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int nr = (int)((num_rows - r0) < rpb ? (num_rows - r0) : rpb);
    const int nz0 = Ap[r0], nz1 = Ap[r0 + nr], base = tile_base[blockIdx.x];
    for (int e = nz0 + (int)threadIdx.x; e < nz1; e += blockDim.x) Aj16[e] = (uint16_t)(Aj[e] - base);
}

// Tries to give `p` (a CSR plan whose cfg is a completed CMI_CSR_STREAM shape) the 16-bit copy.  On success p->cfg.kernel
// becomes CMI_CSR_STREAM_C16 and the plan owns csr16_cols / csr16_base; otherwise nothing changes.  Synchronises `s`.
// wave_k > 0 (the plan found stencil-like rows, plan.hip wave_tiles_fit): the copy is tiled per wave of the wave-tile kernel instead --
// 64 rows per tile, wave_k entries per lane -- and p->cfg then reads block_size 256, rows_per_block 64, items_per_thread wave_k.
int csr16_build(cmi_plan *p, const int *Ap, const int *Aj, hipStream_t s, int wave_k)
{
    const cmi_config &c = p->cfg;
    const int64_t rows = p->rows, nnz = p->nnz;
    if (c.kernel != CMI_CSR_STREAM || c.threads_per_row > 1 || rows <= 0 || nnz < 4 || nnz > INT32_MAX - 65536) return CMI_SUCCESS;
    const int rpb = wave_k > 0 ? kWave : c.rows_per_block, block = wave_k > 0 ? kWave : c.block_size, ipt = c.items_per_thread;
    const int pass_entries = wave_k > 0 ? kWave * wave_k + 3 : block * ipt * 4;
    if (rpb < 1 || rpb > block) return CMI_SUCCESS; // the single-pass kernel gives every row its own lane
    const int64_t tiles = ceil_div(rows, rpb);
    if (tiles > INT32_MAX) return CMI_SUCCESS;
    int32_t *base = nullptr;
    uint16_t *cols16 = nullptr;
    int *flag = nullptr;
    int host = 1;
    hipError_t e = hipMalloc((void **)&base, (size_t)tiles * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&flag, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(csr16_scan_kernel, dim3((unsigned)tiles), dim3(wave_k > 0 ? 64 : 256), 0, s, rows, Ap, Aj, rpb, pass_entries, base, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && host == 0) {
        // (nnz rounded up to whole 8-byte vectors + one: the kernel's last vector load stays inside the allocation)
        e = hipMalloc((void **)&cols16, ((size_t)nnz + 8) * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemsetAsync(cols16, 0, ((size_t)nnz + 8) * sizeof(uint16_t), s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(csr16_encode_kernel, dim3((unsigned)tiles), dim3(wave_k > 0 ? 64 : 256), 0, s, rows, Ap, Aj, rpb, base, cols16);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (flag) (void)hipFree(flag);
    if (e != hipSuccess || host != 0) {
        if (base) (void)hipFree(base);
        if (cols16) (void)hipFree(cols16);
        return e != hipSuccess ? hip_fail(e, "cmi_plan_create: 16-bit column copy") : (int)CMI_SUCCESS;
    }
    p->csr16_base = base;
    p->csr16_cols = cols16;
    p->cfg.kernel = CMI_CSR_STREAM_C16;
    if (wave_k > 0) {
        p->csr16_wave_k = wave_k;
        p->cfg.block_size = 256;
        p->cfg.rows_per_block = kWave;
        p->cfg.items_per_thread = wave_k;
        p->cfg.threads_per_row = 0;
    }
    return CMI_SUCCESS;
}

// ---- PACKED wave tiles (CMI_CSR_STREAM_PACKED on stencil-like rows, round 4) ------------------------------------------------------
// DESIGN.md 9.5 / VERDICT r3 next 3: one stream reads at 7.1-7.3 TB/s, CSR's arrays side by side at 6.4-6.6 -- so lay everything a wave
// needs for its tile of 64 rows in ONE contiguous, line-aligned span of a plan-owned buffer, tiles at a fixed stride:
//     [ 0,  16)  int32 base (smallest column of the tile), int32 cnt (entries of the tile), 8 bytes of padding
//     [16, 144)  64 x uint16: first entry of row l inside the tile (rows past the matrix: cnt)
//     [144, 256) padding
//     [256, 256 + 128 K)          64 K x uint16 column offsets from base, entry order (K = entries per lane = the longest row)
//     [256 + 128 K, 256 + 640 K)  64 K x f64 values (f32: 256 K bytes)
// K = 5, f64: 3456 bytes = 27 lines per tile, against 3460 through the 16-bit copy's four arrays (Ap, base, Aj16, Ax).  The multiply reads
// neither Ap nor Aj nor Ax.  The plan OWNS A COPY OF THE VALUES: refresh them => new plan (cmi_plan_validate_values).
constexpr int kPackHead = 256;
inline int64_t pack16_span(int k, size_t vbytes) { return kPackHead + (int64_t)kWave * k * (2 + (int64_t)vbytes); }

template <typename T>
__global__ void __launch_bounds__(256)
csr16_pack_kernel(int64_t num_rows, int64_t wtiles, int k, const int *__restrict__ Ap, const uint16_t *__restrict__ Aj16, const int32_t *__restrict__ tile_base,
                  const T *__restrict__ Ax, unsigned char *__restrict__ packed)
{
    const int64_t wt = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (wt >= wtiles) return;
    const int64_t span_bytes = kPackHead + (int64_t)kWave * k * (2 + (int64_t)sizeof(T));
    unsigned char *span = packed + wt * span_bytes;
    const int64_t r0 = wt * kWave;
    const int nr = (int)((num_rows - r0) < kWave ? (num_rows - r0) : kWave);
    const int nz0 = Ap[r0], cnt = Ap[r0 + nr] - nz0;
    if (lane == 0) { reinterpret_cast<int32_t *>(span)[0] = tile_base[wt]; reinterpret_cast<int32_t *>(span)[1] = cnt; reinterpret_cast<int32_t *>(span)[2] = 0; reinterpret_cast<int32_t *>(span)[3] = 0; }
    reinterpret_cast<uint16_t *>(span + 16)[lane] = (uint16_t)(lane < nr ? Ap[r0 + lane] - nz0 : cnt);
    for (int i = lane; i < (kPackHead - 144) / 2; i += kWave) reinterpret_cast<uint16_t *>(span + 144)[i] = 0;
    uint16_t *cols = reinterpret_cast<uint16_t *>(span + kPackHead);
    T *vals = reinterpret_cast<T *>(span + kPackHead + (size_t)kWave * k * 2);
    for (int i = lane; i < kWave * k; i += kWave) {
        cols[i] = i < cnt ? Aj16[nz0 + i] : (uint16_t)0;
        vals[i] = i < cnt ? Ax[nz0 + i] : T(0);
    }
}

// after csr16_build(..., wave_k) granted the wave-tiled 16-bit copy: assemble the packed tiles from it and the values; the 16-bit
// arrays are released (the packed buffer holds what the multiply needs).  Synchronises `s`.
int csr16_pack(cmi_plan *p, const int *Ap, const void *values, hipStream_t s)
{
    if (!p->csr16_cols || !p->csr16_base || p->csr16_wave_k <= 0 || !values) return CMI_SUCCESS;
    const int k = p->csr16_wave_k;
    const size_t vbytes = p->dtype == CMI_F64 ? 8 : 4;
    const int64_t wtiles = ceil_div(p->rows, (int64_t)kWave);
    const int64_t bytes = wtiles * pack16_span(k, vbytes);
    unsigned char *buf = nullptr;
    hipError_t e = hipMalloc((void **)&buf, (size_t)bytes);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)ceil_div(wtiles, 4);
        if (p->dtype == CMI_F64) hipLaunchKernelGGL((csr16_pack_kernel<double>), dim3(grid), dim3(256), 0, s, p->rows, wtiles, k, Ap, p->csr16_cols, p->csr16_base, (const double *)values, buf);
        else hipLaunchKernelGGL((csr16_pack_kernel<float>), dim3(grid), dim3(256), 0, s, p->rows, wtiles, k, Ap, p->csr16_cols, p->csr16_base, (const float *)values, buf);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { if (buf) (void)hipFree(buf); return hip_fail(e, "cmi_plan_create: packed wave tiles"); }
    (void)hipFree(p->csr16_cols);
    (void)hipFree(p->csr16_base);
    p->csr16_cols = nullptr;
    p->csr16_base = nullptr;
    p->csr16_packed = buf;
    p->csr16_packed_bytes = bytes;
    p->cfg.kernel = CMI_CSR_STREAM_PACKED;
    return CMI_SUCCESS;
}

// the multiply on packed wave tiles: csr_wave16_kernel's body with every request of a wave inside its tile's span
template <typename T, int K, int POL, bool DOT>
__global__ void __launch_bounds__(1024)
csr_wave16p_kernel(int64_t num_rows, const unsigned char *__restrict__ packed, const T *__restrict__ x, T *__restrict__ y, int64_t num_tiles,
                   int64_t tiles_per_xcd, int swizzle, int accumulate, const T *__restrict__ w, double *__restrict__ dot_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    constexpr int64_t SPAN = kPackHead + (int64_t)kWave * K * (2 + (int64_t)sizeof(T));
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int waves = blockDim.x / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * waves + wave;
    const int64_t r0 = wt * kWave;
    double d = 0.0;
    if (r0 < num_rows) {
        const int nr = (int)((num_rows - r0) < kWave ? (num_rows - r0) : kWave);
        const unsigned char *span = packed + wt * SPAN;
        const int2v head = *reinterpret_cast<const int2v *>(span); // (scalar)
        const int base = head.x, cnt = head.y;
        int a = (int)ld<NT>(reinterpret_cast<const uint16_t *>(span + 16) + lane);
        const uint16_t *cols = reinterpret_cast<const uint16_t *>(span + kPackHead);
        const T *vals = reinterpret_cast<const T *>(span + kPackHead + (size_t)kWave * K * 2);
        T wv = T(0);
        if constexpr (DOT) { if (lane < nr) wv = w[r0 + lane]; }
        T s = (accumulate && lane < nr) ? y[r0 + lane] : T(0);
        if (cnt > 0) {
            T *mine = reinterpret_cast<T *>(smem) + (size_t)wave * kWave * K;
            int c[K];
            T v[K], xv[K];
            // (every slot of the span holds something -- padding slots a zero offset and a zero value -- so nothing is clamped)
#pragma unroll
            for (int k = 0; k < K; k++) c[k] = (int)ld<NT>(cols + k * kWave + lane);
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = ld<NT>(vals + k * kWave + lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[base + c[k]];
            asm volatile("" : "+v"(a));
            const int b = __builtin_amdgcn_update_dpp(cnt, a, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
#pragma unroll
            for (int k = 0; k < K; k++) mine[k * kWave + lane] = v[k] * xv[k];
            __builtin_amdgcn_wave_barrier();
            if (lane < nr)
                for (int j = a; j < b; j++) s = s + mine[j];
        }
        if (lane < nr) {
            st<NTS>(y + r0 + lane, s);
            if constexpr (DOT) d = (double)s * (double)wv;
        }
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---- the multiply, wave-tile form ------------------------------------------------------------------------------------------
// spmv_csr.hip's csr_wave_kernel reading the copy: the wave's tile is the copy's tile (64 rows), every offset is against its base.
template <typename T, int K, int POL, bool DOT>
__global__ void __launch_bounds__(1024)
csr_wave16_kernel(int64_t num_rows, const int *Ap /* not restrict: see the asm below */, const uint16_t *__restrict__ Aj16,
                  const int32_t *__restrict__ tile_base, const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y,
                  int64_t num_tiles, int64_t tiles_per_xcd, int swizzle, int accumulate, const T *__restrict__ w,
                  double *__restrict__ dot_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int waves = blockDim.x / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * waves + wave;
    const int64_t r0 = wt * kWave;
    double d = 0.0;
    if (r0 < num_rows) {
        const int nr = (int)((num_rows - r0) < kWave ? (num_rows - r0) : kWave);
        const int nz0 = Ap[r0], nz1 = Ap[r0 + nr], base = tile_base[wt];
        const int cnt = nz1 - nz0;
        int a = Ap[r0 + (lane < nr ? lane : nr)];
        T wv = T(0);
        if constexpr (DOT) { if (lane < nr) wv = w[r0 + lane]; }
        T s = (accumulate && lane < nr) ? y[r0 + lane] : T(0);
        if (cnt > 0) { // (<= 64 K by the plan's rule, checked when the copy was built)
            T *mine = reinterpret_cast<T *>(smem) + (size_t)wave * kWave * K;
            int c[K];
            T v[K], xv[K];
#pragma unroll
            for (int k = 0; k < K; k++) { const int i = k * kWave + lane; c[k] = (int)ld<NT>(Aj16 + nz0 + (i < cnt ? i : 0)); }
#pragma unroll
            for (int k = 0; k < K; k++) { const int i = k * kWave + lane; v[k] = ld<NT>(Ax + nz0 + (i < cnt ? i : 0)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[base + c[k]];
            asm volatile("" : "+v"(a));
            const int b = __builtin_amdgcn_update_dpp(nz1, a, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
#pragma unroll
            for (int k = 0; k < K; k++) mine[k * kWave + lane] = v[k] * xv[k];
            __builtin_amdgcn_wave_barrier();
            if (lane < nr)
                for (int j = a; j < b; j++) s = s + mine[j - nz0];
        }
        if (lane < nr) {
            st<NTS>(y + r0 + lane, s);
            if constexpr (DOT) d = (double)s * (double)wv;
        }
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---- the multiply ----------------------------------------------------------------------------------------------------
// csr_stream's single-pass path (spmv_csr.hip): tile bounds from two uniform loads, every lane its own row's two offsets,
// IPT index / value vectors per lane requested at once, products parked in LDS, one lane per row adds them in storage
// order.  The index vector is 8 bytes (four uint16) instead of 16; column = tile_base + offset, clamped into [0, cols)
// because the first vector of a tile may begin with up to three entries of the tile before it (encoded against THAT
// tile's base; their products are parked and never read, but the gather must stay inside x).
template <typename T, int IPT, int POL, bool DOT>
__global__ void __launch_bounds__(1024)
csr_stream16_kernel(int64_t num_rows, int64_t num_entries, int num_cols, const int *__restrict__ Ap,
                    const uint16_t *__restrict__ Aj16, const int32_t *__restrict__ tile_base, const T *__restrict__ Ax,
                    const T *__restrict__ x, T *__restrict__ y, int rows_per_block, int64_t num_tiles, int64_t tiles_per_xcd,
                    int swizzle, int accumulate, int lane_strided, const T *__restrict__ w, double *__restrict__ dot_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0;
    const int block = blockDim.x, tid = threadIdx.x;
    T *prod = reinterpret_cast<T *>(smem);
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return;
    const int64_t r0 = tile * rows_per_block;
    const int nr = (int)((num_rows - r0) < rows_per_block ? (num_rows - r0) : rows_per_block);
    const int nz0 = Ap[r0], nz1 = Ap[r0 + nr], base = tile_base[tile];
    int a = Ap[r0 + (tid < nr ? tid : nr)], b = Ap[r0 + (tid + 1 < nr ? tid + 1 : nr)];
    int origin; // the entry whose product sits in prod[0]
    if (lane_strided && nz1 > nz0) {
        // csr_stream's lane-strided request shape (spmv_csr.hip, policy bit kPolStrided): lane l takes entries l, l + block, ... from
        // the tile's first entry -- 128 B of 16-bit offsets and 512 B of f64 values per wave instruction, every line requested by
        // one instruction, nothing of the neighbouring tile read (so no clamp: every offset is against THIS tile's base);
        // branch-free, a lane past the tile's end re-reads its first entry.  Same products, same order: same bits.
        origin = nz0;
        const int cnt = nz1 - nz0;
        constexpr int K = IPT * 4;
        int c[K];
        T v[K], xv[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = k * block + tid;
            c[k] = (int)ld<NT>(Aj16 + nz0 + (i < cnt ? i : 0));
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = k * block + tid;
            v[k] = ld<NT>(Ax + nz0 + (i < cnt ? i : 0));
        }
        __builtin_amdgcn_sched_barrier(0); // every stream request is out before the first gather address is formed
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
        for (int k = 0; k < K; k++) xv[k] = x[base + c[k]];
        asm volatile("" : "+v"(a), "+v"(b)); // the row's two offsets: requested in front of the streams, not sunk behind the barrier
#pragma unroll
        for (int k = 0; k < K; k++) prod[k * block + tid] = v[k] * xv[k];
    } else {
    const int fbase = nz0 & ~3;
    origin = fbase;
    const int cmax = num_cols - 1;
    int c[IPT][4];
    T v[IPT][4];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const int e = fbase + (k * block + tid) * 4;
        if (e < nz1) {
            const ushort4v u = ld<NT>(reinterpret_cast<const ushort4v *>(Aj16 + e)); // (allocation padded: always inside)
            c[k][0] = base + (int)u.x; c[k][1] = base + (int)u.y; c[k][2] = base + (int)u.z; c[k][3] = base + (int)u.w;
            if ((int64_t)e + 4 <= num_entries) {
                if constexpr (sizeof(T) == 8) {
                    const double2v v01 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                    const double2v v23 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
                    v[k][0] = v01.x; v[k][1] = v01.y; v[k][2] = v23.x; v[k][3] = v23.y;
                } else {
                    const float4v vv = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
                    v[k][0] = vv.x; v[k][1] = vv.y; v[k][2] = vv.z; v[k][3] = vv.w;
                }
            } else { // the arrays' last, partial vector
#pragma unroll
                for (int i = 0; i < 4; i++) v[k][i] = (int64_t)e + i < num_entries ? Ax[e + i] : T(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { c[k][i] = 0; v[k][i] = T(0); }
        }
    }
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const int slot = (k * block + tid) * 4;
        if (fbase + slot < nz1) {
            T xv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int col = c[k][i];
                col = col > cmax ? cmax : col; // (base >= 0 and the offset is unsigned: never negative)
                xv[i] = x[col];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) prod[slot + i] = v[k][i] * xv[i];
        }
    }
    }
    T wv = T(0);
    if constexpr (DOT) { if (tid < nr) wv = w[r0 + tid]; }
    __syncthreads();
    double d = 0.0;
    if (tid < nr) {
        T s = accumulate ? y[r0 + tid] : T(0);
        if constexpr (IPT == 1) { for (int j = a; j < b; j++) s = s + prod[j - origin]; }
        else s = sum_in_order(s, prod + (a - origin), b - a);
        st<(POL & kPolStoreNT) != 0>(y + r0 + tid, s);
        if constexpr (DOT) d = (double)s * (double)wv;
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && tid == 0) reset_fold_state(dot_partial);
    }
}

template <typename T>
static int csr16_multiply(const cmi_plan *p, const int *Ap, const T *Ax, const T *x, T *y, int accumulate, hipStream_t s,
                          const T *w, double *dot_partial, int *dot_partials, int pol, int swizzle)
{
    const cmi_config &c = p->cfg;
    const int block = c.block_size, ipt = c.items_per_thread, rpb = c.rows_per_block;
    const int64_t rows = p->rows, nnz = p->nnz;
    if (p->csr16_packed) { // CMI_CSR_STREAM_PACKED on stencil-like rows: packed wave tiles (above); Ap / Ax are not read
        const int K = p->csr16_wave_k, wblock = 256;
        const int64_t wtiles = ceil_div(rows, (int64_t)wblock);
        const int64_t wtpx = ceil_div(wtiles, kXcds);
        const int wswz = swizzle < 0 ? 0 : swizzle;
        const int64_t wgrid = padded_grid(wtiles, wswz);
        if (wgrid > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_packed: grid too large");
        const size_t wlds = (size_t)wblock * K * sizeof(T);
        const bool wdot = w && dot_partial && wtiles <= kPartialCapacity;
        with_policy(pol & 3, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto KK) {
                constexpr int KC = decltype(KK)::value;
                if (wdot) hipLaunchKernelGGL((csr_wave16p_kernel<T, KC, POL, true>), dim3((unsigned)wgrid), dim3(wblock), wlds, s, rows, p->csr16_packed, x, y, wtiles, wtpx, wswz, accumulate, w, dot_partial);
                else      hipLaunchKernelGGL((csr_wave16p_kernel<T, KC, POL, false>), dim3((unsigned)wgrid), dim3(wblock), wlds, s, rows, p->csr16_packed, x, y, wtiles, wtpx, wswz, accumulate, (const T *)nullptr, (double *)nullptr);
            };
            switch (K) {
            case 2: go(std::integral_constant<int, 2>()); break;
            case 3: go(std::integral_constant<int, 3>()); break;
            case 4: go(std::integral_constant<int, 4>()); break;
            case 5: go(std::integral_constant<int, 5>()); break;
            case 6: go(std::integral_constant<int, 6>()); break;
            case 7: go(std::integral_constant<int, 7>()); break;
            case 8: go(std::integral_constant<int, 8>()); break;
            case 9: go(std::integral_constant<int, 9>()); break;
            default: go(std::integral_constant<int, 10>()); break;
            }
        });
        CMI_LAUNCH_CHECK("csr_wave16p spmv");
        if (wdot && dot_partials) *dot_partials = (int)wtiles;
        return CMI_SUCCESS;
    }
    if (p->csr16_wave_k > 0) { // the copy is tiled per wave: the wave-tile kernel, four waves (four tiles of the copy) per workgroup
        const int K = p->csr16_wave_k, wblock = 256;
        const int64_t wtiles = ceil_div(rows, (int64_t)wblock);
        const int64_t wtpx = ceil_div(wtiles, kXcds);
        const int wswz = swizzle < 0 ? 0 : swizzle;
        const int64_t wgrid = padded_grid(wtiles, wswz);
        if (wgrid > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_c16: grid too large");
        const size_t wlds = (size_t)wblock * K * sizeof(T);
        const bool wdot = w && dot_partial && wtiles <= kPartialCapacity;
        with_policy(pol & 3, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto KK) {
                constexpr int KC = decltype(KK)::value;
                if (wdot) hipLaunchKernelGGL((csr_wave16_kernel<T, KC, POL, true>), dim3((unsigned)wgrid), dim3(wblock), wlds, s, rows, Ap, p->csr16_cols, p->csr16_base, Ax, x, y, wtiles, wtpx, wswz, accumulate, w, dot_partial);
                else      hipLaunchKernelGGL((csr_wave16_kernel<T, KC, POL, false>), dim3((unsigned)wgrid), dim3(wblock), wlds, s, rows, Ap, p->csr16_cols, p->csr16_base, Ax, x, y, wtiles, wtpx, wswz, accumulate, (const T *)nullptr, (double *)nullptr);
            };
            switch (K) {
            case 2: go(std::integral_constant<int, 2>()); break;
            case 3: go(std::integral_constant<int, 3>()); break;
            case 4: go(std::integral_constant<int, 4>()); break;
            case 5: go(std::integral_constant<int, 5>()); break;
            case 6: go(std::integral_constant<int, 6>()); break;
            case 7: go(std::integral_constant<int, 7>()); break;
            case 8: go(std::integral_constant<int, 8>()); break;
            case 9: go(std::integral_constant<int, 9>()); break;
            default: go(std::integral_constant<int, 10>()); break;
            }
        });
        CMI_LAUNCH_CHECK("csr_wave16 spmv");
        if (wdot && dot_partials) *dot_partials = (int)wtiles;
        return CMI_SUCCESS;
    }
    const int64_t tiles = ceil_div(rows, rpb);
    const int64_t tpx = ceil_div(tiles, kXcds);
    const int swz = swizzle < 0 ? 0 : swizzle;
    const int64_t grid64 = padded_grid(tiles, swz);
    if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_c16: grid too large");
    const size_t lds = (size_t)block * ipt * 4 * sizeof(T);
    if (lds > 160 * 1024) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_c16: tile does not fit 160 KiB of LDS");
    const bool dot = w && dot_partial && tiles <= kPartialCapacity;
    const int strided = csr_lane_strided(c.nontemporal);
    int st = CMI_SUCCESS;
    with_policy(pol & 3, [&](auto P) {
        constexpr int POL = decltype(P)::value;
        auto go = [&](auto I) {
            constexpr int IPT = decltype(I)::value;
            if (dot) hipLaunchKernelGGL((csr_stream16_kernel<T, IPT, POL, true>), dim3((unsigned)grid64), dim3(block), lds, s, rows, nnz, (int)p->cols, Ap, p->csr16_cols, p->csr16_base, Ax, x, y, rpb, tiles, tpx, swz, accumulate, strided, w, dot_partial);
            else     hipLaunchKernelGGL((csr_stream16_kernel<T, IPT, POL, false>), dim3((unsigned)grid64), dim3(block), lds, s, rows, nnz, (int)p->cols, Ap, p->csr16_cols, p->csr16_base, Ax, x, y, rpb, tiles, tpx, swz, accumulate, strided, (const T *)nullptr, (double *)nullptr);
        };
        switch (ipt) {
        case 1: go(std::integral_constant<int, 1>()); break;
        case 2: go(std::integral_constant<int, 2>()); break;
        case 4: go(std::integral_constant<int, 4>()); break;
        default: st = fail(CMI_ERROR_NOT_SUPPORTED, "csr_stream_c16: items_per_thread must be 1, 2 or 4"); break;
        }
    });
    if (st) return st;
    CMI_LAUNCH_CHECK("csr_stream_c16 spmv");
    if (dot && dot_partials) *dot_partials = (int)tiles;
    return CMI_SUCCESS;
}

int csr16_multiply_f64(const cmi_plan *p, const int *Ap, const double *Ax, const double *x, double *y, int accumulate,
                       hipStream_t s, const double *w, double *dot_partial, int *dot_partials, int pol, int swizzle)
{
    return csr16_multiply<double>(p, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swizzle);
}
int csr16_multiply_f32(const cmi_plan *p, const int *Ap, const float *Ax, const float *x, float *y, int accumulate,
                       hipStream_t s, const float *w, double *dot_partial, int *dot_partials, int pol, int swizzle)
{
    return csr16_multiply<float>(p, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swizzle);
}

} // namespace cmi
