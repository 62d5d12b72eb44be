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
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace cmi {

// ---------------------------------------------------------------------------------------------
// csr_scalar
// ---------------------------------------------------------------------------------------------
template <typename T, int POL>
__global__ void __launch_bounds__(1024)
csr_scalar_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj,
                  const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int accumulate)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < num_rows; row += stride) {
        const int s = Ap[row], e = Ap[row + 1];
        T acc = accumulate ? y[row] : T(0);
        constexpr bool NT = (POL & kPolLoadNT) != 0;
        for (int jj = s; jj < e; jj++) acc = acc + ld<NT>(Ax + jj) * x[ld<NT>(Aj + jj)];
        st<(POL & kPolStoreNT) != 0>(y + row, acc);
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
        // the group's lanes are folded on the DPP path (row_shr / row_bcast: VALU moves, no trip through the LDS crossbar as
        // __shfl_down = ds_bpermute makes); the sum lands in the group's last lane; every lane of a group runs the same trip count
        sum = group_sum_to_last<TPR>(sum);
        if (lane == TPR - 1) y[row] = accumulate ? y[row] + sum : sum;
    }
}

// ---------------------------------------------------------------------------------------------
// csr_stream
// ---------------------------------------------------------------------------------------------
// Tile = blockDim.x * IPT * 4 entries.  LDS: T prod[tile] then int rowptr[rows_per_block + 1].
// VEC: Aj and Ax are 16-byte aligned, so entry index e with e % 4 == 0 is a 16-byte boundary in Aj
// and a 32-byte boundary in Ax (f64) / 16-byte (f32): one int4 + two double2 (or one float4) per lane.
// DOT: the workgroup also leaves sum_r y[r] * w[r] over its rows in dot_partial[tile] (double; lanes
// folded by a fixed wave butterfly, waves in order), so <A x, w> costs no second pass over y -- the
// CG step <A p, p> (reference cusp/krylov/detail/cg.inl:80-83) with w == x == p.
template <typename T, int IPT, bool VEC, int POL, bool DOT = false, bool LONG = false>
__global__ void __launch_bounds__(1024)
csr_stream_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap,
                  const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x,
                  T *__restrict__ y, int rows_per_block, int64_t num_tiles, int64_t tiles_per_xcd,
                  int swizzle, int accumulate, int tpr, int long_len, int lane_strided, int spread_rows, int pairs, const T *__restrict__ w = nullptr,
                  double *__restrict__ dot_partial = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    __shared__ int long_next[LONG ? 1024 / kWave : 1]; // long-row path only: per-wave candidates / partial sums
    __shared__ T long_part[LONG ? 1024 / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0;
    const int block = blockDim.x;
    const int tid = threadIdx.x;
    const int tile_entries = block * IPT * 4;
    T *prod = reinterpret_cast<T *>(smem);
    int *rowptr = reinterpret_cast<int *>(smem + (size_t)tile_entries * sizeof(T));

    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup leaves together: no barrier is skipped by a part
    const int64_t r0 = tile * rows_per_block;
    const int nr = (int)((num_rows - r0) < rows_per_block ? (num_rows - r0) : rows_per_block);

    // ---- fast path (wave-uniform test): the whole tile fits ONE LDS pass of one vector per lane,
    // every row has its own lane and the tile's last vector lies inside the arrays.  This is the
    // shape the tuned short-row configurations produce; it is the general code below with every
    // loop peeled away (measured 3-4 % faster on the 5-point Poisson matrix, same arithmetic).
    // Row pointers stay in REGISTERS here: the tile bounds come from two uniform (scalar) loads, each lane
    // loads the two offsets of its own row, and nothing waits at a barrier before the index / value
    // vectors are requested -- 126.5 us instead of 129.0 on the headline matrix (archive/tools/r2_probe.hip,
    // archive/profiles/r02_probe_timing.txt: csrx flags 1 vs 0), same bits.
    //
    // Two request shapes for the entry streams of that path.  LANE-STRIDED (policy bit kPolStrided, round 2): lane l takes entries
    // l, l + block, l + 2 block, ... counted from the tile's FIRST entry -- every load instruction of a wave is one contiguous
    // span (256 B of indices, 512 B of f64 values), every line of the streams is requested by exactly one instruction, nothing
    // of the neighbouring tile is read and no alignment of the arrays is assumed.  The memory system serves that shape 7-9 %
    // faster than 16-byte vectors per lane whose two value vectors interleave (archive/tools/r2_probe.hip `shape`: 115.9 vs 124.6 us
    // for the headline matrix's bytes, archive/profiles/r02_probe_load_shape.txt), and the multiply keeps about half of that.  The body is
    // BRANCH-FREE: a lane past the tile's last entry re-reads the tile's first one and parks a product nobody reads -- with a
    // predicate per k the compiler waits for each gather before it requests the next (K round trips instead of one; the same
    // file, `csrd` before / after).  Same products, same order of summation: same bits.
    if (lane_strided && tpr == 1 && nr <= block) {
        const int nz0 = Ap[r0], nz1 = Ap[r0 + nr];
        const int cnt = nz1 - nz0;
        if (cnt > 0 && cnt <= tile_entries) {
            int a = Ap[r0 + (tid < nr ? tid : nr)], b = Ap[r0 + (tid + 1 < nr ? tid + 1 : nr)];
            constexpr int K = IPT * 4;
            int c[K];
            T v[K], xv[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int i = k * block + tid;
                const int e = nz0 + (i < cnt ? i : 0);
                c[k] = ld<NT>(Aj + e);
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int i = k * block + tid;
                v[k] = ld<NT>(Ax + nz0 + (i < cnt ? i : 0));
            }
            T wv = T(0);
            if constexpr (DOT) { if (tid < nr) wv = w[r0 + tid]; } // requested before the barrier
            // every stream request is out before the first gather address is formed (left alone, the compiler forms the
            // addresses between the value loads and waits for an index vector with half the requests still unissued) ...
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[c[k]];
            // ... and the row's two offsets were requested in front of the streams: not sunk behind the barrier
            asm volatile("" : "+v"(a), "+v"(b)); // the row's two offsets were requested in front of the streams: not sunk behind the barrier
#pragma unroll
            for (int k = 0; k < K; k++) prod[k * block + tid] = v[k] * xv[k];
            __syncthreads();
            double d = 0.0;
            if (tid < nr) {
                T s = accumulate ? y[r0 + tid] : T(0);
                if constexpr (IPT == 1) { for (int j = a; j < b; j++) s = s + prod[j - nz0]; }
                else s = sum_in_order(s, prod + (a - nz0), b - a);
                st<(POL & kPolStoreNT) != 0>(y + r0 + tid, s);
                if constexpr (DOT) d = (double)s * (double)wv;
            }
            if constexpr (DOT) {
                tile_dot_store(d, dot_slots, dot_partial + tile);
                if (tile == 0 && tid == 0) reset_fold_state(dot_partial);
            }
            return;
        }
    }
    if constexpr (VEC) {
        const int nz0 = Ap[r0], nz1 = Ap[r0 + nr];
        const int fbase = nz0 & ~3;
        if (tpr == 1 && nr <= block && nz1 - fbase <= tile_entries && (int64_t)((nz1 + 3) & ~3) <= num_entries) {
            // Which lane adds which row.  Default: lane r adds row r -- for long rows (80 rows of 45 entries in a 512-lane tile) that is
            // one and a quarter waves adding serially behind the barrier while the other waves have nothing to do, with up to 32 lanes of
            // one LDS instruction colliding on banks.  SPREAD (round 3; kernel argument, uniform): the rows are dealt round the WAVES --
            // wave w adds rows w, w + W, w + 2 W, ... on its lanes 0, 1, 2, ... -- so every SIMD works on the sum phase at once and an LDS
            // instruction carries nr / W lanes.  Same products, same order per row: same bits.
            int myrow = tid;
            if (spread_rows == 1) { const int W = block >> 6; myrow = (tid & (kWave - 1)) * W + (tid >> 6); }
            else if (spread_rows == 2) { // (measurement variant: a contiguous chunk of rows per wave -- neighbouring lanes keep neighbouring rows)
                const int W = block >> 6, chunk = (nr + W - 1) / W, l = tid & (kWave - 1);
                myrow = l < chunk ? (tid >> 6) * chunk + l : block;
            }
            const bool has_row = myrow < nr;
            const int a = Ap[r0 + (has_row ? myrow : nr)], b = Ap[r0 + (has_row ? myrow + 1 : nr)];
            // Request shape of the f64 streams.  16-BYTE VECTORS (rounds 1-2): an int4 of columns and two double2 of values per lane and
            // vector -- the two value loads of a wave interleave, every 128-byte line of Ax is touched by both instructions.  PAIRS
            // (round 3, `pairs`, f64 only): an int2 of columns and ONE double2 of values per lane and load, twice as many loads -- every
            // load instruction covers one contiguous span (512 B of indices, 1 KiB of values) and every line is requested exactly once,
            // the property of the lane-strided / wave-tile kernels; the products land in LDS 16 bytes per lane, lanes contiguous (no
            // bank conflict; the vector form's 32-byte lane stride is a 2-way one).  Same products, same slots: same bits.
            if (sizeof(T) == 8 && pairs && nz1 > nz0) { // (a tile of empty rows has no last pair to clamp to: the guarded vector body below loads nothing)
                constexpr int NP = 2 * IPT;
                int2v c2[NP];
                double2v v2[NP];
                const int lastp = (nz1 - 1) & ~1; // lanes past the tile's last pair re-read it and park products in their own (unread) slots
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    int e = fbase + (k * block + tid) * 2;
                    e = e < lastp ? e : lastp;
                    c2[k] = ld<NT>(reinterpret_cast<const int2v *>(Aj + e));
                }
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    int e = fbase + (k * block + tid) * 2;
                    e = e < lastp ? e : lastp;
                    v2[k] = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const double x0 = x[c2[k].x], x1 = x[c2[k].y];
                    *reinterpret_cast<double2v *>(prod + (k * block + tid) * 2) = double2v{v2[k].x * x0, v2[k].y * x1};
                }
            } else {
            // IPT vectors per lane, all requested before the first product is formed (round 2: the path also serves the
                // longer-row shapes of the table, IPT 2 and 4 -- FEM-like matrices of 27-80 entries per row)
                int4v c[IPT];
                T v[IPT][4];
    #pragma unroll
                for (int k = 0; k < IPT; k++) {
                    const int e = fbase + (k * block + tid) * 4;
                    if (e < nz1) {
                        c[k] = ld<NT>(reinterpret_cast<const int4v *>(Aj + e));
                        if constexpr (sizeof(T) == 8) {
                            const double2v v01 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                            const double2v v23 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
                            v[k][0] = v01.x; v[k][1] = v01.y; v[k][2] = v23.x; v[k][3] = v23.y;
                        } else {
                            const float4v vv = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
                            v[k][0] = vv.x; v[k][1] = vv.y; v[k][2] = vv.z; v[k][3] = vv.w;
                        }
                    } else {
                        c[k] = int4v{0, 0, 0, 0};
                        v[k][0] = v[k][1] = v[k][2] = v[k][3] = T(0);
                    }
                }
    #pragma unroll
                for (int k = 0; k < IPT; k++) {
                    const int slot = (k * block + tid) * 4;
                    if (fbase + slot < nz1) { // (uniform per wave except at the tile's end: the gathers of a wave stay together)
                        const T x0 = x[c[k].x], x1 = x[c[k].y], x2 = x[c[k].z], x3 = x[c[k].w];
                        prod[slot + 0] = v[k][0] * x0; prod[slot + 1] = v[k][1] * x1;
                        prod[slot + 2] = v[k][2] * x2; prod[slot + 3] = v[k][3] * x3;
                    }
                }
            }
            T wv = T(0);
            if constexpr (DOT) { if (has_row) wv = w[r0 + myrow]; } // requested before the barrier
            __syncthreads();
            double d = 0.0;
            if (has_row) {
                T s = accumulate ? y[r0 + myrow] : T(0);
                if constexpr (IPT == 1) { for (int j = a; j < b; j++) s = s + prod[j - fbase]; } // short rows: the plain loop is faster
                else s = sum_in_order(s, prod + (a - fbase), b - a);
                st<(POL & kPolStoreNT) != 0>(y + r0 + myrow, s);
                if constexpr (DOT) d = (double)s * (double)wv;
            }
            if constexpr (DOT) {
                tile_dot_store(d, dot_slots, dot_partial + tile);
                if (tile == 0 && tid == 0) reset_fold_state(dot_partial);
            }
            return;
        }
    }

    for (int i = tid; i <= nr; i += block) rowptr[i] = Ap[r0 + i];
    __syncthreads();
    const int nz0 = rowptr[0], nz1 = rowptr[nr];
    (void)nz0; (void)nz1;

    // tpr lanes share a row (tpr = 1: one lane per row, storage order, bit-exact; tpr > 1 for long
    // rows: lane-strided partial sums + a butterfly inside the tpr-lane group).  A group owns rows
    // g, g + G, ... with G = block / tpr groups (at most 4: rows_per_block <= 4*G, host-checked);
    // the group's lane 0 keeps the running sums across LDS passes.
    const int groups = block / tpr;
    const int grp = tid / tpr, sub = tid - grp * tpr;
    T acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = grp + q * groups;
        acc[q] = (accumulate && sub == 0 && r < nr) ? y[r0 + r] : T(0);
    }

    // Rows [ra, rb) of the tile = entries [rowptr[ra], rowptr[rb]), streamed through LDS in passes of
    // tile_entries.  (With VEC the first vector may start up to 3 entries early: those products belong to
    // rows before ra, are parked and never read.)
    auto run = [&](int ra, int rb) {
        const int e_lo = rowptr[ra], e_hi = rowptr[rb];
        for (int base = VEC ? (e_lo & ~3) : e_lo; base < e_hi; base += tile_entries) {
            // ---- phase 1: stream Aj/Ax, gather x, park products in LDS ----
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                if constexpr (VEC) {
                    const int slot = (k * block + tid) * 4;
                    const int e = base + slot;
                    if (e < e_hi) {
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
                        if (e < e_hi) prod[slot] = ld<NT>(Ax + e) * x[ld<NT>(Aj + e)];
                    }
                }
            }
            __syncthreads();
            // ---- phase 2: row sums out of LDS ----
            if (tpr == 1) { // one lane per row, products added in storage order
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = tid + q * block;
                    if (r >= ra && r < rb) {
                        int a = rowptr[r], b = rowptr[r + 1];
                        a = a > base ? a : base;
                        b = b < base + tile_entries ? b : base + tile_entries;
                        acc[q] = sum_in_order(acc[q], prod + (a - base), b - a);
                    }
                }
            } else { // tpr lanes per row (wave-uniform branch: tpr is a kernel argument)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = grp + q * groups;
                    if (r >= ra && r < rb) { // the same for every lane of a group
                        int a = rowptr[r], b = rowptr[r + 1];
                        a = a > base ? a : base;
                        b = b < base + tile_entries ? b : base + tile_entries;
                        T s = T(0); // (batching these strided reads as sum_strided does costs the whole kernel 6 VGPRs: not here)
                        for (int j = a + sub; j < b; j += tpr) s = s + prod[j - base];
                        for (int o = tpr >> 1; o > 0; o >>= 1) s = s + __shfl_down(s, o, tpr);
                        if (sub == 0) acc[q] = acc[q] + s;
                    }
                }
            }
            if (base + tile_entries < e_hi) __syncthreads(); // another pass reuses prod (uniform condition)
        }
    };

    // A row of long_len entries or more is not left to one lane (or one lane group): the whole workgroup
    // streams it straight from the arrays -- 16-byte vectors, partial sums in registers, nothing parked in
    // LDS -- and folds lanes, then waves, in a fixed order.  Deterministic, but re-associated: such rows are
    // in the <= 1e-6 class; every shorter row keeps the storage-order sum.  Compiled into the LONG instances
    // only (the unrolled streaming loop costs ~40 VGPRs, i.e. occupancy, which the ordinary instances keep):
    // the host launches one when the matrix's row-length profile shows such a row and threads_per_row != 1.
    auto long_row = [&](int r) {
        const int a = rowptr[r], b = rowptr[r + 1];
        T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
        if constexpr (VEC) {
            const int a4 = (a + 3) & ~3, b4 = b & ~3; // a4 <= b4: the row has at least 8 entries
            if (tid < a4 - a) s0 = ld<NT>(Ax + a + tid) * x[ld<NT>(Aj + a + tid)];
            if (tid < b - b4) s1 = ld<NT>(Ax + b4 + tid) * x[ld<NT>(Aj + b4 + tid)];
#pragma unroll 4
            for (int e = a4 + tid * 4; e < b4; e += block * 4) {
                const int4v c = ld<NT>(reinterpret_cast<const int4v *>(Aj + e));
                if constexpr (sizeof(T) == 8) {
                    const double2v v01 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
                    const double2v v23 = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
                    s0 = s0 + v01.x * x[c.x]; s1 = s1 + v01.y * x[c.y];
                    s2 = s2 + v23.x * x[c.z]; s3 = s3 + v23.y * x[c.w];
                } else {
                    const float4v v = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
                    s0 = s0 + v.x * x[c.x]; s1 = s1 + v.y * x[c.y];
                    s2 = s2 + v.z * x[c.z]; s3 = s3 + v.w * x[c.w];
                }
            }
        } else {
#pragma unroll 4
            for (int e = a + tid; e < b; e += block) s0 = s0 + ld<NT>(Ax + e) * x[ld<NT>(Aj + e)];
        }
        T s = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) s = s + __shfl_down(s, o);
        if ((tid & (kWave - 1)) == 0) long_part[tid / kWave] = s;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (sub == 0 && grp + q * groups == r) { // the lane that stores this row
                T t = T(0);
                for (int wv = 0; wv < block / kWave; wv++) t = t + long_part[wv];
                acc[q] = acc[q] + t;
            }
        }
        __syncthreads(); // long_part is free again
    };

    bool mine = false;
    if constexpr (LONG)
        for (int r = tid; r < nr; r += block) mine |= (rowptr[r + 1] - rowptr[r] >= long_len);
    if (!LONG || !__syncthreads_or(mine)) {
        run(0, nr);
    } else { // runs of ordinary rows between the long ones (every branch below is workgroup-uniform)
        int cur = 0;
        while (cur < nr) {
            int cand = nr;
            for (int r = cur + tid; r < nr; r += block)
                if (rowptr[r + 1] - rowptr[r] >= long_len) { cand = r; break; }
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) { const int v = __shfl_down(cand, o); cand = v < cand ? v : cand; }
            if ((tid & (kWave - 1)) == 0) long_next[tid / kWave] = cand;
            __syncthreads();
            int nxt = nr;
            for (int wv = 0; wv < block / kWave; wv++) nxt = long_next[wv] < nxt ? long_next[wv] : nxt;
            __syncthreads(); // long_next may be rewritten
            if (nxt > cur) {
                run(cur, nxt);
                __syncthreads(); // the next run's first pass reuses prod
            }
            if (nxt < nr) long_row(nxt);
            cur = nxt + 1;
        }
    }

    double d = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = grp + q * groups;
        if (sub == 0 && r < nr) {
            st<(POL & kPolStoreNT) != 0>(y + r0 + r, acc[q]);
            if constexpr (DOT) d += (double)acc[q] * (double)w[r0 + r];
        }
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && tid == 0) reset_fold_state(dot_partial);
    }
}

// ---------------------------------------------------------------------------------------------
// csr_wave: csr_stream's lane-strided single-pass body with WAVE-PRIVATE tiles (round 2)
// ---------------------------------------------------------------------------------------------
// For matrices whose rows all have (nearly) the same short length -- stencils: the plan selects it from the row-length profile
// (plan.hip wave_tiles_fit).  Each 64-lane wave owns rows_per_wave consecutive rows: lane l requests entries l, l + 64, ...,
// l + 64 (K - 1) of the wave's tile (K = the longest row, so a tile of 64 rows always fits 64 K slots and every lane of every
// request carries an entry), gathers x, parks the products in the wave's own LDS region and adds its row in storage order.  No
// s_barrier: a wave's LDS instructions execute in order, so nothing waits for another wave's loads; every lane owns a row in the
// sum phase (csr_stream: 192 of 256); the tile bounds are two scalar loads and the row's end is the next lane's start (one
// row-offset load per lane, wave_shl:1 on the DPP path).  Headline matrix: 124.8 -> 120-121 us (archive/tools/r2_probe.hip csrw / csrw1,
// archive/profiles/r02_probe_wave_tiles.txt).  Same products, same order of summation as the host loop: bit-exact.
// A tile that does not fit (rows longer than K: an explicit config, or a plan-less call, which knows only the mean row length) takes
// further PASSES of 64 K entries through the same body, every lane carrying its row's running sum across them (round 4; before: one lane
// per row straight from the arrays).
// LDS: T prod[waves][64 K].
template <typename T, int K, int POL, bool DOT>
__global__ void __launch_bounds__(1024)
csr_wave_kernel(int64_t num_rows, const int *Ap /* not restrict: see the asm below */, const int *__restrict__ Aj,
                const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int rows_per_wave, int64_t num_tiles,
                int64_t tiles_per_xcd, int swizzle, int accumulate, const T *__restrict__ w, double *__restrict__ dot_partial, int dot_ablate = 0)
{
    // dot_ablate (measurements only, $CMI_DOT_ABLATE; WRONG <y, w> by design): bit 1 -- w is not loaded; bit 2 -- no workgroup combine (no
    // barrier: wave 0 stores its own partial)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 1024 / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int waves = blockDim.x / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1); // (scalar: s_loads below)
    const int64_t r0 = (tile * waves + wave) * (int64_t)rows_per_wave;
    double d = 0.0;
    if (r0 < num_rows) {
        const int nr = (int)((num_rows - r0) < rows_per_wave ? (num_rows - r0) : rows_per_wave);
        const int nz0 = Ap[r0], nz1 = Ap[r0 + nr];
        const int cnt = nz1 - nz0;
        int a = Ap[r0 + (lane < nr ? lane : nr)];
        T wv = T(0);
        if constexpr (DOT) { if (lane < nr) wv = (dot_ablate & 1) ? T(1) : w[r0 + lane]; }
        if (cnt > 0 && cnt <= kWave * K) { // (uniform per wave)
            T *mine = reinterpret_cast<T *>(smem) + (size_t)wave * kWave * K;
            int c[K];
            T v[K], xv[K];
#pragma unroll
            for (int k = 0; k < K; k++) { const int i = k * kWave + lane; c[k] = ld<NT>(Aj + nz0 + (i < cnt ? i : 0)); }
#pragma unroll
            for (int k = 0; k < K; k++) { const int i = k * kWave + lane; v[k] = ld<NT>(Ax + nz0 + (i < cnt ? i : 0)); }
            __builtin_amdgcn_sched_barrier(0); // every stream request is out before the first gather address is formed
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[c[k]];
            asm volatile("" : "+v"(a)); // the row offset was requested in front of the streams
            const int b = __builtin_amdgcn_update_dpp(nz1, a, 0x130 /* wave_shl:1: the next lane's start; lane 63 keeps nz1 */, 0xf, 0xf, false);
#pragma unroll
            for (int k = 0; k < K; k++) mine[k * kWave + lane] = v[k] * xv[k];
            __builtin_amdgcn_wave_barrier(); // (compiler only: the hardware runs a wave's LDS instructions in order)
            if (lane < nr) {
                T s = accumulate ? y[r0 + lane] : T(0);
                for (int j = a; j < b; j++) s = s + mine[j - nz0];
                st<NTS>(y + r0 + lane, s);
                if constexpr (DOT) d = (double)s * (double)wv;
            }
        } else if (cnt > 0) {
            // The tile does not fit the wave's 64 K slots (rows longer than the caller's K -- a plan-less call knows only the MEAN row
            // length): the same body in PASSES of 64 K entries, every lane carrying its row's running sum across the passes and adding the
            // part of its row that lies inside the current one -- still the host loop's order, so still its bits; a row of thousands of
            // entries costs one lane a long serial sum (what csr_stream's threads_per_row = 1 costs too).  Round 4: this is what lets a
            // plan-less multiply take the wave-tile kernel without knowing the longest row.
            T *mine = reinterpret_cast<T *>(smem) + (size_t)wave * kWave * K;
            const int b = __builtin_amdgcn_update_dpp(nz1, a, 0x130, 0xf, 0xf, false);
            T s = (accumulate && lane < nr) ? y[r0 + lane] : T(0);
            for (int base = nz0; base < nz1; base += kWave * K) {
                T pr[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int i = base + k * kWave + lane;
                    const int e = i < nz1 ? i : nz0;
                    pr[k] = ld<NT>(Ax + e) * x[ld<NT>(Aj + e)];
                }
#pragma unroll
                for (int k = 0; k < K; k++) mine[k * kWave + lane] = pr[k];
                __builtin_amdgcn_wave_barrier();
                if (lane < nr) {
                    const int lo = a > base ? a : base, hi = b < base + kWave * K ? b : base + kWave * K;
                    for (int j = lo; j < hi; j++) s = s + mine[j - base];
                }
                __builtin_amdgcn_wave_barrier(); // (the next pass overwrites the slots: in order behind these reads on the hardware)
            }
            if (lane < nr) {
                st<NTS>(y + r0 + lane, s);
                if constexpr (DOT) d = (double)s * (double)wv;
            }
        } else if (lane < nr) { // no entry in these rows
            const T s = accumulate ? y[r0 + lane] : T(0);
            st<NTS>(y + r0 + lane, s);
            if constexpr (DOT) d = (double)s * (double)wv;
        }
    }
    if constexpr (DOT) {
        if (dot_ablate & 2) {
            d = wave_sum_to_last(d);
            if (threadIdx.x == kWave - 1) dot_partial[tile] = d;
        } else
            tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---------------------------------------------------------------------------------------------
// csr_wave on IRREGULAR short rows: a plan-built partition of the rows into wave tiles
// ---------------------------------------------------------------------------------------------
// Wave tile t owns the rows whose FIRST entry lies in [t Q, (t + 1) Q), Q = 64 K - longest row: its entries are at most 64 K (the
// wave's request slots, filled to Q / 64 K >= 90 %), its rows about Q / mean <= 64 (K is chosen for that, plan.hip) -- more rows
// than lanes (a stretch of very short or empty rows) take another turn of the row-sum loop.  The partition is one parallel pass
// over the row offsets (wave_partition_kernel: row r opens every tile between its predecessor's and its own), 8 bytes per tile of
// plan-owned memory.  Same body as csr_wave_kernel otherwise; same products, storage-order sums: bit-exact.
__global__ void __launch_bounds__(256)
wave_partition_kernel(int64_t num_rows, const int *__restrict__ Ap, int q, int64_t tiles, int32_t *__restrict__ start)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_rows) return;
    const int64_t prev = r == 0 ? -1 : (int64_t)(Ap[r - 1] / q);
    const int e = Ap[r];
    const int64_t mine = r == num_rows ? tiles : (int64_t)(e / q); // (the sentinel row closes every remaining tile)
    // (first row, first entry) per tile, interleaved: a wave reads its tile's bounds and its successor's with ONE 16-byte scalar load
    for (int64_t t = prev + 1; t <= mine; t++) { start[2 * t] = (int32_t)r; start[2 * t + 1] = e; }
}

int wave_partition_build(cmi_plan *p, const int *Ap, int k, hipStream_t s, int q_override)
{
    const int64_t rows = p->rows, nnz = p->nnz;
    const int q = q_override > 0 ? q_override : kWave * k - (int)p->prof.max_len;
    if (q < 1 || rows <= 0 || nnz <= 0) return CMI_SUCCESS;
    const int64_t tiles = nnz / q + 1;
    int32_t *start = nullptr;
    hipError_t e = hipMalloc((void **)&start, (size_t)(tiles + 1) * 2 * sizeof(int32_t));
    if (e != hipSuccess) return hip_fail(e, "cmi_plan_create: wave partition");
    hipLaunchKernelGGL(wave_partition_kernel, dim3((unsigned)ceil_div(rows + 1, 256)), dim3(256), 0, s, rows, Ap, q, tiles, start);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s); // (the plan is complete when cmi_plan_create returns, like every other plan)
    if (e != hipSuccess) { (void)hipFree(start); return hip_fail(e, "cmi_plan_create: wave partition"); }
    p->wave_row_start = start;
    p->wave_tiles = tiles;
    p->wave_q = q;
    return CMI_SUCCESS;
}

template <typename T, int K, int POL, bool DOT>
__global__ void __launch_bounds__(256)
csr_wavep_kernel(const int32_t *__restrict__ start, int64_t wave_tiles, const int *Ap /* not restrict: see the asm below */,
                 const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y,
                 int64_t num_tiles, int64_t tiles_per_xcd, int swizzle, int accumulate, const T *__restrict__ w,
                 double *__restrict__ dot_partial)
{
    __shared__ T prod[4][kWave * K];
    __shared__ double dot_slots[DOT ? 4 : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * 4 + wave;
    double d = 0.0;
    if (wt < wave_tiles) {
        const int2v lo = *reinterpret_cast<const int2v *>(start + 2 * wt), hi = *reinterpret_cast<const int2v *>(start + 2 * wt + 2);
        const int rs = lo.x, nz0 = lo.y, re = hi.x, nz1 = hi.y; // {first row, first entry} of this tile and of the next: one scalar hop
        const int nr = re - rs;
        if (nr > 0) { // (uniform per wave)
            const int cnt = nz1 - nz0; // <= 64 K by construction
            const int first_turn_end = Ap[rs + (nr < kWave ? nr : kWave)]; // (scalar) where the 64th row of the tile ends
            int a = Ap[rs + (lane < nr ? lane : nr)], b = 0;
            T *mine = prod[wave];
            if (cnt > 0) {
                int c[K];
                T v[K], xv[K];
#pragma unroll
                for (int k = 0; k < K; k++) { const int i = k * kWave + lane; c[k] = ld<NT>(Aj + nz0 + (i < cnt ? i : 0)); }
#pragma unroll
                for (int k = 0; k < K; k++) { const int i = k * kWave + lane; v[k] = ld<NT>(Ax + nz0 + (i < cnt ? i : 0)); }
                __builtin_amdgcn_sched_barrier(0); // every stream request is out before the first gather address is formed
#pragma unroll
                for (int k = 0; k < K; k++) asm volatile("" : "+v"(c[k]));
#pragma unroll
                for (int k = 0; k < K; k++) xv[k] = x[c[k]];
                asm volatile("" : "+v"(a)); // the row offset was requested in front of the streams
#pragma unroll
                for (int k = 0; k < K; k++) mine[k * kWave + lane] = v[k] * xv[k];
                __builtin_amdgcn_wave_barrier(); // (compiler only: the hardware runs a wave's LDS instructions in order)
            }
            b = __builtin_amdgcn_update_dpp(first_turn_end, a, 0x130 /* wave_shl:1: the next lane's start; lane 63 keeps the 64th row's end */, 0xf, 0xf, false);
            for (int r = lane; r < nr; r += kWave) { // (one turn, except over a stretch of very short rows)
                if (r >= kWave) { a = Ap[rs + r]; b = Ap[rs + r + 1]; }
                T sum = accumulate ? y[rs + r] : T(0);
                for (int j = a; j < b; j++) sum = sum + mine[j - nz0];
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

// ---------------------------------------------------------------------------------------------
// csr_wavev: wave-private tiles on a plan-built row partition with the 16-BYTE-VECTOR body (round 3) -- rows of ~16-250 entries
// ---------------------------------------------------------------------------------------------
// What the counters showed for csr_stream on FEM-like rows (ldoor-like, 45 per row; profiles/r03_long_rows_pmc_csr_stream.json): the
// bytes are right (1.07 x algorithmic) but a wave spends ~70 % of its life neither issuing nor waiting for an instruction -- it sits
// at the workgroup barrier while the slowest of eight waves finishes its three dependent round trips, and behind the barrier two of
// the eight waves own all 80 rows of the tile and add them while the tile's 32 KiB of LDS stay allocated: 19 resident waves per CU
// where 32 fit.  Here the tile belongs to ONE wave: V index vectors (int4) and their value vectors per lane, requested in one go;
// the products are parked in the wave's own LDS region (64 V x 4 slots) and the lanes that own a row add it in storage order.  No
// s_barrier anywhere: a wave starts its sums the moment its own gathers land, every wave owns rows, and the LDS goes back when
// that wave ends.  Tile t = the rows whose FIRST entry lies in [t Q, (t + 1) Q) with Q = 256 V - longest row - 3 (the 3: the tile's
// first vector starts at the 16-byte boundary at or below its first entry), so a tile always fits its wave's slots; the partition
// is wave_partition_kernel's (8 bytes per tile, plan-owned).  Same products, storage-order sums: bit-exact.
// The last vector of the ARRAYS may reach past num_entries: that one wave sums its rows straight from the arrays.
// ABL != 0: ablation instances for measurements only ($CMI_WAVEV_ABLATE, f64 / V = 4; WRONG results by design) --
//   bit 1: no x gathers (the columns are still loaded);  bit 2: no LDS, no row sums (every lane adds its own products, the wave
//   folds them and lanes < rows store the total);  bit 4: products parked in LDS as usual, but a row's lane reads only its first one.
//   bit 8 (with bit 2): the LDS ALLOCATION is kept (one store per lane), nothing else of it.
// WPB: wave tiles per workgroup (4: the shipped shape; 1 / 2: $CMI_WAVEV_WPB, V = 4 -- a workgroup's LDS goes back when its LAST wave ends).
// LDIV (ablation 10 / 11 only): the kept allocation is 1 / LDIV of the real one.
template <typename T, int V, int POL, bool DOT, int ABL = 0, int WPB = 4, int LDIV = 1>
__global__ void __launch_bounds__(kWave * WPB)
csr_wavev_kernel(const int32_t *__restrict__ start, int64_t wave_tiles, int64_t num_entries, const int *Ap /* not restrict: see csr_wave */,
                 const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t num_tiles,
                 int64_t tiles_per_xcd, int swizzle, int accumulate, const T *__restrict__ w, double *__restrict__ dot_partial)
{
    // Request shape: every load instruction of the wave covers ONE contiguous span and every 128-byte line of the streams is requested
    // by exactly one instruction (csr_stream's f64 body asks for a lane's four values with two 16-byte loads 16 bytes apart: both
    // instructions touch every line, which is why the nt hint COSTS that body 13-17 %, profiles/r03_long_rows_policy_sweep.txt).
    //   f64: E = 2 entries per load -- an int2 of columns (512 B per wave instruction) and a double2 of values (1 KiB); 2 V of each
    //   f32: E = 4 -- an int4 and a float4 (1 KiB each); V of each
    constexpr int E = sizeof(T) == 8 ? 2 : 4, NL = (V * 4) / E, SLOTS = kWave * V * 4;
    typedef int __attribute__((ext_vector_type(E))) idx_t;
    typedef T __attribute__((ext_vector_type(E))) val_t;
    __shared__ __attribute__((aligned(16))) T prod[WPB][SLOTS / LDIV];
    __shared__ double dot_slots[DOT ? WPB : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * WPB + wave;
    double d = 0.0;
    if (wt < wave_tiles) {
        const int2v lo = *reinterpret_cast<const int2v *>(start + 2 * wt), hi = *reinterpret_cast<const int2v *>(start + 2 * wt + 2);
        const int rs = lo.x, nz0 = lo.y, re = hi.x, nz1 = hi.y; // {first row, first entry} of this tile and of the next: one scalar hop
        const int nr = re - rs;
        if (nr > 0) { // (uniform per wave)
            const int fbase = nz0 & ~(E - 1);
            const int first_turn_end = Ap[rs + (nr < kWave ? nr : kWave)]; // (scalar) where the 64th row of the tile ends
            int a = Ap[rs + (lane < nr ? lane : nr)], b = 0;
            T *mine = prod[wave];
            const bool fits = nz1 > nz0 && (int64_t)((nz1 + E - 1) & ~(E - 1)) <= num_entries && nz1 - fbase <= SLOTS; // (uniform)
            if (fits) {
                const int last = (nz1 - 1) & ~(E - 1); // the last load position that holds an entry of the tile; lanes past it re-read it (and
                                                       // park products in their OWN slots, which nobody reads): no branch between the requests
                idx_t c[NL];
                val_t v[NL];
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    int e = fbase + (k * kWave + lane) * E;
                    e = e < last ? e : last;
                    c[k] = ld<NT>(reinterpret_cast<const idx_t *>(Aj + e));
                }
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    int e = fbase + (k * kWave + lane) * E;
                    e = e < last ? e : last;
                    v[k] = ld<NT>(reinterpret_cast<const val_t *>(Ax + e));
                }
                __builtin_amdgcn_sched_barrier(0); // every stream request is out before the first gather address is formed
                val_t xv[NL];
#pragma unroll
                for (int k = 0; k < NL; k++)
#pragma unroll
                    for (int i = 0; i < E; i++) {
                        if constexpr ((ABL & 1) != 0) { asm volatile("" ::"v"(c[k][i])); xv[k][i] = T(1); }
                        else xv[k][i] = x[c[k][i]];
                    }
                asm volatile("" : "+v"(a)); // the row offset was requested in front of the streams
                T lane_sum = T(0);
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    val_t pr;
#pragma unroll
                    for (int i = 0; i < E; i++) pr[i] = v[k][i] * xv[k][i];
                    if constexpr ((ABL & 2) != 0) {
#pragma unroll
                        for (int i = 0; i < E; i++) lane_sum = lane_sum + pr[i];
                    } else
                        *reinterpret_cast<val_t *>(mine + (k * kWave + lane) * E) = pr; // 16 bytes per lane, lanes contiguous: no bank conflict
                }
                __builtin_amdgcn_wave_barrier(); // (compiler only: the hardware runs a wave's LDS instructions in order)
                if constexpr ((ABL & 2) != 0) {
                    if constexpr ((ABL & 8) != 0) *reinterpret_cast<volatile T *>(mine + lane) = lane_sum;
                    // every lane's products must be NEEDED: without this the compiler masks the loads of the lanes >= nr off (session 22's
                    // first "no LDS" figures were of a kernel that read a third of the matrix)
                    lane_sum = (T)wave_sum_to_last((double)lane_sum);
                    lane_sum = (T)__shfl((double)lane_sum, kWave - 1);
                    if (lane < nr) st<NTS>(y + rs + lane, lane_sum);
                    return;
                }
            }
            b = __builtin_amdgcn_update_dpp(first_turn_end, a, 0x130 /* wave_shl:1: the next lane's start; lane 63 keeps the 64th row's end */, 0xf, 0xf, false);
            for (int r = lane; r < nr; r += kWave) { // (one turn, except over a stretch of very short rows)
                if (r >= kWave) { a = Ap[rs + r]; b = Ap[rs + r + 1]; }
                T sum = accumulate ? y[rs + r] : T(0);
                if constexpr ((ABL & 4) != 0) { if (fits && b > a) sum = sum + mine[a - fbase]; }
                else if (fits) sum = sum_in_order(sum, mine + (a - fbase), b - a);
                else for (int j = a; j < b; j++) sum = sum + Ax[j] * x[Aj[j]]; // (the array's last vector, or an empty tile)
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

// ---------------------------------------------------------------------------------------------
// csr_wavex: csr_wavev + an x WINDOW in LDS shared by the workgroup's four wave tiles (round 3) -- gather-bound band matrices
// ---------------------------------------------------------------------------------------------
// Where the columns of a row lie anywhere inside a band (FEM matrices after a bandwidth-reducing ordering, seeded 2..60 per row within
// +-2000..5000 columns: 0.31-0.44 of peak with csr_stream, 0.50-0.57 with csr_wavev), the multiply is bound by the GATHERS: every x entry
// is its own cache-line lookup in the L1 (1.04 lookups per entry, profiles/r03_long_rows_experiments.txt) at 64 lookups per wave
// instruction.  Here the workgroup first copies the x window its rows sit in the middle of -- `window` consecutive entries around the
// diagonal position of its row range, read coalesced, 16 bytes per lane -- into LDS, and a lane whose column falls inside the window
// gathers from LDS (a wave instruction of 64 random 8-byte reads costs the LDS a few cycles, not 64 tag lookups); columns outside go
// to memory as before (exec-masked: skipped when no lane needs it).  Everything else is csr_wavev: same partition, same request
// shape, same products in the same slots, storage-order sums -- bit-exact.  The window loads are issued IN FRONT of the streams, so the
// wait for them leaves the streams in flight.  LDS: window x sizeof(T) + 4 x 256 V x sizeof(T) per workgroup.
template <typename T, int V, int POL, bool DOT>
__global__ void __launch_bounds__(256)
csr_wavex_kernel(const int32_t *__restrict__ start, int64_t wave_tiles, int64_t num_entries, int64_t num_rows, int64_t num_cols, const int *Ap,
                 const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t num_tiles,
                 int64_t tiles_per_xcd, int swizzle, int accumulate, int window, const T *__restrict__ w, double *__restrict__ dot_partial)
{
    constexpr int E = sizeof(T) == 8 ? 2 : 4, NL = (V * 4) / E, SLOTS = kWave * V * 4, XE = 16 / (int)sizeof(T);
    typedef int __attribute__((ext_vector_type(E))) idx_t;
    typedef T __attribute__((ext_vector_type(E))) val_t;
    typedef T __attribute__((ext_vector_type(XE))) xvec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double dot_slots[DOT ? 4 : 1];
    T *xwin = reinterpret_cast<T *>(smem);                       // [window]
    T *prod = xwin + window;                                     // [4][SLOTS]
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= num_tiles) return; // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    const int64_t wt = tile * 4 + wave;
    // the workgroup's row range -> the window's first column (uniform: scalar loads)
    const int64_t wt_first = tile * 4, wt_last = wt_first + 4 < wave_tiles ? wt_first + 4 : wave_tiles;
    const int row_lo = start[2 * wt_first], row_hi = start[2 * wt_last];
    int64_t centre = ((int64_t)row_lo + row_hi) / 2;
    if (num_cols != num_rows) centre = (int64_t)((double)centre * (double)num_cols / (double)(num_rows > 0 ? num_rows : 1));
    int64_t w0 = centre - window / 2;
    if (w0 > num_cols - window) w0 = num_cols - window;
    if (w0 < 0) w0 = 0;
    w0 &= ~(int64_t)(XE - 1); // 16-byte aligned reads of x (cmi_malloc'ed vectors are; an unaligned x: the host does not select this kernel)
    // ---- the window: requested first ----
    const int per = window / (256 * XE); // vectors per thread (host: window is a multiple of 256 * XE)
    xvec_t xw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < per) {
            const int64_t e = w0 + ((int64_t)j * 256 + threadIdx.x) * XE;
            if (e + XE <= num_cols) xw[j] = *reinterpret_cast<const xvec_t *>(x + e);
            else {
#pragma unroll
                for (int i = 0; i < XE; i++) xw[j][i] = e + i < num_cols ? x[e + i] : T(0);
            }
        }
    }
    double d = 0.0;
    int rs = 0, nz0 = 0, nz1 = 0, nr = 0, fbase = 0, first_turn_end = 0, a = 0, b = 0;
    bool fits = false;
    idx_t c[NL];
    val_t v[NL];
    T *mine = prod + (size_t)wave * SLOTS;
    if (wt < wave_tiles) {
        const int2v lo = *reinterpret_cast<const int2v *>(start + 2 * wt), hi = *reinterpret_cast<const int2v *>(start + 2 * wt + 2);
        rs = lo.x; nz0 = lo.y; nz1 = hi.y;
        nr = hi.x - rs;
        if (nr > 0) {
            fbase = nz0 & ~(E - 1);
            first_turn_end = Ap[rs + (nr < kWave ? nr : kWave)];
            a = Ap[rs + (lane < nr ? lane : nr)];
            fits = nz1 > nz0 && (int64_t)((nz1 + E - 1) & ~(E - 1)) <= num_entries && nz1 - fbase <= SLOTS;
            if (fits) {
                const int last = (nz1 - 1) & ~(E - 1);
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    int e = fbase + (k * kWave + lane) * E;
                    e = e < last ? e : last;
                    c[k] = ld<NT>(reinterpret_cast<const idx_t *>(Aj + e));
                }
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    int e = fbase + (k * kWave + lane) * E;
                    e = e < last ? e : last;
                    v[k] = ld<NT>(reinterpret_cast<const val_t *>(Ax + e));
                }
            }
        }
    }
    // ---- window into LDS (its loads are the oldest: the wait leaves the streams in flight), then the workgroup meets ----
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (j < per) *reinterpret_cast<xvec_t *>(xwin + ((size_t)j * 256 + threadIdx.x) * XE) = xw[j];
    __syncthreads();
    if (nr > 0) {
        if (fits) {
            val_t xv[NL];
            const int iw0 = (int)w0;
#pragma unroll
            for (int k = 0; k < NL; k++)
#pragma unroll
                for (int i = 0; i < E; i++) {
                    const unsigned off = (unsigned)(c[k][i] - iw0);
                    xv[k][i] = off < (unsigned)window ? xwin[off] : x[c[k][i]];
                }
#pragma unroll
            for (int k = 0; k < NL; k++) {
                val_t pr;
#pragma unroll
                for (int i = 0; i < E; i++) pr[i] = v[k][i] * xv[k][i];
                *reinterpret_cast<val_t *>(mine + (k * kWave + lane) * E) = pr;
            }
            __builtin_amdgcn_wave_barrier();
        }
        b = __builtin_amdgcn_update_dpp(first_turn_end, a, 0x130, 0xf, 0xf, false);
        for (int r = lane; r < nr; r += kWave) {
            if (r >= kWave) { a = Ap[rs + r]; b = Ap[rs + r + 1]; }
            T sum = accumulate ? y[rs + r] : T(0);
            if (fits) sum = sum_in_order(sum, mine + (a - fbase), b - a);
            else for (int j = a; j < b; j++) sum = sum + Ax[j] * x[Aj[j]];
            st<NTS>(y + rs + r, sum);
            if constexpr (DOT) d += (double)sum * (double)w[rs + r];
        }
    }
    if constexpr (DOT) {
        tile_dot_store(d, dot_slots, dot_partial + tile);
        if (tile == 0 && threadIdx.x == 0) reset_fold_state(dot_partial);
    }
}

// ---------------------------------------------------------------------------------------------
// csr_stream_pipe: persistent, software-pipelined csr_stream
// ---------------------------------------------------------------------------------------------
// The plain csr_stream workgroup pays three DEPENDENT global round trips per tile (row pointers ->
// index/value streams -> x gather) and lives for one tile; with 8 workgroups per CU that leaves too
// few bytes in flight to saturate HBM (measured 5.5 TB/s algorithmic with HBM traffic == compulsory
// bytes, i.e. latency-bound, not traffic-bound).  Here a workgroup is persistent and walks tiles
// t, t+G, t+2G, ...; while it multiplies and sums tile i it already has in flight
//   * the 16-byte index/value vectors of tile i+1 (registers),
//   * the row pointers of tile i+1 (one per lane, register), and
//   * the two scalar tile bounds Ap[r0], Ap[r0+nr] of tile i+2 (wave-uniform -> scalar loads),
// issued in that order AFTER the x gathers of tile i so the in-order vmcnt wait for the gathers
// leaves them outstanding.  LDS (products + row pointers) is double-buffered: one barrier per tile.
// Same arithmetic as csr_stream (one lane per row, storage order): bit-identical to the host loop.
// Requires 16-byte aligned Aj/Ax and rows_per_block < blockDim.x; a tile whose entries do not fit
// the LDS tile (possible only for irregular matrices) is summed straight from global memory.
template <typename T> struct tile_regs {
    int4v c;
    T v0, v1, v2, v3;
};

// Unconditional 16-byte vector loads of entries [e, e+4): the caller clamps e to a valid, 4-aligned
// position, so there is no branch between the loads (see the vmcnt note in the kernel).
template <typename T, bool NT>
__device__ __forceinline__ void load_tile_vectors(tile_regs<T> &r, const int *__restrict__ Aj,
                                                  const T *__restrict__ Ax, int e)
{
    r.c = ld<NT>(reinterpret_cast<const int4v *>(Aj + e));
    if constexpr (sizeof(T) == 8) {
        const double2v a = ld<NT>(reinterpret_cast<const double2v *>(Ax + e));
        const double2v b = ld<NT>(reinterpret_cast<const double2v *>(Ax + e + 2));
        r.v0 = a.x; r.v1 = a.y; r.v2 = b.x; r.v3 = b.y;
    } else {
        const float4v a = ld<NT>(reinterpret_cast<const float4v *>(Ax + e));
        r.v0 = a.x; r.v1 = a.y; r.v2 = a.z; r.v3 = a.w;
    }
}

// vmcnt note.  gfx950 retires vector-memory operations in issue order and `s_waitcnt vmcnt(N)` waits
// for all but the N youngest, so "wait for the x gathers but leave the next tile's loads in flight"
// is only expressible if the number of loads issued after the gathers is the same on every path.
// Hence: no branch around any load inside the loop (addresses are clamped instead; the last
// iteration re-requests its own tile), and the tile bounds are fetched with a VECTOR load (lane 0:
// Ap[r0], other lanes: Ap[r0+nr]) and broadcast with readlane -- a scalar load would share lgkmcnt
// with the LDS traffic and force lgkmcnt(0) stalls.
template <typename T, int POL, bool ACC>
__global__ void __launch_bounds__(1024)
csr_stream_pipe_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap,
                       const int *__restrict__ Aj, const T *__restrict__ Ax, const T *__restrict__ x,
                       T *__restrict__ y, int rows_per_block, int64_t num_tiles, int chunked)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool NT = (POL & kPolLoadNT) != 0;
    const int block = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int tile_entries = block * 4;
    T *prod = reinterpret_cast<T *>(smem);                                            // [2][tile_entries]
    int *rowptr = reinterpret_cast<int *>(smem + 2 * (size_t)tile_entries * sizeof(T)); // [2][block]
    const int last_vec = (int)((num_entries - 4) & ~(int64_t)3); // host guarantees num_entries >= 4

    // tile schedule: strided (t, t+G, ...) or chunked (a contiguous run per workgroup)
    int64_t tile, tile_end, tile_step;
    if (chunked) {
        const int64_t per = (num_tiles + gridDim.x - 1) / gridDim.x;
        tile = (int64_t)blockIdx.x * per;
        tile_end = tile + per < num_tiles ? tile + per : num_tiles;
        tile_step = 1;
    } else {
        tile = blockIdx.x;
        tile_end = num_tiles;
        tile_step = gridDim.x;
    }
    if (tile >= tile_end) return;

    auto rows_of = [&](int64_t t, int64_t &r0, int &nr) {
        r0 = t * rows_per_block;
        nr = (int)((num_rows - r0) < rows_per_block ? (num_rows - r0) : rows_per_block);
    };
    auto clamp_vec = [&](int e) { return e < last_vec ? e : last_vec; };
    auto next_of = [&](int64_t t) { return t + tile_step < tile_end ? t + tile_step : t; };
    auto load_bounds = [&](int64_t t) { // lane 0 of each wave: Ap[r0]; every other lane: Ap[r0+nr]
        int64_t r0; int nr;
        rows_of(t, r0, nr);
        return Ap[r0 + (lane == 0 ? 0 : nr)];
    };

    // ---- prologue ----
    int64_t r0_c; int nr_c;
    rows_of(tile, r0_c, nr_c);
    int bnd = load_bounds(tile);
    int nz0_c = __builtin_amdgcn_readlane(bnd, 0), nz1_c = __builtin_amdgcn_readlane(bnd, 1);
    tile_regs<T> cur;
    load_tile_vectors<T, NT>(cur, Aj, Ax, clamp_vec((nz0_c & ~3) + tid * 4));
    int rp_c = Ap[r0_c + (tid < nr_c ? tid : nr_c)];
    bnd = load_bounds(next_of(tile)); // bounds of the second tile, consumed in iteration 0

    int buf = 0;
    for (;;) {
        const int base = nz0_c & ~3;
        // the tile fits one LDS pass and its last vector lies inside the arrays (wave-uniform)
        const bool fits = (nz1_c - base <= tile_entries) && ((int64_t)((nz1_c + 3) & ~3) <= num_entries);
        // every lane owns a row; lanes past the tile's last row shadow it (same value, same address)
        const int rr = tid < nr_c ? tid : nr_c - 1;
        // ---- A: gather x for the current tile (its vectors were requested one tile ago) ----
        const T x0 = x[cur.c.x], x1 = x[cur.c.y], x2 = x[cur.c.z], x3 = x[cur.c.w];
        T acc = T(0);
        if constexpr (ACC) acc = y[r0_c + rr];
        __builtin_amdgcn_sched_barrier(0);
        // ---- B: request the bounds of tile i+2, then the streams of tile i+1 ----
        const int64_t tile_n = next_of(tile);
        const bool has_next = tile_n != tile; // wave-uniform
        int64_t r0_n; int nr_n;
        rows_of(tile_n, r0_n, nr_n);
        const int nz0_n = __builtin_amdgcn_readlane(bnd, 0), nz1_n = __builtin_amdgcn_readlane(bnd, 1);
        bnd = load_bounds(next_of(tile_n));
        tile_regs<T> nxt;
        load_tile_vectors<T, NT>(nxt, Aj, Ax, clamp_vec((nz0_n & ~3) + tid * 4));
        const int rp_n = Ap[r0_n + (tid < nr_n ? tid : nr_n)];
        __builtin_amdgcn_sched_barrier(0);
        // ---- C: products and row pointers into this tile's LDS buffer (slots past the tile's
        //         entries are written too -- nobody reads them -- so nothing here is conditional) ----
        T *pbuf = prod + buf * tile_entries;
        int *rbuf = rowptr + buf * block;
        pbuf[tid * 4 + 0] = cur.v0 * x0; pbuf[tid * 4 + 1] = cur.v1 * x1;
        pbuf[tid * 4 + 2] = cur.v2 * x2; pbuf[tid * 4 + 3] = cur.v3 * x3;
        rbuf[tid] = rp_c;
        __syncthreads();
        // ---- E: one lane per row, storage order ----
        {
            const int a = rbuf[rr], b = rbuf[rr + 1];
            if (fits) {
                const T *p = pbuf + (a - base);
                const int len = b - a;
                T v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = k < len ? p[k] : T(0); // reads issue back to back
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < len) acc = acc + v[k];
                for (int k = 8; k < len; k++) acc = acc + p[k];
            } else { // oversized / array-tail tile: straight from global memory (correct, not fast)
                for (int jj = a; jj < b; jj++) acc = acc + Ax[jj] * x[Aj[jj]];
            }
            st<(POL & kPolStoreNT) != 0>(y + r0_c + rr, acc);
        }
        if (!has_next) break;
        // ---- rotate the pipeline ----
        cur = nxt; rp_c = rp_n;
        nz0_c = nz0_n; nz1_c = nz1_n;
        r0_c = r0_n; nr_c = nr_n;
        tile = tile_n;
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// csr_balanced: merge-path split of (row ends + entries) -- for irregular row lengths
// ---------------------------------------------------------------------------------------------
// The other kernels give a workgroup ROWS; one row of a million entries then serialises a workgroup
// (measured: 8 such rows in a 2M-row matrix take 37 ms instead of 50 us).  Here the unit of work is a
// tile of kBalItems (2048) ITEMS of the merged sequence "entry 0, entry 1, ..., row-0-ends, entry k, ..." --
// consuming an entry costs one product, consuming a row end one store -- so every tile does the same
// amount of work whatever the row lengths, empty rows included (Merrill & Garland's merge-based SpMV;
// the reference's KTT `csr_kernel_balanced`, cuda/ktt/kernels/csr_kernel.h:316-375, splits entries only
// and needs a row_starts array recomputed on the host side).  A workgroup owns a contiguous chunk of
// tiles: ONE cooperative 512-ary search of the row offsets finds where its chunk starts, after that the
// end of a tile is the start of the next.  Per tile: the row offsets and the products go to LDS (16-byte
// vector loads, as csr_stream), then a group of lanes per row (1..64, chosen per tile from the number of
// rows in it) sums the row's segment.  Rows that lie inside one tile are stored (or added to y when
// accumulating) without atomics; the first and last row of a tile may continue in a neighbour tile and
// go through global_atomic_add -- so y is zero-filled first when not accumulating.  Re-associates the
// row sums: parity class of csr_vector (<= 1e-6 relative), not bit-exact.
constexpr int kBalBlock = 512;
constexpr int kBalItems = 2048; // items per tile = 4 per lane

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBalBlock)
csr_balanced_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap, const int *__restrict__ Aj,
                    const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t num_tiles, int64_t per, int accumulate,
                    int64_t num_chunks, int64_t chunks_per_xcd, int swizzle)
{
    // LDS is double-buffered by tile parity: a tile's two barriers (offsets+counts, products) are then all the
    // synchronisation there is -- whoever writes buffer b for tile t+2 has passed tile t+1's barriers, which
    // every wave reaches only after its reads of tile t.
    __shared__ __attribute__((aligned(16))) T prod_buf[2][kBalItems + 8];
    __shared__ int ro_buf[2][kBalBlock + 2];
    __shared__ int wave_counts_buf[2][kBalBlock / kWave];
    const int tid = threadIdx.x;

    // `per` nominal tiles per workgroup, workgroups in launch order: items [d, d_end) of the merged sequence.
    // Only the chunk's ends sit on multiples of kBalItems (all the independent search below needs); inside the
    // chunk a tile ends after kBalItems items OR after the kBalBlock-th row end, whichever comes first, so at
    // most one row per lane ends in a tile (runs of empty or one-entry rows simply make shorter tiles).
    // workgroup -> chunk of `per` tiles: in launch order, or dealt to the XCDs in runs of `swizzle` chunks (tile_of_block) so
    // that the x window a run gathers lands in one L2
    const int64_t total_items = num_rows + num_entries;
    const int64_t chunk = tile_of_block(blockIdx.x, chunks_per_xcd, swizzle);
    if (chunk >= num_chunks) return; // whole workgroup
    int64_t d = chunk * per * kBalItems;
    int64_t d_end = d + per * kBalItems;
    if (d_end > total_items) d_end = total_items;
    if (d >= d_end) return; // whole workgroup
    (void)num_tiles;

    // ---- where does item d fall?  i0 = rows whose end item precedes it.  Row i's end is item
    //      Ap[i+1] + i of the merged sequence (strictly increasing in i): kBalBlock-ary search. ----
    int64_t lo = 0, hi = num_rows;
    while (lo < hi) {
        const int64_t step = (hi - lo + kBalBlock - 1) / kBalBlock;
        const int64_t p = lo + (int64_t)tid * step;
        const int consumed = p < hi ? ((int64_t)Ap[p + 1] + p < d) : 0;
        const int c = __syncthreads_count(consumed); // the true probes are a prefix
        if (c == 0) { hi = lo; break; }
        const int64_t last_true = lo + (int64_t)(c - 1) * step;
        const int64_t first_false = last_true + step;
        lo = last_true + 1;
        hi = first_false < hi ? first_false : hi;
    }
    int64_t i0 = lo;         // first row not yet finished
    int64_t j0 = d - i0;     // first entry not yet consumed

    // Row offsets of the next tile are requested one tile ahead (registers `pref`, `pref_up`: Ap[i0 + tid] and
    // Ap[i0 + tid + 1]), so a tile's dependent chain is  entries -> x gather -> sums,  not  offsets -> entries -> ...
    auto offset_at = [&](int64_t r) { return Ap[r < num_rows ? r : num_rows]; };
    int pref = offset_at(i0 + tid);
    int pref_up = offset_at(i0 + tid + 1); // the row's END offset (lane kBalBlock-1 holds Ap[i0 + kBalBlock])

    for (int parity = 0; d < d_end; parity ^= 1) {
        T *prod = prod_buf[parity];
        int *ro = ro_buf[parity];
        int *wave_counts = wave_counts_buf[parity];
        int64_t d1 = d + kBalItems < d_end ? d + kBalItems : d_end; // items [d, d1)
        const int avail = (int)((num_rows - i0) < kBalBlock ? (num_rows - i0) : kBalBlock); // candidate rows
        ro[tid] = pref;
        if (tid == kBalBlock - 1) ro[kBalBlock] = pref_up;
        // rows ending in the tile: end item Ap[i+1] + i < d1 -- a prefix of the candidates; counted per wave
        // from registers (ballot), summed after the one barrier that also publishes the offsets
        const unsigned long long ended = __ballot(tid < avail && (int64_t)pref_up + i0 + tid < d1);
        if ((tid & (kWave - 1)) == 0) wave_counts[tid / kWave] = __popcll(ended);
        __syncthreads();
        int c = 0;
#pragma unroll
        for (int w = 0; w < kBalBlock / kWave; w++) c += wave_counts[w];
        if (c == kBalBlock) { // every candidate ends before d1: stop right behind the last one's end item
            const int64_t stop = (int64_t)ro[kBalBlock] + i0 + kBalBlock; // its end item + 1
            if (stop < d1) d1 = stop;
        }
        // next tile's offsets: in flight while this tile's entries are processed
        pref = offset_at(i0 + c + tid);
        pref_up = offset_at(i0 + c + tid + 1);
        const int64_t j1 = d1 - (i0 + c); // the items below d1 are (i0 + c) row ends and j1 entries
        // the row after the last finished one may have its first entries here
        const bool tail_row = (i0 + c < num_rows) && (j1 > (int64_t)ro[c]);
        const int nr = c + (tail_row ? 1 : 0);

        // ---- products of entries [j0, j1) into LDS ----
        const int64_t base = VEC ? (j0 & ~(int64_t)3) : j0;
        const int span = (int)(j1 - base);
        if constexpr (VEC) {
            for (int v = tid * 4; v < span; v += kBalBlock * 4) {
                const int64_t e = base + v;
                T p0, p1, p2, p3;
                if (e + 4 <= num_entries) {
                    const int4v cidx = *reinterpret_cast<const int4v *>(Aj + e);
                    if constexpr (sizeof(T) == 8) {
                        const double2v v01 = *reinterpret_cast<const double2v *>(Ax + e);
                        const double2v v23 = *reinterpret_cast<const double2v *>(Ax + e + 2);
                        p0 = v01.x * x[cidx.x]; p1 = v01.y * x[cidx.y]; p2 = v23.x * x[cidx.z]; p3 = v23.y * x[cidx.w];
                    } else {
                        const float4v vv = *reinterpret_cast<const float4v *>(Ax + e);
                        p0 = vv.x * x[cidx.x]; p1 = vv.y * x[cidx.y]; p2 = vv.z * x[cidx.z]; p3 = vv.w * x[cidx.w];
                    }
                } else {
                    p0 = e + 0 < num_entries ? Ax[e + 0] * x[Aj[e + 0]] : T(0);
                    p1 = e + 1 < num_entries ? Ax[e + 1] * x[Aj[e + 1]] : T(0);
                    p2 = e + 2 < num_entries ? Ax[e + 2] * x[Aj[e + 2]] : T(0);
                    p3 = e + 3 < num_entries ? Ax[e + 3] * x[Aj[e + 3]] : T(0);
                }
                prod[v + 0] = p0; prod[v + 1] = p1; prod[v + 2] = p2; prod[v + 3] = p3;
            }
        } else {
            for (int v = tid; v < span; v += kBalBlock) prod[v] = Ax[base + v] * x[Aj[base + v]];
        }
        __syncthreads();

        // ---- row segments: tpr lanes per row, tpr = largest power of two with kBalBlock/tpr >= rows (<= 64) ----
        int tpr = 1;
        while (tpr < kWave && (kBalBlock / (tpr * 2)) >= nr) tpr *= 2;
        const int groups = kBalBlock / tpr;
        const int grp = tid / tpr, sub = tid - grp * tpr;
        for (int k = grp; k < nr; k += groups) { // the same k for every lane of a group
            int64_t a = ro[k], b = (k < c) ? (int64_t)ro[k + 1] : j1;
            const bool whole = (a >= j0) && (k < c); // starts and ends inside this tile
            if (a < j0) a = j0;
            if (b > j1) b = j1;
            const int first = (int)(a - base) + sub, span = (int)(b - a) - sub;
            T sum = sum_strided(T(0), prod + first, span > 0 ? (span + tpr - 1) / tpr : 0, tpr);
            for (int o = tpr >> 1; o > 0; o >>= 1) sum = sum + __shfl_down(sum, o, tpr);
            if (sub == 0) {
                T *dst = y + i0 + k;
                if (whole) *dst = accumulate ? *dst + sum : sum;
                else if (b > a) unsafeAtomicAdd(dst, sum); // continues in a neighbouring tile (y zero-filled / accumulating)
            }
        }
        i0 += c;
        j0 = j1;
        d = d1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) zero_fill_kernel(int64_t n, T *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = T(0);
}

// ---------------------------------------------------------------------------------------------
// row-length profile: which matrices get csr_balanced when the caller leaves the kernel choice open
// ---------------------------------------------------------------------------------------------
// The tuning table is keyed by the MEAN row length; skew is invisible to it, and skew is what breaks the
// row-tile kernels: csr_stream's one-lane-per-row sum (the bit-exact order) spends ~12 ns per entry of a
// long row, serially (tools/irregular_probe.py --sweep: 64 rows of 65536 entries turn 55 us into 0.8 ms).
// So the first multiply of a matrix measures its row lengths (one pass over the row offsets + a 16-byte
// read-back, ~20 us: the longest row, and how many entries sit in rows of kLongRowMin or more) and the
// result is remembered, keyed by (row-offset pointer, rows, entries, device).  It decides two things:
//   * csr_stream launches its LONG instance (long rows streamed by their whole workgroup) when there is
//     such a row and the caller did not ask for storage order everywhere (threads_per_row == 1);
//   * with the kernel choice left open, csr_balanced replaces the row-tile kernel when the longest row
//     alone would cost more than the whole multiply, or a quarter of the entries sit in long rows.
// A stale entry -- offsets edited in place, or the address reused -- can only cost speed: every kernel
// is correct for every matrix.  CMI_CSR_PROFILE=0 turns the whole thing off.
constexpr int kLongRowMin = 512, kLongRowPerLane = 128; // csr_stream's cooperative long-row path (see the kernel)

// out[0] = longest row, out[1] = entries that sit in rows of kLongRowMin entries or more, out[2] = Ap[0], out[3] = Ap[num_rows]
// (64-bit words; the two ends let plan creation refuse row offsets that do not match the caller's num_entries)
__global__ void __launch_bounds__(256) max_row_length_kernel(int64_t num_rows, const int *__restrict__ Ap, unsigned long long *__restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[2] = (unsigned long long)(long long)Ap[0]; out[3] = (unsigned long long)(long long)Ap[num_rows]; }
    __shared__ int slots[256 / kWave];
    __shared__ unsigned long long lslots[256 / kWave];
    int m = 0;
    unsigned long long in_long = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows; i += stride) {
        const int len = Ap[i + 1] - Ap[i];
        m = len > m ? len : m;
        if (len >= kLongRowMin) in_long += (unsigned long long)len;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const int v = __shfl_down(m, o);
        m = v > m ? v : m;
        in_long += __shfl_down(in_long, o);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { slots[threadIdx.x / kWave] = m; lslots[threadIdx.x / kWave] = in_long; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) { m = slots[w] > m ? slots[w] : m; in_long += lslots[w]; }
        atomicMax(out, (unsigned long long)(m > 0 ? m : 0));
        if (in_long) atomicAdd(out + 1, in_long);
    }
}

int measure_row_lengths(int64_t rows, const int *Ap, hipStream_t s, int64_t *max_len, int64_t *entries_in_long_rows, int64_t *ends)
{
    unsigned long long *dev = nullptr;
    CMI_HIP(hipMalloc((void **)&dev, 4 * sizeof(unsigned long long)));
    unsigned long long host[4] = {0, 0, 0, 0};
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(host), s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(rows, 256 * 4);
        if (blocks > kCus * 8) blocks = kCus * 8;
        hipLaunchKernelGGL(max_row_length_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, rows, Ap, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "row-length profile");
    *max_len = (int64_t)host[0];
    if (entries_in_long_rows) *entries_in_long_rows = (int64_t)host[1];
    if (ends) { ends[0] = (int64_t)host[2]; ends[1] = (int64_t)host[3]; }
    return CMI_SUCCESS;
}

// Column locality of a CSR matrix (plans made WITH the column indices, cmi_plan_create_csr): out[0] = entries whose column lies within
// `halfwin` of the diagonal position of their row (row * cols / rows) -- what an LDS x window centred on a workgroup's rows would serve --
// and out[1] = entries whose column is 16 or more away from their predecessor's in the row (no shared 128-byte line of x: every such
// entry is its own L1 lookup).  One lane per row, set-up only.
__global__ void __launch_bounds__(256) column_locality_kernel(int64_t num_rows, int64_t num_cols, const int *__restrict__ Ap, const int *__restrict__ Aj, int halfwin,
                                                              unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long slots[2][256 / kWave];
    unsigned long long inside = 0, jumps = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < num_rows; r += stride) {
        const int a = Ap[r], b = Ap[r + 1];
        const int64_t centre = num_cols == num_rows ? r : (int64_t)((double)r * (double)num_cols / (double)num_rows);
        int prev = -(1 << 30);
        for (int j = a; j < b; j++) {
            const int c = Aj[j];
            const int64_t dlt = (int64_t)c - centre;
            inside += (dlt < halfwin && dlt > -(int64_t)halfwin) ? 1u : 0u;
            if (j > a) { const int g = c - prev; jumps += (g >= 16 || g <= -16) ? 1u : 0u; }
            prev = c;
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { inside += __shfl_down(inside, o); jumps += __shfl_down(jumps, o); }
    if ((threadIdx.x & (kWave - 1)) == 0) { slots[0][threadIdx.x / kWave] = inside; slots[1][threadIdx.x / kWave] = jumps; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) { inside += slots[0][w]; jumps += slots[1][w]; }
        if (inside) atomicAdd(out, inside);
        if (jumps) atomicAdd(out + 1, jumps);
    }
}

int measure_column_locality(int64_t rows, int64_t cols, const int *Ap, const int *Aj, int halfwin, hipStream_t s, int64_t *inside, int64_t *jumps)
{
    unsigned long long *dev = nullptr, host[2] = {0, 0};
    CMI_HIP(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(host), s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(rows, 256);
        if (blocks > kCus * 16) blocks = kCus * 16;
        hipLaunchKernelGGL(column_locality_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, rows, cols, Ap, Aj, halfwin, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "column locality profile");
    *inside = (int64_t)host[0];
    *jumps = (int64_t)host[1];
    return CMI_SUCCESS;
}

// cost of the longest row inside the row-tile kernel vs the whole multiply at streaming speed.  The row's
// workgroup streams it cooperatively (kLongRowUs per entry: 1.2-2.4 ns measured, one workgroup is latency-bound at
// ~10 GB/s; tools/irregular_probe.py --sweep, archive/profiles/r01_irregular_rows.txt); with
// threads_per_row == 1 asked for, one lane adds it in storage order (kSerialRowUs per entry).
constexpr double kLongRowUs = 0.002, kSerialRowUs = 0.012;
bool prefers_balanced(int64_t rows, int64_t nnz, const row_profile &pr, size_t value_bytes, bool strict_order)
{
    const double stream_us = ((double)nnz * (4 + value_bytes) + (double)rows * (4 + 2 * value_bytes)) / 5.0e6; // 5 TB/s
    const double floor_us = 20.0; // launch + latency floor of any multiply
    if ((double)pr.max_len * (strict_order ? kSerialRowUs : kLongRowUs) > (stream_us > floor_us ? stream_us : floor_us)) return true;
    // a heavy tail: when a quarter of the entries sit in long rows, one workgroup per long row is the slower split
    // (measured: 2000 rows of 2e4 among 2M short ones, power-law lengths; tools/irregular_probe.py) -- a TAIL, though: rows that are all
    // long (400..1200 each, 50 000 of them) are what the table's lane-group shapes of the row-tile kernel are for: 95.4 / 76.4 us (f64 / f32)
    // plan-less against 131.2 / 118.5 through a plan that turned them over to the merge-path kernel (and csr_vector T = 64: 116.4 / 94.2),
    // profiles/r04_auto_regret_set2_before.txt.  So: only where the longest row is 16+ times the mean.
    const double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
    return pr.in_long * 4 > nnz && stream_us > floor_us && (double)pr.max_len > 16.0 * mean;
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(int64_t work_items, int block, int items_per_block_thread = 1)
{
    // One-shot grids: a workgroup per chunk of work, no grid-stride revisits.  Measured on MI355X
    // (tools/membench.hip): streaming kernels launched with exactly as many workgroups as there is
    // work reach 6.3-6.9 TB/s, the same kernels on a grid capped at 8 workgroups per CU and
    // grid-strided 4.5-5.0 TB/s for stores.  The kernels keep their stride loop, so the cap below
    // (2^22 workgroups) only matters for absurd sizes.
    int64_t blocks = ceil_div(work_items, (int64_t)block * items_per_block_thread);
    const int64_t cap = (int64_t)1 << 22;
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

// XCD dealing of the fused SpMV + <y, w> instance when the shape is the table's (see the call site)
static int dot_swizzle(int table_swizzle, const cmi_plan *plan, int inside_a_solve = 0)
{
    static const int env = [] { const char *e = std::getenv("CMI_DOT_SWIZZLE"); return e ? std::atoi(e) : -1; }();
    if (plan && plan->cfg_explicit) return table_swizzle; // a shape the caller gave the plan: as given
    return env >= 0 ? env : inside_a_solve;
}

template <typename T, bool VEC, int POL, bool DOT = false, bool LONG = false>
static int launch_stream_ipt(int ipt, int grid, int block, size_t lds, hipStream_t s, int64_t rows, int64_t nnz,
                             const int *Ap, const int *Aj, const T *Ax, const T *x, T *y, int rpb, int64_t tiles,
                             int64_t tpx, int swz, int acc, int tpr, int long_len, int strided, int spread, int pairs, const T *w = nullptr, double *dot_partial = nullptr)
{
#define CMI_STREAM_LAUNCH(IPT)                                                                                              \
    hipLaunchKernelGGL((csr_stream_kernel<T, IPT, VEC, POL, DOT, LONG>), dim3(grid), dim3(block), lds, s, rows, nnz, Ap, Aj, \
                       Ax, x, y, rpb, tiles, tpx, swz, acc, tpr, long_len, strided, spread, pairs, w, dot_partial)
    switch (ipt) {
    case 1: CMI_STREAM_LAUNCH(1); break;
    case 2: CMI_STREAM_LAUNCH(2); break;
    case 4: CMI_STREAM_LAUNCH(4); break;
    default: return fail(CMI_ERROR_NOT_SUPPORTED, "csr_stream: items_per_thread must be 1, 2 or 4");
    }
#undef CMI_STREAM_LAUNCH
    return CMI_SUCCESS;
}

// `plan` (may be NULL): the matrix's resolved launch shape and row-length profile (plan.hip).  Without one the table's
// row-tile kernel runs whatever the row lengths -- nothing is measured, allocated or waited for inside a multiply.
template <typename T>
static int spmv_csr(int dtype, int64_t rows, int64_t cols, int64_t nnz, const int *Ap, const int *Aj, const T *Ax,
                    const T *x, T *y, int accumulate, const cmi_config *user, void *stream, const T *w = nullptr,
                    double *dot_partial = nullptr, int *dot_partials = nullptr, const cmi_plan *plan = nullptr)
{
    // w != nullptr: the caller wants <y, w> too.  *dot_partials = number of per-tile partials the kernel
    // left in dot_partial, or 0 when the selected kernel cannot fuse it (the caller then runs a plain dot).
    if (dot_partials) *dot_partials = 0;
    if (rows < 0 || cols < 0 || nnz < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX || nnz > INT32_MAX - 65536)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: sizes exceed the int32 index type");
    if (rows == 0) return CMI_SUCCESS;
    if (!Ap || !y || (nnz > 0 && (!Aj || !Ax || !x)))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr: null array");
    cmi_config c;
    int64_t known_max_len = -1; // the matrix's longest row: known only through a plan
    if (plan) {
        c = plan->cfg; // resolved at plan creation (table / caller's config, completed; balanced if the profile says so)
        known_max_len = plan->prof.max_len;
        // A plan is made from the row offsets (and columns) alone: an AUTO plan that chose a wave-tile kernel with 16-byte loads never saw Aj, Ax
        // or x.  Arrays that are not 16-byte aligned (offset views, sliced tensors) then run the table's row-tile kernel, which takes any
        // alignment -- as they did before those kernels existed.  A kernel the caller ASKED for keeps its hard error below.
        if ((c.kernel == CMI_CSR_STREAM_WAVEV || c.kernel == CMI_CSR_STREAM_WAVEX || c.kernel == CMI_CSR_STREAM_WAVER) && !plan->kernel_asked) {
            const bool aligned = reinterpret_cast<uintptr_t>(Aj) % 16 == 0 && reinterpret_cast<uintptr_t>(Ax) % 16 == 0 &&
                                 (c.kernel != CMI_CSR_STREAM_WAVEX || reinterpret_cast<uintptr_t>(x) % 16 == 0);
            if (!aligned) select_config(CMI_FORMAT_CSR, dtype, rows, cols, nnz, nullptr, &c);
        }
    } else {
        select_config(CMI_FORMAT_CSR, dtype, rows, cols, nnz, user, &c);
        // PLAN-LESS and no kernel asked for (the literal replacement of the reference's selector, csr_vector_spmv.h:225-258: arrays in,
        // launch out, nothing measured): where the table says csr_stream and the MEAN row length sits within 0.5 % below an integer K <= 10
        // -- what a stencil of 10^5+ rows looks like from its sizes alone -- the wave-tile kernel runs with K entries per lane, the shape a
        // plan would give it (the plan-less call ran csr_stream at 0.80 where the plan's csr_wave gets 0.84).  Measured,
        // tools/planless_wave_probe.py, profiles/r04_planless_wave_rule.txt: 5- / 7- / 9-point stencils 0.96-0.99 of the table kernel's time,
        // a tridiagonal matrix 0.90.  A matrix that only LOOKS like one (random row lengths 1..9 whose mean happens to be 4.99) is still
        // correct and still streams -- tiles that overflow their 64 K slots take a second pass -- at 1.01-1.09 of the table kernel's time:
        // the price of a false positive, which the 0.5 % window makes rare.  $CMI_PLANLESS_WAVE=0: never.
        static const int planless_wave = [] { const char *e = std::getenv("CMI_PLANLESS_WAVE"); return e ? std::atoi(e) : 1; }();
        if (planless_wave && (!user || (user->kernel == CMI_KERNEL_AUTO && !user->block_size && !user->rows_per_block && !user->items_per_thread && !user->threads_per_row)) &&
            c.kernel == CMI_CSR_STREAM && c.threads_per_row <= 1 && rows >= 4096 && nnz > 0) {
            const double mean = (double)nnz / (double)rows;
            const int k = (int)std::ceil(mean);
            if (k >= 2 && k <= kWaveTileMaxK && mean >= 0.995 * k) {
                c.kernel = CMI_CSR_STREAM_WAVE;
                c.block_size = 256;
                c.rows_per_block = 256;
                c.items_per_thread = k;
                c.threads_per_row = 0;
                c.nontemporal &= ~kPolStrided;
                if (!(user && user->nontemporal)) {
                    if (nnz * (int64_t)(sizeof(int) + sizeof(T)) > kInfinityCacheBytes + kInfinityCacheBytes / 4) c.nontemporal |= kPolLoadNT;
                    c.nontemporal |= kPolStoreNT;
                }
            }
        }
    }
    hipStream_t s = as_stream(stream);
    const int block = c.block_size;
    int pol = c.nontemporal & 3;
    // The fused <y, w> instance runs inside a solver, between vector kernels whose vectors (p, r, y: 240 MB) would fit the
    // 256 MiB Infinity Cache if the matrix streams did not push them out: here the once-read index / value streams carry the nt
    // hint whatever the table says for the stand-alone multiply (where plain loads measured equal or better).  CG iteration on the
    // headline matrix 263-268 -> 255-257 us, with the 16-bit column copy 248-251 -> 245-248 (archive/profiles/r02_cg_dot_policy.txt;
    // the y-store hint and load hints in the vector kernels measured no effect: r02_cg_y_store_policy.txt,
    // r02_cg_vector_load_policy.txt).  $CMI_DOT_POLICY=0..3 overrides (measurements).
    if (w && dot_partial) {
        static const int dot_pol = [] { const char *e = std::getenv("CMI_DOT_POLICY"); return e ? std::atoi(e) & 3 : -1; }();
        // (only for a matrix that does not itself fit the cache: see select_config's residency rule, tuning.hip)
        const bool resident = nnz * (int64_t)(sizeof(int) + sizeof(T)) <= kInfinityCacheBytes + kInfinityCacheBytes / 4;
        pol = dot_pol >= 0 ? dot_pol : resident ? pol : (pol | kPolLoadNT);
    }
    int st = CMI_SUCCESS;

    switch (c.kernel) {
    case CMI_CSR_SCALAR: {
        const int grid = grid_for(rows, block);
        with_policy(pol, [&](auto P) {
            hipLaunchKernelGGL((csr_scalar_kernel<T, decltype(P)::value>), dim3(grid), dim3(block), 0, s, rows, Ap, Aj, Ax, x, y, accumulate);
        });
        break;
    }
    case CMI_CSR_VECTOR: {
        const int tpr = c.threads_per_row;
        const int grid = grid_for(rows * tpr, block);
        st = (pol & kPolLoadNT) ? launch_vector<T, true>(tpr, grid, block, s, rows, Ap, Aj, Ax, x, y, accumulate)
                                : launch_vector<T, false>(tpr, grid, block, s, rows, Ap, Aj, Ax, x, y, accumulate);
        if (st) return st;
        break;
    }
    case CMI_CSR_STREAM: {
        int rpb = c.rows_per_block;
        const int ipt = c.items_per_thread;
        // The fused <y, w> instance wants whole waves of rows: with the table's 176 rows per tile (2.75 waves) the dot costs
        // +9.3 us on the headline matrix, with 192 (3 waves) +3.7 us (archive/tools/r2_probe.hip csrx flags 5 vs 1 at rpb 176 / 192,
        // archive/profiles/r02_probe_dot_ablation.txt).  So a table-chosen shape (not a caller's explicit one) is rounded up to the
        // next multiple of 64 rows when the tile's single LDS pass still holds them.
        if (w && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO || user->rows_per_block == 0) && c.threads_per_row <= 1 && rows > 0) {
            const int up = (rpb + kWave - 1) / kWave * kWave;
            const double mean = (double)nnz / (double)rows;
            const int64_t tile_entries = (int64_t)block * ipt * 4;
            if (up != rpb && up <= block && (double)up * mean + 3.0 <= (double)tile_entries) rpb = up;
        }
        // ... and one partial per tile must fit the workspace: a table shape with small tiles (f32's 96 rows of 128 lanes at 10^7
        // rows: 104 000 tiles) is doubled -- lanes and rows together, the same fill of the LDS pass -- until it does
        int block = c.block_size; // (shadows the function's: the dot instance may widen it)
        if (w && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO || user->rows_per_block == 0) && c.threads_per_row <= 1)
            while (ceil_div(rows, rpb) > kPartialCapacity && block * 2 <= 1024) { block *= 2; rpb *= 2; }
        int tpr = c.threads_per_row <= 1 ? 1 : c.threads_per_row;
        if (tpr > 64 || (tpr & (tpr - 1)) != 0) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_stream: threads_per_row must be 0/1 or a power of two <= 64");
        // threads_per_row == 0: rows of kLongRowPerLane entries per lane of a group (at least kLongRowMin) or more
        // are streamed by the whole workgroup (re-associated); == 1: storage order for every row, whatever its length
        const int long_len = c.threads_per_row == 1 ? 0 : (tpr * kLongRowPerLane > kLongRowMin ? tpr * kLongRowPerLane : kLongRowMin);
        if (rpb < 1 || rpb > 4 * (block / tpr)) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: rows_per_block must be in [1, 4*block_size/threads_per_row]");
        const int64_t tiles = ceil_div(rows, rpb);
        const int64_t tpx = ceil_div(tiles, kXcds);
        int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        // ... and its tiles go out in launch order: inside the solve that measured 2.3-3.5 us per iteration better than any chunk
        // dealing, for every tile shape (tools/cg_dot_shape_probe.py, archive/profiles/r02_cg_dot_shape.txt) -- stand-alone it is the
        // other way round (section 3.1 of DESIGN.md).  A caller's explicit shape is left alone; $CMI_DOT_SWIZZLE overrides.
        if (w && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO)) swz = dot_swizzle(swz, plan);
        const int64_t grid64 = swz == 0 ? tiles : swz == 1 ? tpx * kXcds : ceil_div(tiles, (int64_t)kXcds * swz) * kXcds * swz;
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: grid too large");
        const size_t lds = (size_t)block * ipt * 4 * sizeof(T) + (size_t)(rpb + 1) * sizeof(int);
        if (lds > 160 * 1024) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream: tile does not fit 160 KiB of LDS");
        const bool vec = (reinterpret_cast<uintptr_t>(Aj) % 16 == 0) && (reinterpret_cast<uintptr_t>(Ax) % 16 == 0);
        const bool dot = w && dot_partial && vec && tiles <= kPartialCapacity;
        // the LONG instance only for a matrix whose plan shows such a row (no plan: the ordinary instance, which sums
        // any row, one lane or lane group at a time -- correct, slow on a long row)
        const bool lng = long_len > 0 && nnz >= long_len && known_max_len >= long_len;
        const int strided = csr_lane_strided(c.nontemporal);
        // row sums dealt round the waves ($CMI_CSR_SPREAD=1: row r -> wave r % W; =2: a contiguous chunk of rows per wave): MEASURED SLOWER on
        // the long-row matrices it was meant for (ldoor-like 100 -> 110 us, nlpkkt120-like 213 -> 236 / 218 us, profiles/r03_long_rows_experiments.txt)
        // -- lane r adds row r stays the default; the switch stays for measurements
        static const int spread_env = [] { const char *e = std::getenv("CMI_CSR_SPREAD"); return e ? std::atoi(e) : 0; }();
        const int spread = tpr == 1 ? spread_env : 0;
        // f64 streams requested as (int2, double2) pairs instead of 16-byte vectors in the single-pass tile path: $CMI_CSR_PAIRS=0/1, else the
        // config's policy bit kPolPairs
        static const int pairs_env = [] { const char *e = std::getenv("CMI_CSR_PAIRS"); return e ? std::atoi(e) : -1; }();
        const int pairs = pairs_env >= 0 ? pairs_env : ((c.nontemporal & kPolPairs) != 0);
#define CMI_STREAM_GO(VEC_, DOT_, LONG_, ...) \
    launch_stream_ipt<T, VEC_, POL, DOT_, LONG_>(ipt, (int)grid64, block, lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, tpx, swz, accumulate, tpr, long_len, strided, spread, pairs, ##__VA_ARGS__)
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            if (dot) {
                st = lng ? CMI_STREAM_GO(true, true, true, w, dot_partial) : CMI_STREAM_GO(true, true, false, w, dot_partial);
                return;
            }
            if (vec) st = lng ? CMI_STREAM_GO(true, false, true) : CMI_STREAM_GO(true, false, false);
            else     st = lng ? CMI_STREAM_GO(false, false, true) : CMI_STREAM_GO(false, false, false);
        });
#undef CMI_STREAM_GO
        if (st) return st;
        if (dot && dot_partials) *dot_partials = (int)tiles;
        break;
    }
    case CMI_CSR_STREAM_WAVE: {
        const int K = c.items_per_thread;
        if (plan && plan->wave_row_start) { // irregular short rows: the plan's partition (above)
            if (K < 2 || K > kWaveTileMaxK) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_wave: items_per_thread (entries per lane) must be 2..10");
            const int64_t tiles = ceil_div(plan->wave_tiles, (int64_t)4);
            const int64_t tpx = ceil_div(tiles, kXcds);
            int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
            if (w && dot_partial) swz = dot_swizzle(swz, plan, swz);
            const int64_t grid64 = padded_grid(tiles, swz);
            if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_wave: grid too large");
            const bool dot = w && dot_partial && tiles <= kPartialCapacity;
            with_policy(pol, [&](auto P) {
                constexpr int POL = decltype(P)::value;
                auto go = [&](auto KK) {
                    constexpr int KC = decltype(KK)::value;
                    if (dot) hipLaunchKernelGGL((csr_wavep_kernel<T, KC, POL, true>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, w, dot_partial);
                    else     hipLaunchKernelGGL((csr_wavep_kernel<T, KC, POL, false>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
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
            if (dot && dot_partials) *dot_partials = (int)tiles;
            break;
        }
        const int waves = block / kWave;
        if (K < 2 || K > kWaveTileMaxK) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_wave: items_per_thread (entries per lane) must be 2..10");
        if (c.rows_per_block < waves || c.rows_per_block % waves != 0 || c.rows_per_block / waves > kWave)
            return fail(CMI_ERROR_INVALID_VALUE, "csr_wave: rows_per_block must be block_size/64 waves x 1..64 rows each");
        const int rpw = c.rows_per_block / waves;
        const int64_t tiles = ceil_div(rows, (int64_t)c.rows_per_block);
        const int64_t tpx = ceil_div(tiles, kXcds);
        int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        // (the wave-tile kernel keeps the table's chunk dealing inside a solve too: 239 against 243.5 us per CG iteration in launch
        //  order, archive/profiles/r02_cg_wave_dot.txt -- csr_stream's dot instance is the other way round)
        if (w && dot_partial && (!user || user->kernel == CMI_KERNEL_AUTO)) swz = dot_swizzle(swz, plan, swz);
        const int64_t grid64 = padded_grid(tiles, swz);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_wave: grid too large");
        const size_t lds = (size_t)block * K * sizeof(T);
        if (lds > 64 * 1024) return fail(CMI_ERROR_INVALID_VALUE, "csr_wave: block_size x items_per_thread products do not fit 64 KiB of LDS");
        const bool dot = w && dot_partial && tiles <= kPartialCapacity;
        static const int dot_ablate = [] { const char *e = std::getenv("CMI_DOT_ABLATE"); return e ? std::atoi(e) : 0; }();
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto KK) {
                constexpr int KC = decltype(KK)::value;
                if (dot) hipLaunchKernelGGL((csr_wave_kernel<T, KC, POL, true>), dim3((unsigned)grid64), dim3(block), lds, s, rows, Ap, Aj, Ax, x, y, rpw, tiles, tpx, swz, accumulate, w, dot_partial, dot_ablate);
                else     hipLaunchKernelGGL((csr_wave_kernel<T, KC, POL, false>), dim3((unsigned)grid64), dim3(block), lds, s, rows, Ap, Aj, Ax, x, y, rpw, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
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
        if (dot && dot_partials) *dot_partials = (int)tiles;
        break;
    }
    case CMI_CSR_STREAM_WAVEV: { // wave-private tiles, 16-byte-vector body, on the plan's row partition
        const int V = c.items_per_thread;
        if (!plan || !plan->wave_row_start || plan->wave_q <= 0) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_WAVEV runs through a plan (cmi_plan_create) only");
        if (V != 1 && V != 2 && V != 4) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_wavev: items_per_thread (index vectors per lane) must be 1, 2 or 4");
        if (reinterpret_cast<uintptr_t>(Aj) % 16 != 0 || reinterpret_cast<uintptr_t>(Ax) % 16 != 0) return fail(CMI_ERROR_INVALID_VALUE, "csr_wavev: Aj and Ax must be 16-byte aligned");
        static const int wpb_env = [] { const char *e = std::getenv("CMI_WAVEV_WPB"); const int v = e ? std::atoi(e) : 4; return v == 1 || v == 2 ? v : 4; }();
        const int wpb = V == 4 ? wpb_env : 4; // wave tiles per workgroup
        const int64_t tiles = ceil_div(plan->wave_tiles, (int64_t)wpb);
        const int64_t tpx = ceil_div(tiles, kXcds);
        int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        const int64_t grid64 = padded_grid(tiles, swz);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_wavev: grid too large");
        const bool dot = w && dot_partial && tiles <= kPartialCapacity;
        if (wpb != 4) {
            with_policy(pol, [&](auto P) {
                constexpr int POL = decltype(P)::value;
                auto go = [&](auto WW) {
                    constexpr int WC = decltype(WW)::value;
                    if (dot) hipLaunchKernelGGL((csr_wavev_kernel<T, 4, POL, true, 0, WC>), dim3((unsigned)grid64), dim3(kWave * WC), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, w, dot_partial);
                    else     hipLaunchKernelGGL((csr_wavev_kernel<T, 4, POL, false, 0, WC>), dim3((unsigned)grid64), dim3(kWave * WC), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
                };
                if (wpb == 1) go(std::integral_constant<int, 1>()); else go(std::integral_constant<int, 2>());
            });
            if (dot && dot_partials) *dot_partials = (int)tiles;
            break;
        }
        if constexpr (sizeof(T) == 8) { // measurements only: ablated instances (wrong results by design), f64 / V = 4 / nt loads and stores
            static const int ablate = [] { const char *e = std::getenv("CMI_WAVEV_ABLATE"); return e ? std::atoi(e) : 0; }();
            if (ablate > 0 && ablate < 16 && V == 4 && !dot) {
                auto run = [&](auto A) {
                    constexpr int AB = decltype(A)::value;
                    hipLaunchKernelGGL((csr_wavev_kernel<T, 4, 3, false, AB>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
                };
                switch (ablate) {
                case 1: run(std::integral_constant<int, 1>()); break;
                case 2: run(std::integral_constant<int, 2>()); break;
                case 3: run(std::integral_constant<int, 3>()); break;
                case 4: run(std::integral_constant<int, 4>()); break;
                case 5: run(std::integral_constant<int, 5>()); break;
                case 10: {
                    static const int ldiv = [] { const char *e = std::getenv("CMI_WAVEV_LDIV"); return e ? std::atoi(e) : 1; }();
                    auto run_l = [&](auto L) {
                        constexpr int LD = decltype(L)::value;
                        hipLaunchKernelGGL((csr_wavev_kernel<T, 4, 3, false, 10, 4, LD>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
                    };
                    switch (ldiv) {
                    case 2: run_l(std::integral_constant<int, 2>()); break;
                    case 4: run_l(std::integral_constant<int, 4>()); break;
                    case 8: run_l(std::integral_constant<int, 8>()); break;
                    case 16: run_l(std::integral_constant<int, 16>()); break;
                    default: run_l(std::integral_constant<int, 1>()); break;
                    }
                    break;
                }
                case 11: run(std::integral_constant<int, 11>()); break;
                default: return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_WAVEV_ABLATE: 1..5, 10, 11");
                }
                break;
            }
        }
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto VV) {
                constexpr int VC = decltype(VV)::value;
                if (dot) hipLaunchKernelGGL((csr_wavev_kernel<T, VC, POL, true>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, w, dot_partial);
                else     hipLaunchKernelGGL((csr_wavev_kernel<T, VC, POL, false>), dim3((unsigned)grid64), dim3(256), 0, s, plan->wave_row_start, plan->wave_tiles, nnz, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, (const T *)nullptr, (double *)nullptr);
            };
            switch (V) {
            case 1: go(std::integral_constant<int, 1>()); break;
            case 2: go(std::integral_constant<int, 2>()); break;
            default: go(std::integral_constant<int, 4>()); break;
            }
        });
        if (dot && dot_partials) *dot_partials = (int)tiles;
        break;
    }
    case CMI_CSR_STREAM_WAVEX: { // csr_wavev + an x window in LDS per workgroup (gather-bound band matrices)
        const int V = c.items_per_thread;
        if (!plan || !plan->wave_row_start || plan->wave_q <= 0) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_WAVEX runs through a plan (cmi_plan_create) only");
        if (V != 2 && V != 4) return fail(CMI_ERROR_NOT_SUPPORTED, "csr_wavex: items_per_thread (index vectors per lane) must be 2 or 4");
        if (reinterpret_cast<uintptr_t>(Aj) % 16 != 0 || reinterpret_cast<uintptr_t>(Ax) % 16 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0)
            return fail(CMI_ERROR_INVALID_VALUE, "csr_wavex: Aj, Ax and x must be 16-byte aligned");
        const int xe = 16 / (int)sizeof(T);
        int window = c.rows_per_block > 0 ? c.rows_per_block : 4096; // (the config's rows_per_block field carries the window length for this kernel)
        window = (window + 256 * xe - 1) / (256 * xe) * (256 * xe);
        if (window > 8 * 256 * xe) window = 8 * 256 * xe;
        const int64_t tiles = ceil_div(plan->wave_tiles, (int64_t)4);
        const int64_t tpx = ceil_div(tiles, kXcds);
        const int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        const int64_t grid64 = padded_grid(tiles, swz);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_wavex: grid too large");
        const size_t lds = ((size_t)window + (size_t)4 * kWave * V * 4) * sizeof(T);
        const bool dot = w && dot_partial && tiles <= kPartialCapacity;
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            auto go = [&](auto VV) {
                constexpr int VC = decltype(VV)::value;
                if (dot) hipLaunchKernelGGL((csr_wavex_kernel<T, VC, POL, true>), dim3((unsigned)grid64), dim3(256), lds, s, plan->wave_row_start, plan->wave_tiles, nnz, rows, cols, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, window, w, dot_partial);
                else     hipLaunchKernelGGL((csr_wavex_kernel<T, VC, POL, false>), dim3((unsigned)grid64), dim3(256), lds, s, plan->wave_row_start, plan->wave_tiles, nnz, rows, cols, Ap, Aj, Ax, x, y, tiles, tpx, swz, accumulate, window, (const T *)nullptr, (double *)nullptr);
            };
            if (V == 2) go(std::integral_constant<int, 2>()); else go(std::integral_constant<int, 4>());
        });
        if (dot && dot_partials) *dot_partials = (int)tiles;
        break;
    }
    case CMI_CSR_STREAM_WAVER:
    case CMI_CSR_STREAM_PACKED: { // the plan's run-compressed column copy on wave tiles (spmv_csr_runs.hip); Aj is read by the array-tail fall-back only
        if (!plan) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_WAVER / _PACKED run through a plan of cmi_plan_create_csr only");
        if (plan->csr16_packed) { // stencil-like rows: packed wave tiles of the 16-bit copy (spmv_csr16.hip); neither Ap nor Aj nor Ax is read
            const int swz16 = (w && dot_partial) ? dot_swizzle(c.xcd_swizzle, plan, c.xcd_swizzle) : c.xcd_swizzle;
            if constexpr (std::is_same<T, double>::value) return csr16_multiply_f64(plan, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swz16);
            else return csr16_multiply_f32(plan, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, swz16);
        }
        if constexpr (std::is_same<T, double>::value) return csr_runs_multiply_f64(plan, Ap, Aj, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, c.xcd_swizzle);
        else return csr_runs_multiply_f32(plan, Ap, Aj, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, c.xcd_swizzle);
    }
    case CMI_CSR_STREAM_C16: { // the plan's 16-bit column copy (spmv_csr16.hip); Aj itself is not read
        if (!plan || !plan->csr16_cols) return fail(CMI_ERROR_NOT_SUPPORTED, "CMI_CSR_STREAM_C16 runs through a plan of cmi_plan_create_csr only");
        if (reinterpret_cast<uintptr_t>(Ax) % 16 != 0) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_c16: values must be 16-byte aligned");
        if constexpr (std::is_same<T, double>::value) return csr16_multiply_f64(plan, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, (w && dot_partial) ? dot_swizzle(c.xcd_swizzle, plan, plan->csr16_wave_k > 0 ? c.xcd_swizzle : 0) : c.xcd_swizzle);
        else return csr16_multiply_f32(plan, Ap, Ax, x, y, accumulate, s, w, dot_partial, dot_partials, pol, (w && dot_partial) ? dot_swizzle(c.xcd_swizzle, plan, plan->csr16_wave_k > 0 ? c.xcd_swizzle : 0) : c.xcd_swizzle);
    }
    case CMI_CSR_STREAM_PIPE: {
        const int rpb = c.rows_per_block;
        if (rpb < 1 || rpb >= block) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_pipe: rows_per_block must be in [1, block_size-1]");
        const bool vec = (reinterpret_cast<uintptr_t>(Aj) % 16 == 0) && (reinterpret_cast<uintptr_t>(Ax) % 16 == 0);
        if (!vec || nnz < 4) return fail(CMI_ERROR_INVALID_VALUE, "csr_stream_pipe: needs 16-byte aligned Aj/Ax and >= 4 entries (use CMI_CSR_STREAM)");
        const int64_t tiles = ceil_div(rows, rpb);
        const size_t lds = 2 * ((size_t)block * 4 * sizeof(T) + (size_t)block * sizeof(int));
        const int bpc = c.blocks_per_cu > 0 ? c.blocks_per_cu : 8;
        int64_t grid64 = (int64_t)kCus * bpc;
        if (grid64 > tiles) grid64 = tiles;
        const int chunked = c.xcd_swizzle != 0;
        with_policy(pol, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            if (accumulate) hipLaunchKernelGGL((csr_stream_pipe_kernel<T, POL, true>), dim3((int)grid64), dim3(block), lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, chunked);
            else            hipLaunchKernelGGL((csr_stream_pipe_kernel<T, POL, false>), dim3((int)grid64), dim3(block), lds, s, rows, nnz, Ap, Aj, Ax, x, y, rpb, tiles, chunked);
        });
        break;
    }
    case CMI_CSR_BALANCED: {
        if (rows + nnz > ((int64_t)1 << 40)) return fail(CMI_ERROR_INVALID_VALUE, "csr_balanced: matrix too large");
        const int64_t tiles = ceil_div(rows + nnz, kBalItems);
        // A workgroup walks `per` consecutive tiles (one search of the row offsets, then tile ends chain) and the
        // workgroups are dealt in launch order, so the tiles in flight form ONE window sweeping the arrays --
        // measured 2x faster than giving each of 2048 resident workgroups its own distant chunk (2048 DRAM fronts).
        // blocks_per_cu > 0 asks for that persistent shape instead (grid = CUs * blocks_per_cu).
        int64_t per = c.items_per_thread > 0 ? c.items_per_thread : 4;
        int64_t grid64 = ceil_div(tiles, per);
        if (c.blocks_per_cu > 0) {
            grid64 = (int64_t)kCus * c.blocks_per_cu;
            if (grid64 > tiles) grid64 = tiles;
            per = ceil_div(tiles, grid64);
        }
        const int64_t chunks = grid64;
        const int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        const int64_t cpx = ceil_div(chunks, kXcds);
        grid64 = padded_grid(chunks, swz);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "csr_balanced: grid too large");
        if (!accumulate) // rows split across tiles are completed with atomics: they start from zero
            hipLaunchKernelGGL((zero_fill_kernel<T>), dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, s, rows, y);
        const bool vec = (reinterpret_cast<uintptr_t>(Aj) % 16 == 0) && (reinterpret_cast<uintptr_t>(Ax) % 16 == 0);
        if (vec) hipLaunchKernelGGL((csr_balanced_kernel<T, true>), dim3((unsigned)grid64), dim3(kBalBlock), 0, s, rows, nnz, Ap, Aj, Ax, x, y, tiles, per, accumulate, chunks, cpx, swz);
        else     hipLaunchKernelGGL((csr_balanced_kernel<T, false>), dim3((unsigned)grid64), dim3(kBalBlock), 0, s, rows, nnz, Ap, Aj, Ax, x, y, tiles, per, accumulate, chunks, cpx, swz);
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

namespace cmi {
static int plan_is_csr(const cmi_plan *plan, int dtype, const char *who)
{
    if (!plan) { set_error("%s: null plan", who); return CMI_ERROR_INVALID_VALUE; }
    if (plan->format != CMI_FORMAT_CSR || plan->dtype != dtype) { set_error("%s: the plan was made for another format or value type", who); return CMI_ERROR_INVALID_VALUE; }
    return CMI_SUCCESS;
}

// y <- A x and *dot_dev <- <y, w> in one pass where the selected kernel can (csr_stream, aligned
// arrays); otherwise the plain SpMV followed by the library's dot.  Either way deterministic.
template <typename T>
static int spmv_csr_dot(int dtype, int64_t rows, int64_t cols, int64_t nnz, const int *Ap, const int *Aj, const T *Ax, const T *x,
                        T *y, const T *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream, const cmi_plan *plan)
{
    if ((!w && rows > 0) || !dot_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr_dot: null w, result or workspace");
    int partials = 0;
    const int st = spmv_csr<T>(dtype, rows, cols, nnz, Ap, Aj, Ax, x, y, 0, cfg, stream, w, (double *)workspace, &partials, plan);
    if (st) return st;
    if (partials > 0) {
        const int st2 = reduce_partials_f64(partials, (double *)workspace, dot_dev, as_stream(stream));
        if (st2) return st2;
        CMI_LAUNCH_CHECK("csr spmv dot");
        return CMI_SUCCESS;
    }
    if constexpr (std::is_same<T, double>::value) return cmi_blas_dot_f64(rows, y, w, dot_dev, workspace, stream);
    else return cmi_blas_dotd_f32(rows, y, w, dot_dev, workspace, stream);
}
} // namespace cmi

CMI_API int cmi_spmv_csr_plan_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                                  const double *x, double *y, int accumulate, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F64, "cmi_spmv_csr_plan_f64")) return st;
    return cmi::spmv_csr<double>(CMI_F64, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, accumulate, nullptr, stream, nullptr, nullptr, nullptr, plan);
}
CMI_API int cmi_spmv_csr_plan_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                                  const float *x, float *y, int accumulate, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F32, "cmi_spmv_csr_plan_f32")) return st;
    return cmi::spmv_csr<float>(CMI_F32, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, accumulate, nullptr, stream, nullptr, nullptr, nullptr, plan);
}

CMI_API int cmi_spmv_csr_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                                 const int32_t *Aj, const double *Ax, const double *x, double *y, const double *w,
                                 double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_csr_dot<double>(CMI_F64, num_rows, num_cols, num_entries, Ap, Aj, Ax, x, y, w, dot_dev, workspace, cfg, stream, nullptr);
}
CMI_API int cmi_spmv_csr_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                                 const int32_t *Aj, const float *Ax, const float *x, float *y, const float *w,
                                 double *dot_dev, void *workspace, const cmi_config *cfg, void *stream)
{
    return cmi::spmv_csr_dot<float>(CMI_F32, num_rows, num_cols, num_entries, Ap, Aj, Ax, x, y, w, dot_dev, workspace, cfg, stream, nullptr);
}
CMI_API int cmi_spmv_csr_dot_plan_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                                      const double *x, double *y, const double *w, double *dot_dev, void *workspace, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F64, "cmi_spmv_csr_dot_plan_f64")) return st;
    return cmi::spmv_csr_dot<double>(CMI_F64, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, w, dot_dev, workspace, nullptr, stream, plan);
}
CMI_API int cmi_spmv_csr_dot_plan_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                                      const float *x, float *y, const float *w, double *dot_dev, void *workspace, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F32, "cmi_spmv_csr_dot_plan_f32")) return st;
    return cmi::spmv_csr_dot<float>(CMI_F32, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, w, dot_dev, workspace, nullptr, stream, plan);
}

// y <- A x and the per-tile partials of <y, w> LEFT in the workspace (no fold): the fold rides at the front of
// cmi_cg_update_fold_* (blas1.hip).  *npartials = their count, or 0 when the plan's kernel cannot fuse the dot (y is computed;
// the caller then runs cmi_blas_dot_* and the plain cmi_cg_update_*).
CMI_API int cmi_spmv_csr_dot_plan_partials_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                                               const double *x, double *y, const double *w, void *workspace, int *npartials, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F64, "cmi_spmv_csr_dot_plan_partials_f64")) return st;
    if ((!w && plan->rows > 0) || !workspace || !npartials) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr_dot_plan_partials: null w, workspace or count");
    return cmi::spmv_csr<double>(CMI_F64, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, 0, nullptr, stream, w, (double *)workspace, npartials, plan);
}
CMI_API int cmi_spmv_csr_dot_plan_partials_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                                               const float *x, float *y, const float *w, void *workspace, int *npartials, void *stream)
{
    if (int st = cmi::plan_is_csr(plan, CMI_F32, "cmi_spmv_csr_dot_plan_partials_f32")) return st;
    if ((!w && plan->rows > 0) || !workspace || !npartials) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_csr_dot_plan_partials: null w, workspace or count");
    return cmi::spmv_csr<float>(CMI_F32, plan->rows, plan->cols, plan->nnz, Ap, Aj, Ax, x, y, 0, nullptr, stream, w, (double *)workspace, npartials, plan);
}

// Longest row of a CSR matrix (device pass + read-back; synchronises the stream).  A plan does this itself;
// exposed for hosts that keep their own profile.
CMI_API int cmi_csr_max_row_length(int64_t num_rows, const int32_t *Ap, int64_t *max_length_host, void *stream)
{
    if (num_rows < 0 || !max_length_host) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_max_row_length: bad argument");
    *max_length_host = 0;
    if (num_rows == 0) return CMI_SUCCESS;
    if (!Ap) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_max_row_length: null row offsets");
    return cmi::measure_row_lengths(num_rows, Ap, cmi::as_stream(stream), max_length_host, nullptr, nullptr);
}
