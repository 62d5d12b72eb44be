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

// ---------------------------------------------------------------------------------------------
// coo_tile: ROW-SORTED entries, no zero fill, no atomics, storage-order sums
// ---------------------------------------------------------------------------------------------
// The order-agnostic kernels above pay for not knowing the order: y is zero-filled first (80 MB on the headline
// matrix) and every run end is a read-modify-write at the memory side -- 1.10 GB moved for 0.96 GB of compulsory bytes
// (archive/profiles/r02_formats_pmc_before.json).  The reference's contract for coo_matrix IS sorted entries
// (cusp/coo_matrix.h:72); a plan (plan.hip) checks that once, and then this kernel runs: csr_stream's single-pass tile
// with the row pointers built on the fly from the row indices.
//
// A workgroup of 256 lanes owns tile t = entries [E0, E1) = 1024 consecutive entries, and with them every row that
// STARTS there, plus the rows without entries in front of each such row; the last tile also owns the empty rows behind
// the last entry.  Owned rows are the contiguous range [row(E0 - 1) + 1, row(E1 - 1)], known from two uniform loads.
//   1. every lane requests four entries as 16-byte vectors (row, column, value); sixteen lanes of wave 0 ALSO request the
//      64 entries BEHIND the tile -- the tail of the row that straddles into the next tile (those lines are the
//      neighbour's: an L2 hit when the tiles share an XCD) -- before anyone waits, so that wave 0 pays no second round
//      trip; then x is gathered, products and row indices are parked in LDS, and every lane flags the entries that
//      START a row (row differs from the predecessor's: the left neighbour's last entry, by DPP);
//   2. still without a barrier, each WAVE compacts the positions of the row starts among its own 256 entries into its
//      own list (a wave scan of the flag counts) and publishes the first of them;
//   3. ONE barrier; lane i of wave w adds the products of the i-th row that starts in wave w's entries, positions
//      [start_i, start_i+1) -- the wave's last row ends at the first start of a later wave, or in the parked tail, or,
//      if longer than that, is walked on through the arrays -- IN STORAGE ORDER with the LDS reads batched
//      (sum_in_order): the host loop's order (sequential/multiply/coo_spmv.h:60-66: y[i] = y[i] + V*x for the entries
//      as stored), bit-identical to it.  Consecutive lanes hold consecutive rows: y is stored in contiguous runs (nt).
// Rows without entries inside the owned range are rare: when there are any (owned range longer than the number of row
// starts: a uniform test) the range is zero-filled cooperatively first (not when accumulating).
constexpr int kCooTile = 1024, kCooTail = 64, kCooBlock = 256, kCooWaves = kCooBlock / kWave, kCooPerWave = kCooTile / kCooWaves;

template <typename T> struct coo_vec { int r[4], c[4]; T v[4]; };

template <typename T, int POL, bool ACC>
__global__ void __launch_bounds__(kCooBlock)
coo_tile_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ai, const int *__restrict__ Aj,
                const T *__restrict__ Ax, const T *__restrict__ x, T *__restrict__ y, int64_t tiles, int64_t tiles_per_xcd,
                int swizzle)
{
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    __shared__ __attribute__((aligned(16))) T prod[kCooTile + kCooTail];
    __shared__ __attribute__((aligned(16))) int rowidx[kCooTile + kCooTail]; // row of entry E0 + k
    __shared__ unsigned short starts[kCooWaves][kCooPerWave];                 // per wave: positions of its row starts, ascending
    __shared__ int wave_count[kCooWaves], wave_first[kCooWaves];              // per wave: number of starts, the first one (or 1 << 30)
    __shared__ int tail_first;                                                // first row start among the parked tail, or 1 << 30
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= tiles) return;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t E0 = tile * kCooTile;
    const int64_t E1 = E0 + kCooTile < num_entries ? E0 + kCooTile : num_entries;
    const int n_main = (int)(E1 - E0);
    const int n_tail = n_main == kCooTile ? (int)((num_entries - E1) < kCooTail ? (num_entries - E1) : kCooTail) : 0;
    // owned rows (uniform loads).  Clamped into [0, num_rows): with the sorted, in-range indices a plan vouches for the
    // clamps never bind; with anything else they keep every access inside y.
    int64_t prev_row = E0 > 0 ? Ai[E0 - 1] : -1;
    int64_t last_row = E1 == num_entries ? num_rows - 1 : Ai[E1 - 1];
    if (prev_row < -1) prev_row = -1;
    if (prev_row >= num_rows) prev_row = num_rows - 1;
    if (last_row >= num_rows) last_row = num_rows - 1;
    const int64_t Rfirst = prev_row + 1;

    // ---- 1. request the tile (+ its tail), then gather, park and flag ---------------------------------------------
    // Entries [e, e + 4), `live` of them exist.  BRANCH-FREE: every lane issues the same three 16-byte loads, from
    // min(e, num_entries - 4) (the launcher guarantees num_entries >= 4) -- only the array's last, partial vector and lanes
    // past the end are moved, by `shift` entries, and pick their entries out of the loaded ones afterwards.  (A vector
    // path and a scalar path side by side made the compiler drain the loads at the join: the tail's loads then started
    // only after the tile's had returned.)  The moved loads are 4-byte aligned: int4u / vec_u.
    typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef T vec2u __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    typedef T vec4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
    auto request = [&](int64_t e, int live, coo_vec<T> &q) {
        const int64_t last = num_entries - 4;
        const int64_t el = e < last ? e : last;
        const int shift = (int)(e - el); // 0 for every whole vector
#define CMI_LDU(TYPE, PTR) (NT ? __builtin_nontemporal_load(reinterpret_cast<const TYPE *>(PTR)) : *reinterpret_cast<const TYPE *>(PTR))
        const int4u rv = CMI_LDU(int4u, Ai + el); // (spelled out: a template parameter would drop the typedef's alignment)
        const int4u cv = CMI_LDU(int4u, Aj + el);
        int r[8] = {rv.x, rv.y, rv.z, rv.w, -2, -2, -2, -2}, c[8] = {cv.x, cv.y, cv.z, cv.w, 0, 0, 0, 0};
        T v[8];
        if constexpr (sizeof(T) == 8) {
            const vec2u a = CMI_LDU(vec2u, Ax + el);
            const vec2u b = CMI_LDU(vec2u, Ax + el + 2);
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        } else {
            const vec4u a = CMI_LDU(vec4u, Ax + el);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        }
#undef CMI_LDU
        v[4] = v[5] = v[6] = v[7] = T(0);
#pragma unroll
        for (int k = 0; k < 4; k++) { // entry k of this lane = loaded entry k + shift (selects; shift is 0 almost everywhere)
            const int j = k + (shift < 4 ? shift : 4);
            int rr = r[k], cc = c[k];
            T vv = v[k];
#pragma unroll
            for (int t = 1; t <= 4; t++)
                if (j == k + t) { rr = r[k + t]; cc = c[k + t]; vv = v[k + t]; }
            const bool dead = k >= live; // -2: no such entry (never equal to a row); column 0, value 0
            q.r[k] = dead ? -2 : rr;
            q.c[k] = dead ? 0 : cc;
            q.v[k] = dead ? T(0) : vv;
        }
    };
    // products and row indices into slots [slot, slot + 4); returns the flags of the row starts (bit k)
    auto park = [&](int64_t e, int slot, int live, const coo_vec<T> &q) -> int {
        // the four gathers are UNCONDITIONAL (an entry that does not exist has column 0 and value 0; its slot is never summed):
        // a gather behind a per-entry test is serialised by the compiler -- four round trips instead of one
        T p[4];
        const T x0 = x[q.c[0]], x1 = x[q.c[1]], x2 = x[q.c[2]], x3 = x[q.c[3]];
        p[0] = q.v[0] * x0; p[1] = q.v[1] * x1; p[2] = q.v[2] * x2; p[3] = q.v[3] * x3;
        // the row in front of this vector: the previous lane's last entry (a live lane's left neighbour holds four entries);
        // the first lane of a wave reads it from the array
        int pr = wave_shift_up(q.r[3], -2);
        if (lane == 0) pr = live > 0 ? (e > 0 ? Ai[e - 1] : -1) : -2;
        int flags = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            prod[slot + k] = p[k];
            rowidx[slot + k] = q.r[k];
            if (k < live && q.r[k] != (k ? q.r[k - 1] : pr)) flags |= 1 << k;
        }
        return flags;
    };
    const int k0 = tid * 4;
    const int live = n_main - k0 >= 4 ? 4 : (n_main - k0 > 0 ? n_main - k0 : 0);
    const bool tail_lane = tid < kCooTail / 4; // sixteen lanes of wave 0
    const int t0 = tid * 4;
    const int tlive = tail_lane ? (n_tail - t0 >= 4 ? 4 : (n_tail - t0 > 0 ? n_tail - t0 : 0)) : 0;
    coo_vec<T> qm, qt;
    request(E0 + k0, live, qm);
    if (tail_lane) request(E1 + t0, tlive, qt); // in flight together with the main vectors
    const int flags = park(E0 + k0, k0, live, qm);

    // ---- 2. this wave's row starts, compacted (no barrier: the list is the wave's own) ---------------------------------
    const int cnt = __builtin_popcount(flags);
    const int incl = wave_inclusive_sum(cnt); // DPP scan (no LDS crossbar round trips)
    {
        int pos = incl - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (flags & (1 << k)) starts[wave][pos++] = (unsigned short)(k0 + k);
        const int first = wave_min_to_last(flags ? k0 + __builtin_ctz(flags) : 1 << 30, 1 << 30);
        if (lane == kWave - 1) { wave_count[wave] = incl; wave_first[wave] = first; }
    }
    if (wave == 0) { // the parked tail: first row start among its entries (wave-uniform branch)
        int first = 1 << 30;
        if (tail_lane) {
            const int tf = park(E1 + t0, kCooTile + t0, tlive, qt);
            if (tf) first = kCooTile + t0 + __builtin_ctz(tf);
        }
        first = wave_min_to_last(first, 1 << 30);
        if (lane == kWave - 1) tail_first = first;
    }
    __syncthreads();

    // ---- 3. one lane per row, storage order -------------------------------------------------------------------------
    const int my_count = wave_count[wave];
    int next_first = tail_first; // where this wave's last row ends: the first row start behind the wave's entries
    int nstarts = 0;
#pragma unroll
    for (int w = kCooWaves - 1; w >= 0; w--) {
        if (w > wave && wave_first[w] < (1 << 30)) next_first = wave_first[w];
        nstarts += wave_count[w];
    }
    // rows without entries inside the owned range (uniform): zero them first; the row sums then overwrite their own rows
    const int64_t owned = last_row - Rfirst + 1;
    if (!ACC && owned > nstarts) {
        for (int64_t g = Rfirst + tid; g <= last_row; g += kCooBlock) st<NTS>(y + g, T(0));
        __threadfence_block();
        __syncthreads();
    }
    for (int i = lane; i < my_count; i += kWave) { // (more than one turn only for rows of fewer than four entries)
        const int a = starts[wave][i];
        const int R = rowidx[a];
        int b;
        bool open_end = false; // the row runs past everything parked
        if (i + 1 < my_count) b = starts[wave][i + 1];
        else if (next_first < (1 << 30)) b = next_first;
        else { b = n_main + n_tail; open_end = n_tail == kCooTail; }
        if (R >= 0 && R < num_rows) {
            T s = ACC ? y[R] : T(0);
            s = sum_in_order(s, prod + a, b - a);
            if (open_end)
                for (int64_t e = E1 + kCooTail; e < num_entries && Ai[e] == R; e++) s = s + Ax[e] * x[Aj[e]];
            st<NTS>(y + R, s);
        }
    }
}

// Row indices non-decreasing and inside [0, rows)?  One pass, a flag (bit 0).  Bit 1: some row holds more than kCooTile
// entries (Ai[e] == Ai[e + kCooTile], meaningful for sorted input) -- the tile kernel finishes such a row with a serial walk
// past its tile (one lane, global loads: ~1 us per entry), so a plan keeps those matrices on the order-agnostic kernels
// (tools/hyb_fuse_probe.py: a power-law tail of 3000-entry rows took 640 us through the tile kernel).
__global__ void __launch_bounds__(256)
coo_sorted_kernel(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ai, int *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int mine = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_entries; e += stride) {
        const int r = Ai[e];
        mine |= r < 0 || r >= num_rows || (e > 0 && Ai[e - 1] > r);
        if (e + kCooTile < num_entries && Ai[e + kCooTile] == r) mine |= 2;
    }
    if (__any(mine & 1) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(bad, 1);
    if (__any(mine & 2) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(bad, 2);
}

int coo_rows_sorted(int64_t rows, int64_t nnz, const int *Ai, hipStream_t s, int *sorted, int *long_runs)
{
    *sorted = 0;
    if (long_runs) *long_runs = 0;
    int *flag = nullptr;
    CMI_HIP(hipMalloc((void **)&flag, sizeof(int)));
    int host = 1;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(nnz, 256 * 8);
        if (blocks > kCus * 16) blocks = kCus * 16;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(coo_sorted_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, nnz, Ai, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(flag);
    if (e != hipSuccess) return hip_fail(e, "coo order check");
    *sorted = (host & 1) == 0;
    if (long_runs) *long_runs = (host & 2) != 0;
    return CMI_SUCCESS;
}

template <typename T>
static int spmv_coo(int dtype, int64_t rows, int64_t cols, int64_t nnz, const int *Ai, const int *Aj, const T *Ax,
                    const T *x, T *y, int accumulate, const cmi_config *user, void *stream, const cmi_plan *plan = nullptr)
{
    if (rows < 0 || cols < 0 || nnz < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: negative size");
    if (rows > INT32_MAX || cols > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: sizes exceed the int32 index type");
    if (rows == 0) return CMI_SUCCESS;
    if (!y || (nnz > 0 && (!Ai || !Aj || !Ax || !x))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: null array");
    if (plan && plan->coo_csr_plan) { // sorted entries: the plan's row offsets + the CSR kernels (the row indices are not read)
        if constexpr (std::is_same<T, double>::value) return cmi_spmv_csr_plan_f64(plan->coo_csr_plan, plan->coo_offsets, Aj, Ax, x, y, accumulate, stream);
        else return cmi_spmv_csr_plan_f32(plan->coo_csr_plan, plan->coo_offsets, Aj, Ax, x, y, accumulate, stream);
    }
    cmi_config c;
    if (plan) c = plan->cfg;
    else select_config(CMI_FORMAT_COO, dtype, rows, cols, nnz, user, &c);
    if (c.kernel != CMI_COO_SEGMENTED && c.kernel != CMI_COO_LANE4 && c.kernel != CMI_COO_TILE)
        return fail(CMI_ERROR_NOT_SUPPORTED, "cmi_spmv_coo: config.kernel is not a COO kernel");
    hipStream_t s = as_stream(stream);
    const bool aligned = reinterpret_cast<uintptr_t>(Ai) % 16 == 0 && reinterpret_cast<uintptr_t>(Aj) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(Ax) % 16 == 0;
    if (c.kernel == CMI_COO_TILE && aligned && nnz >= 4) { // sorted entries (a plan checked, or the caller's explicit config vouches)
        if (nnz > INT32_MAX - 4096) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: too many entries for the tile kernel");
        const int64_t tiles = ceil_div(nnz, kCooTile);
        const int swz = c.xcd_swizzle < 0 ? 0 : c.xcd_swizzle;
        const int64_t tpx = ceil_div(tiles, kXcds);
        const int64_t grid64 = padded_grid(tiles, swz);
        if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo: grid too large");
        with_policy(c.nontemporal & 3, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            if (accumulate) hipLaunchKernelGGL((coo_tile_kernel<T, POL, true>), dim3((unsigned)grid64), dim3(kCooBlock), 0, s, rows, nnz, Ai, Aj, Ax, x, y, tiles, tpx, swz);
            else            hipLaunchKernelGGL((coo_tile_kernel<T, POL, false>), dim3((unsigned)grid64), dim3(kCooBlock), 0, s, rows, nnz, Ai, Aj, Ax, x, y, tiles, tpx, swz);
        });
        CMI_LAUNCH_CHECK("coo tile spmv");
        return CMI_SUCCESS;
    }
    if (c.kernel == CMI_COO_TILE) { c.kernel = CMI_COO_LANE4; if (c.items_per_thread > 32 || c.items_per_thread < 1) c.items_per_thread = 4; } // unaligned views / no entries: the order-agnostic path
    // y = initialize(y): zero bytes are +0.0 (sequential/multiply/coo_spmv.h:56-57)
    if (!accumulate) CMI_HIP(hipMemsetAsync(y, 0, (size_t)rows * sizeof(T), s));
    if (nnz == 0) return CMI_SUCCESS;
    const int block = c.block_size;
    const int steps = c.items_per_thread < 1 ? 1 : c.items_per_thread; // steps per wave interval
    const int waves_per_block = block / kWave;
    const bool nt = (c.nontemporal & kPolLoadNT) != 0;
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

// ---------------------------------------------------------------------------------------------
// HYB in ONE launch (needs a plan: cmi_plan_create_hyb found the COO part sorted by row)
// ---------------------------------------------------------------------------------------------
// The reference runs HYB as two multiplies -- ELL with the caller's `initialize`, then COO accumulating on top
// (sequential/multiply/hyb_spmv.h:55-56; device: generic/multiply/spmv.h:275-290) -- so y is written, read again and
// written again, and the COO half needs its own zero-fill-free accumulate path.  Per row the arithmetic is ONE chain:
// init, + the ELL slots in slot order, + the row's COO entries in entry order.  Here a workgroup owns kHybTileRows
// consecutive rows, one lane per row: the lane walks its ELL slots (as ell_row_kernel does), then the workgroup stages the
// COO entries of ITS rows -- the plan's tile_start[] says which: entries [tile_start[t], tile_start[t+1]) -- through LDS
// 256 at a time, marks where each row's run begins and ends, and every lane adds its run to the same accumulator in entry
// order.  y is written once, with the bits of the host loops.  Bytes: ELL part + 16 per COO entry + 16 per row + 4 per tile.
// DOT: the workgroup also leaves sum_r y[r] * w[r] over its rows in dot_partial[tile] (the CG step <A p, p>, as csr_stream DOT).
template <typename T, int POL, bool ACC, bool DOT = false>
__global__ void __launch_bounds__(kHybTileRows)
hyb_tile_kernel(int64_t num_rows, int width, int64_t pitch, const int *__restrict__ eAj, const T *__restrict__ eAx,
                const int *__restrict__ cAi, const int *__restrict__ cAj, const T *__restrict__ cAx,
                const int32_t *__restrict__ tile_start, const T *__restrict__ x, T *__restrict__ y, int64_t tiles,
                int64_t tiles_per_xcd, int swizzle, const T *__restrict__ w = nullptr, double *__restrict__ dot_partial = nullptr)
{
    __shared__ double dot_slots[DOT ? kHybTileRows / kWave : 1];
    constexpr bool NT = (POL & kPolLoadNT) != 0, NTS = (POL & kPolStoreNT) != 0;
    constexpr int R = kHybTileRows;
    __shared__ T prod[R];
    __shared__ int lrow[R];
    __shared__ int sbeg[R], send[R];
    const int64_t tile = tile_of_block(blockIdx.x, tiles_per_xcd, swizzle);
    if (tile >= tiles) return;
    const int tid = threadIdx.x;
    const int64_t row0 = tile * R, row = row0 + tid;
    const bool live = row < num_rows;
    const int cb = tile_start[tile], ce = tile_start[tile + 1];
    // the first COO chunk's loads go out before the ELL walk (they do not depend on it)
    int ci = 0, cj = 0;
    T cv = T(0);
    if (cb + tid < ce) {
        ci = ld<NT>(cAi + cb + tid);
        cj = ld<NT>(cAj + cb + tid);
        cv = ld<NT>(cAx + cb + tid);
    }
    T cx = cb < ce ? x[cj] : T(0); // (lanes past the chunk gather x[0])
    T acc = T(0);
    if (live) {
        if (ACC) acc = y[row];
        auto chunk = [&](auto Kc, int n0) {
            constexpr int K = decltype(Kc)::value;
            int col[K];
            T val[K], xv[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                col[k] = ld<NT>(eAj + (int64_t)(n0 + k) * pitch + row);
                val[k] = ld<NT>(eAx + (int64_t)(n0 + k) * pitch + row);
            }
#pragma unroll
            for (int k = 0; k < K; k++) xv[k] = x[col[k] < 0 ? 0 : col[k]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; k++)
                if (col[k] != -1) acc = acc + val[k] * xv[k];
        };
        int n0 = 0;
        for (; n0 + 8 <= width; n0 += 8) chunk(std::integral_constant<int, 8>(), n0);
        if (n0 + 4 <= width) { chunk(std::integral_constant<int, 4>(), n0); n0 += 4; }
        if (n0 + 2 <= width) { chunk(std::integral_constant<int, 2>(), n0); n0 += 2; }
        if (n0 + 1 <= width) { chunk(std::integral_constant<int, 1>(), n0); }
    }
    for (int c0 = cb; c0 < ce; c0 += R) { // workgroup-uniform trip count
        const int n = ce - c0 < R ? ce - c0 : R;
        // the next chunk's three streams are requested now, its x gather after the first barrier (the column has arrived by
        // then): a chunk's global round trips overlap the LDS phase of the chunk before it
        const int c1 = c0 + R;
        const bool more = c1 < ce, mine = c1 + tid < ce;
        int ni = 0, nj = 0;
        T nv = T(0), nx = T(0);
        if (mine) {
            ni = ld<NT>(cAi + c1 + tid);
            nj = ld<NT>(cAj + c1 + tid);
            nv = ld<NT>(cAx + c1 + tid);
        }
        const int r = ci - (int)row0; // 0 .. R-1: the plan's tile_start[] brackets exactly this tile's rows
        sbeg[tid] = 0;
        send[tid] = 0;
        if (tid < n) {
            prod[tid] = cv * cx;
            lrow[tid] = r;
        }
        __syncthreads();
        if (more) nx = x[nj]; // (lanes past the chunk gather x[0])
        if (tid < n) {
            if (tid == 0 || lrow[tid - 1] != r) sbeg[r] = tid;
            if (tid == n - 1 || lrow[tid + 1] != r) send[r] = tid + 1;
        }
        __syncthreads();
        const int a = sbeg[tid], b = send[tid];
        acc = sum_in_order(acc, prod + a, b - a); // (rows without entries in this chunk: a == b == 0)
        if (more) {
            __syncthreads(); // this chunk's runs have been added: prod / lrow / sbeg / send are free again
            ci = ni; cj = nj; cv = nv; cx = nx;
        }
    }
    if (live) st<NTS>(y + row, acc);
    if constexpr (DOT) {
        tile_dot_store(live ? (double)acc * (double)w[row] : 0.0, dot_slots, dot_partial + tile);
        if (tile == 0 && tid == 0) reset_fold_state(dot_partial);
    }
}

// tile_start[t] = first COO entry whose row is >= t * kHybTileRows (entries sorted by row), t = 0 .. tiles
__global__ void __launch_bounds__(256)
hyb_tile_start_kernel(int64_t tiles, int coo_entries, const int *__restrict__ Ai, int32_t *__restrict__ tile_start)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles) return;
    const int64_t first_row = t * kHybTileRows;
    int lo = 0, hi = coo_entries; // lower bound
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (Ai[mid] < first_row) lo = mid + 1; else hi = mid;
    }
    tile_start[t] = lo;
}

// most COO entries any one tile holds
__global__ void __launch_bounds__(256)
hyb_tile_max_kernel(int64_t tiles, const int32_t *__restrict__ tile_start, int *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int n = t < tiles ? tile_start[t + 1] - tile_start[t] : 0;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { const int v = __shfl_down(n, o); n = v > n ? v : n; }
    if ((threadIdx.x & (kWave - 1)) == 0 && n > 0) atomicMax(out, n);
}

// tile_start[] for the COO part (sorted by row) and *max_in_tile_dev <- the most entries of one tile (zeroed here)
int hyb_tile_starts(int64_t rows, int64_t coo_entries, const int *coo_Ai, int32_t *tile_start, int *max_in_tile_dev, hipStream_t s)
{
    const int64_t tiles = ceil_div(rows, kHybTileRows);
    CMI_HIP(hipMemsetAsync(max_in_tile_dev, 0, sizeof(int), s));
    hipLaunchKernelGGL(hyb_tile_start_kernel, dim3((unsigned)ceil_div(tiles + 1, 256)), dim3(256), 0, s, tiles, (int)coo_entries, coo_Ai, tile_start);
    hipLaunchKernelGGL(hyb_tile_max_kernel, dim3((unsigned)ceil_div(tiles, 256)), dim3(256), 0, s, tiles, tile_start, max_in_tile_dev);
    CMI_LAUNCH_CHECK("hyb tile starts");
    return CMI_SUCCESS;
}

template <typename T>
static int spmv_hyb_plan(const cmi_plan *plan, int dtype, int64_t pitch, const int *eAj, const T *eAx, const int *cAi,
                         const int *cAj, const T *cAx, const T *x, T *y, int accumulate, void *stream,
                         const T *wdot = nullptr, double *dot_partial = nullptr, int *dot_partials = nullptr)
{
    if (dot_partials) *dot_partials = 0; // > 0: the one-launch kernel left that many partials of <y, wdot>
    if (!plan || plan->format != CMI_FORMAT_HYB || plan->dtype != dtype)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_hyb_plan: null plan, or a plan made for another format or value type");
    const int64_t rows = plan->rows, width = plan->hyb_width, coo = plan->hyb_coo;
    if (!plan->hyb_tile_start) { // COO part empty or not sorted by row: the two launches of cmi_spmv_hyb_*
        int st;
        if constexpr (std::is_same<T, double>::value) st = cmi_spmv_ell_f64(rows, plan->cols, width, pitch, eAj, eAx, nullptr, x, y, accumulate, &plan->cfg, stream);
        else st = cmi_spmv_ell_f32(rows, plan->cols, width, pitch, eAj, eAx, nullptr, x, y, accumulate, &plan->cfg, stream);
        if (st || coo == 0) return st;
        if (plan->hyb_coo_plan) return spmv_coo<T>(dtype, rows, plan->cols, coo, cAi, cAj, cAx, x, y, 1, nullptr, stream, plan->hyb_coo_plan);
        return spmv_coo<T>(dtype, rows, plan->cols, coo, cAi, cAj, cAx, x, y, 1, &plan->hyb_coo_cfg, stream);
    }
    if (rows == 0) return CMI_SUCCESS;
    if (width > 0 && pitch < rows) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_hyb_plan: pitch < num_rows");
    if (!y || !x || (width > 0 && (!eAj || !eAx)) || !cAi || !cAj || !cAx) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_hyb_plan: null array");
    const int64_t tiles = ceil_div(rows, kHybTileRows);
    const int swz = plan->cfg.xcd_swizzle < 0 ? 0 : plan->cfg.xcd_swizzle;
    const int64_t tpx = ceil_div(tiles, kXcds);
    const int64_t grid64 = padded_grid(tiles, swz);
    if (grid64 > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_hyb_plan: grid too large");
    hipStream_t s = as_stream(stream);
    if (wdot && dot_partial && !accumulate && tiles <= kPartialCapacity) {
        with_policy(plan->cfg.nontemporal & 3, [&](auto P) {
            constexpr int POL = decltype(P)::value;
            hipLaunchKernelGGL((hyb_tile_kernel<T, POL, false, true>), dim3((unsigned)grid64), dim3(kHybTileRows), 0, s, rows, (int)width, pitch, eAj, eAx, cAi, cAj, cAx, plan->hyb_tile_start, x, y, tiles, tpx, swz, wdot, dot_partial);
        });
        CMI_LAUNCH_CHECK("hyb tile spmv dot");
        if (dot_partials) *dot_partials = (int)tiles;
        return CMI_SUCCESS;
    }
    with_policy(plan->cfg.nontemporal & 3, [&](auto P) {
        constexpr int POL = decltype(P)::value;
        if (accumulate) hipLaunchKernelGGL((hyb_tile_kernel<T, POL, true>), dim3((unsigned)grid64), dim3(kHybTileRows), 0, s, rows, (int)width, pitch, eAj, eAx, cAi, cAj, cAx, plan->hyb_tile_start, x, y, tiles, tpx, swz);
        else            hipLaunchKernelGGL((hyb_tile_kernel<T, POL, false>), dim3((unsigned)grid64), dim3(kHybTileRows), 0, s, rows, (int)width, pitch, eAj, eAx, cAi, cAj, cAx, plan->hyb_tile_start, x, y, tiles, tpx, swz);
    });
    CMI_LAUNCH_CHECK("hyb tile spmv");
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

CMI_API int cmi_spmv_coo_plan_f64(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const double *Ax,
                                  const double *x, double *y, int accumulate, void *stream)
{
    if (!plan || plan->format != CMI_FORMAT_COO || plan->dtype != CMI_F64)
        return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo_plan_f64: null plan, or a plan made for another format or value type");
    return cmi::spmv_coo<double>(CMI_F64, plan->rows, plan->cols, plan->nnz, Ai, Aj, Ax, x, y, accumulate, nullptr, stream, plan);
}
CMI_API int cmi_spmv_coo_plan_f32(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const float *Ax,
                                  const float *x, float *y, int accumulate, void *stream)
{
    if (!plan || plan->format != CMI_FORMAT_COO || plan->dtype != CMI_F32)
        return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo_plan_f32: null plan, or a plan made for another format or value type");
    return cmi::spmv_coo<float>(CMI_F32, plan->rows, plan->cols, plan->nnz, Ai, Aj, Ax, x, y, accumulate, nullptr, stream, plan);
}

// y <- A x and *dot_dev <- <y, w> (a double) through a COO plan: sorted entries run the CSR kernel's fused dot on the plan's row
// offsets; otherwise the multiply followed by the library's dot.
CMI_API int cmi_spmv_coo_dot_plan_f64(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const double *Ax, const double *x,
                                      double *y, const double *w, double *dot_dev, void *workspace, void *stream)
{
    if (!plan || plan->format != CMI_FORMAT_COO || plan->dtype != CMI_F64)
        return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo_dot_plan_f64: null plan, or a plan made for another format or value type");
    if (plan->coo_csr_plan) return cmi_spmv_csr_dot_plan_f64(plan->coo_csr_plan, plan->coo_offsets, Aj, Ax, x, y, w, dot_dev, workspace, stream);
    const int st = cmi_spmv_coo_plan_f64(plan, Ai, Aj, Ax, x, y, 0, stream);
    return st ? st : cmi_blas_dot_f64(plan->rows, y, w, dot_dev, workspace, stream);
}
CMI_API int cmi_spmv_coo_dot_plan_f32(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const float *Ax, const float *x,
                                      float *y, const float *w, double *dot_dev, void *workspace, void *stream)
{
    if (!plan || plan->format != CMI_FORMAT_COO || plan->dtype != CMI_F32)
        return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_coo_dot_plan_f32: null plan, or a plan made for another format or value type");
    if (plan->coo_csr_plan) return cmi_spmv_csr_dot_plan_f32(plan->coo_csr_plan, plan->coo_offsets, Aj, Ax, x, y, w, dot_dev, workspace, stream);
    const int st = cmi_spmv_coo_plan_f32(plan, Ai, Aj, Ax, x, y, 0, stream);
    return st ? st : cmi_blas_dotd_f32(plan->rows, y, w, dot_dev, workspace, stream);
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

// HYB through a plan: one launch when the plan found the COO part sorted by row (hyb_tile_kernel: y written once, the host
// loops' bits), else the two launches above with the plan's launch shapes.
CMI_API int cmi_spmv_hyb_plan_f64(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                                  const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x,
                                  double *y, int accumulate, void *stream)
{
    return cmi::spmv_hyb_plan<double>(plan, CMI_F64, ell_pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, accumulate, stream);
}
CMI_API int cmi_spmv_hyb_plan_f32(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                                  const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x,
                                  float *y, int accumulate, void *stream)
{
    return cmi::spmv_hyb_plan<float>(plan, CMI_F32, ell_pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, accumulate, stream);
}

// y <- A x and *dot_dev <- <y, w> (a double) through a HYB plan: in the one-launch kernel's single pass where the plan runs it,
// else the multiply followed by the library's dot.
template <typename T>
static int hyb_dot_plan(const cmi_plan *plan, int dtype, int64_t pitch, const int *eAj, const T *eAx, const int *cAi, const int *cAj,
                        const T *cAx, const T *x, T *y, const T *w, double *dot_dev, void *workspace, void *stream)
{
    if (!plan || (!w && plan->rows > 0) || !dot_dev || !workspace) return cmi::fail(CMI_ERROR_INVALID_VALUE, "cmi_spmv_hyb_dot_plan: null plan, w, result or workspace");
    int partials = 0;
    const int st = cmi::spmv_hyb_plan<T>(plan, dtype, pitch, eAj, eAx, cAi, cAj, cAx, x, y, 0, stream, w, (double *)workspace, &partials);
    if (st) return st;
    if (partials > 0) return cmi::reduce_partials_f64(partials, (double *)workspace, dot_dev, cmi::as_stream(stream));
    if constexpr (std::is_same<T, double>::value) return cmi_blas_dot_f64(plan->rows, y, w, dot_dev, workspace, stream);
    else return cmi_blas_dotd_f32(plan->rows, y, w, dot_dev, workspace, stream);
}
CMI_API int cmi_spmv_hyb_dot_plan_f64(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                                      const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x,
                                      double *y, const double *w, double *dot_dev, void *workspace, void *stream)
{
    return hyb_dot_plan<double>(plan, CMI_F64, ell_pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, w, dot_dev, workspace, stream);
}
CMI_API int cmi_spmv_hyb_dot_plan_f32(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                                      const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x,
                                      float *y, const float *w, double *dot_dev, void *workspace, void *stream)
{
    return hyb_dot_plan<float>(plan, CMI_F32, ell_pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, w, dot_dev, workspace, stream);
}
