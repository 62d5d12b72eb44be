// common.h -- shared by the HIP translation units of libcusp_mi355x.so.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/cusp_mi355x.h"

#define CMI_API extern "C" __attribute__((visibility("default")))

namespace cmi {

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kXcds = 8;    // MI355X: 8 XCDs, blocks dealt round-robin over them
constexpr int kCus  = 256;

// thread-local last-error message
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

inline int fail(cmi_status st, const char *what)
{
    set_error("%s", what);
    return st;
}

inline int hip_fail(hipError_t e, const char *what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? CMI_ERROR_ALLOC : CMI_ERROR_HIP;
}

#define CMI_HIP(call)                                      \
    do {                                                   \
        hipError_t e__ = (call);                           \
        if (e__ != hipSuccess) return ::cmi::hip_fail(e__, #call); \
    } while (0)

// after a kernel launch: report launch-time errors (the reference never checks)
#define CMI_LAUNCH_CHECK(name)                             \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return ::cmi::hip_fail(e__, "launch " name); \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// tuning table lookup (tuning.hip)
void select_config(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                   const cmi_config *user, cmi_config *out);

void waver_rule(int dtype, cmi_waver_rule *out); // tuning.hip: csr_waver's shape and AUTO gates (the table's "waver_rule", else the defaults)
// row-length profile of a CSR matrix (spmv_csr.hip): longest row, entries sitting in rows of kLongRowMin or more
struct row_profile { int64_t max_len = -1, in_long = 0; }; // max_len < 0: not measured
int measure_row_lengths(int64_t rows, const int *Ap, hipStream_t s, int64_t *max_len, int64_t *entries_in_long_rows, int64_t *ends = nullptr); // ends: {Ap[0], Ap[rows]}
int measure_column_locality(int64_t rows, int64_t cols, const int *Ap, const int *Aj, int halfwin, hipStream_t s, int64_t *inside, int64_t *jumps); // spmv_csr.hip
bool prefers_balanced(int64_t rows, int64_t nnz, const row_profile &pr, size_t value_bytes, bool strict_order);
// are the row indices of a COO matrix non-decreasing and inside [0, rows)?  (spmv_coo_hyb.hip; synchronises the stream)
int coo_rows_sorted(int64_t rows, int64_t nnz, const int *Ai, hipStream_t s, int *sorted, int *long_runs = nullptr); // long_runs: a row of > 1024 entries
// ELL lanes per row (spmv_ell_dia.hip): 1 = the row kernel (storage-order sums), 2..16 = the slices kernel.  Auto rule,
// from tools/ell_wide_probe.py (archive/profiles/r02_ell_wide_lanes_f64.txt): every lane keeps >= kEllSliceMinSlots slots
// (lanes = width / 16 rounded down to a power of two, at most 16); two lanes only pay below kEllSliceMaxRows2 rows
// (one lane per row no longer fills 256 CUs x 8 waves x 64 lanes), four or more up to kEllSliceMaxRows.
constexpr int64_t kEllSliceMaxRows2 = 131072, kEllSliceMaxRows = 524288;
constexpr int kEllSliceMinSlots = 16;
int ell_lanes_per_row(const cmi_config &c, int64_t rows, int64_t width);

// Deterministic fold of `npartial` (<= kPartialCapacity) doubles at the start of a
// cmi_blas_workspace_bytes() buffer into *result (blas1.hip; fixed tree, no atomics).
constexpr int kPartialCapacity = 1 << 17; // (r4: 2^16 -> 2^17, so that wave tiles of 1024 entries per workgroup keep the fused dot up to 134 M entries)
constexpr int kFoldChunk = 1024;
constexpr int kFoldedMax = kPartialCapacity / kFoldChunk;
int reduce_partials_f64(int npartial, double *workspace, double *result, hipStream_t s);
// workspace layout: TWO fold areas of kFoldArea doubles (cmi_blas_workspace_bytes); the plain entry points use the first.
//   area (doubles): [0, kPartialCapacity) partials | kFoldedMax folded | the fold's ticket counter (+ pad) |
//                   kScalarCopies copies of the folded scalar, one 128-byte line each ("slots", fold-ahead only)
// Every kernel that leaves partials also resets the ticket and the slots (reset_fold_state), so the workspace needs no
// initialisation.
constexpr int kScalarCopies = 16, kScalarStride = 16; // doubles
constexpr int kFoldArea = kPartialCapacity + kFoldedMax + 2 + kScalarCopies * kScalarStride;
// "not folded yet": a quiet NaN with a payload no reduction produces
constexpr unsigned long long kPendingBits = 0x7FF8C0DEC0DE0000ull;
__device__ __forceinline__ unsigned int *ticket_of(double *area)
{
    return reinterpret_cast<unsigned int *>(area + kPartialCapacity + kFoldedMax);
}
__device__ __forceinline__ double *slots_of(double *area) { return area + kPartialCapacity + kFoldedMax + 2; }
// by ONE thread of the kernel that leaves partials in `area` (the consumer's fold starts from a clean ticket and "pending" slots;
// nobody reads either before that kernel has ended)
__device__ __forceinline__ void reset_fold_state(double *area)
{
    *ticket_of(area) = 0;
    double *s = slots_of(area);
#pragma unroll
    for (int c = 0; c < kScalarCopies; c++) reinterpret_cast<unsigned long long *>(s)[c * kScalarStride] = kPendingBits;
}

// ---- cross-lane steps on the DPP path (gfx9 family: row_shr, row_bcast:15/31, wave_shr:1) -----------------------------
// `__shfl_*` compiles to ds_bpermute_b32: a trip through the LDS crossbar per step, and a wave-wide reduction or scan is
// six DEPENDENT steps -- ~1500 cycles at the very end of a workgroup's life, i.e. residency: the fused <y, w> of
// csr_stream cost 10-15 us of 128 on the headline matrix with it (archive/tools/r2_probe.hip: csrx flags 5/9/17 against 1,
// archive/profiles/r02_probe_timing_session5.txt).  DPP moves ride on the VALU.  All deterministic: a fixed tree.
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118, kDppRowBcast15 = 0x142,
              kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138;
// value of the source lane selected by CTRL, `fill` where there is none (or the row is masked off)
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ int dpp_take(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ double dpp_take(double v, double fill)
{
    const int lo = dpp_take<CTRL, ROW_MASK>(__double2loint(v), __double2loint(fill));
    const int hi = dpp_take<CTRL, ROW_MASK>(__double2hiint(v), __double2hiint(fill));
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_take(float v, float fill)
{
    return __int_as_float(dpp_take<CTRL, ROW_MASK>(__float_as_int(v), __float_as_int(fill)));
}
// sum over every aligned group of G lanes (G a power of two, 2..64), left in the group's LAST lane; fixed tree
template <int G, typename T> __device__ __forceinline__ T group_sum_to_last(T v)
{
    if constexpr (G >= 2) v = v + dpp_take<kDppRowShr1>(v, T(0));
    if constexpr (G >= 4) v = v + dpp_take<kDppRowShr2>(v, T(0));
    if constexpr (G >= 8) v = v + dpp_take<kDppRowShr4>(v, T(0));
    if constexpr (G >= 16) v = v + dpp_take<kDppRowShr8>(v, T(0));
    if constexpr (G >= 32) v = v + dpp_take<kDppRowBcast15, 0xa>(v, T(0));
    if constexpr (G >= 64) v = v + dpp_take<kDppRowBcast31, 0xc>(v, T(0));
    return v;
}
// inclusive prefix sum over the 64 lanes of the wave
__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v += dpp_take<kDppRowShr1>(v, 0);
    v += dpp_take<kDppRowShr2>(v, 0);
    v += dpp_take<kDppRowShr4>(v, 0);
    v += dpp_take<kDppRowShr8>(v, 0);            // inclusive inside every row of 16 lanes
    v += dpp_take<kDppRowBcast15, 0xa>(v, 0);    // rows 1 and 3 add the total of the row before them
    v += dpp_take<kDppRowBcast31, 0xc>(v, 0);    // rows 2 and 3 add the total of rows 0-1
    return v;
}
// lane 63 receives the reduction of all 64 lanes (other lanes: partial results)
__device__ __forceinline__ double wave_sum_to_last(double v)
{
    v += dpp_take<kDppRowShr1>(v, 0.0);
    v += dpp_take<kDppRowShr2>(v, 0.0);
    v += dpp_take<kDppRowShr4>(v, 0.0);
    v += dpp_take<kDppRowShr8>(v, 0.0);
    v += dpp_take<kDppRowBcast15, 0xa>(v, 0.0);
    v += dpp_take<kDppRowBcast31, 0xc>(v, 0.0);
    return v;
}
__device__ __forceinline__ int wave_min_to_last(int v, int big)
{
    int t;
    t = dpp_take<kDppRowShr1>(v, big); v = t < v ? t : v;
    t = dpp_take<kDppRowShr2>(v, big); v = t < v ? t : v;
    t = dpp_take<kDppRowShr4>(v, big); v = t < v ? t : v;
    t = dpp_take<kDppRowShr8>(v, big); v = t < v ? t : v;
    t = dpp_take<kDppRowBcast15, 0xa>(v, big); v = t < v ? t : v;
    t = dpp_take<kDppRowBcast31, 0xc>(v, big); v = t < v ? t : v;
    return v;
}
// the left neighbour's value (lane - 1); lane 0 gets `fill`
__device__ __forceinline__ int wave_shift_up(int v, int fill) { return dpp_take<kDppWaveShr1>(v, fill); }

// One partial of a fused dot product per workgroup: lanes folded by a fixed DPP tree (wave_sum_to_last), waves in order
// (`slots`: one double per wave of the workgroup, in LDS).  Every thread of the workgroup must call it.
__device__ __forceinline__ void tile_dot_store(double d, double *slots, double *out)
{
    d = wave_sum_to_last(d);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == kWave - 1) slots[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)((blockDim.x + kWave - 1) / kWave); w++) s += slots[w];
        *out = s;
    }
}

// XCD-aware tile index.  Workgroups b and b+8 share an XCD (observed round-robin placement; a
// different placement changes speed only, never results).  mode 0: tile = b.  mode 1: every XCD walks
// one contiguous eighth of the tiles.  mode C >= 2: tiles are dealt to the XCDs in chunks of C
// consecutive tiles, so the x window a chunk gathers (its rows +- the matrix bandwidth) is fetched
// into ONE L2 while the whole grid still sweeps the arrays front to back.
__device__ __forceinline__ int64_t tile_of_block(int64_t b, int64_t tiles_per_xcd, int mode)
{
    if (mode == 0) return b;
    if (mode == 1) return (b % kXcds) * tiles_per_xcd + b / kXcds;
    const int64_t q = b / kXcds, r = b % kXcds, C = mode;
    return ((q / C) * kXcds + r) * C + q % C;
}

// grid of a tile kernel: one workgroup per tile, padded so that tile_of_block maps whole chunk rounds
inline int64_t padded_grid(int64_t tiles, int swizzle)
{
    if (swizzle <= 0) return tiles;
    if (swizzle == 1) return ceil_div(tiles, kXcds) * kXcds;
    return ceil_div(tiles, (int64_t)kXcds * swizzle) * kXcds * swizzle;
}

template <typename T> struct vec2;
template <> struct vec2<double> { typedef double __attribute__((ext_vector_type(2))) type; };
template <> struct vec2<float>  { typedef float  __attribute__((ext_vector_type(2))) type; };
typedef int __attribute__((ext_vector_type(4))) int4v;
typedef int __attribute__((ext_vector_type(2))) int2v;
typedef double __attribute__((ext_vector_type(2))) double2v;
typedef float __attribute__((ext_vector_type(4))) float4v;

// cache policy of a kernel = cmi_config.nontemporal: bit 0 -> once-read matrix streams are loaded
// with the nt hint, bit 1 -> y is stored with the nt hint (measured on MI355X: the y store is the
// expensive 10 % of CSR SpMV's bytes; nt stores cut it by ~40 %, tools/csr_ablate.hip).
constexpr int kPolLoadNT = 1, kPolStoreNT = 2;
constexpr int64_t kInfinityCacheBytes = 256ll << 20; // MALL, shared by the 8 XCDs
constexpr int kWaveTileMaxK = 10; // csr_wave: entries per lane (= the longest row of the matrix)
constexpr int kPolPairs = 8;   // csr_stream, f64, single-pass tile path: streams requested as (int2, double2) pairs -- every line requested once (round 3)
constexpr int kPolStrided = 4; // csr_stream only: entry streams requested lane-strided (a dword / a value per lane per instruction), not as 16-byte vectors
// $CMI_CSR_STRIDED=0/1 overrides the bit (measurements: A/B of the two request shapes through every tool and test)
inline int csr_lane_strided(int policy_bits)
{
    static const int env = [] { const char *e = std::getenv("CMI_CSR_STRIDED"); return e ? std::atoi(e) : -1; }();
    return env >= 0 ? (env != 0) : ((policy_bits & kPolStrided) != 0);
}

template <bool NT, typename V> __device__ __forceinline__ V ld(const V *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT, typename V> __device__ __forceinline__ void st(V *p, V v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// s + p[0] + p[stride] + ... + p[(n-1)*stride], added in that order (stride 1: the reference host loop's order),
// with the LDS reads of eight terms issued before the first add: a dependent read -> add chain costs ~40 ns per
// entry, batched reads ~8-12 ns, and the bits are the same.  (Also tried: reading the last 1..7 terms with clamped
// positions so that a five-entry row costs one LDS round trip -- 1.4 % SLOWER on the headline matrix, whose
// workgroups hide that latency anyway; the short tail stays a plain loop.)
template <typename T> __device__ __forceinline__ T sum_strided(T s, const T *p, int n, int stride)
{
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        const T *q = p + (size_t)j * stride;
        const T v0 = q[0], v1 = q[stride], v2 = q[2 * stride], v3 = q[3 * stride], v4 = q[4 * stride], v5 = q[5 * stride],
                v6 = q[6 * stride], v7 = q[7 * stride];
        s = s + v0; s = s + v1; s = s + v2; s = s + v3; s = s + v4; s = s + v5; s = s + v6; s = s + v7;
    }
    for (; j < n; j++) s = s + p[(size_t)j * stride];
    return s;
}
template <typename T> __device__ __forceinline__ T sum_in_order(T s, const T *p, int n)
{
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        const T v0 = p[j], v1 = p[j + 1], v2 = p[j + 2], v3 = p[j + 3], v4 = p[j + 4], v5 = p[j + 5], v6 = p[j + 6], v7 = p[j + 7];
        s = s + v0; s = s + v1; s = s + v2; s = s + v3; s = s + v4; s = s + v5; s = s + v6; s = s + v7;
    }
    for (; j < n; j++) s = s + p[j];
    return s;
}

// run `f(std::integral_constant<int, POL>)` for the runtime policy value 0..3
template <typename F> inline void with_policy(int pol, F f)
{
    switch (pol & 3) {
    case 0: f(std::integral_constant<int, 0>()); break;
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    default: f(std::integral_constant<int, 3>()); break;
    }
}

} // namespace cmi

// The plan object of include/cusp_mi355x.h (plan.hip): what the library learnt about one matrix.  Read-only after
// cmi_plan_create*; owns no device memory except a HYB plan's `hyb_tile_start` (one int per 256 rows) and the opt-in
// 16-bit column copy of a CMI_CSR_STREAM_C16 plan.
struct cmi_plan {
    int format, dtype;
    int64_t rows, cols, nnz;
    cmi_config cfg;        // resolved launch shape (kernel CMI_CSR_BALANCED when the profile switched kernels)
    bool cfg_explicit = false; // the caller named the kernel (its launch shape is then left as given)
    cmi::row_profile prof; // CSR
    int coo_sorted;        // COO, HYB's COO part: 1 / 0; -1 otherwise
    // HYB (cmi_plan_create_hyb): ELL width, COO entries, the COO part's launch shape for the two-launch path, and -- when the
    // COO part is sorted by row -- for every tile of kHybTileRows rows the first COO entry at or after its first row
    int64_t hyb_width = 0, hyb_coo = 0;
    cmi_config hyb_coo_cfg = {};
    int32_t *hyb_tile_start = nullptr; // device, tiles + 1 entries; null: two launches
    cmi_plan *hyb_coo_plan = nullptr;  // two launches: the COO half's own plan (sorted: row offsets + CSR kernels, accumulating)
    // COO whose entries are sorted by row (plan.hip): the row offsets the row indices imply (num_rows + 1 ints, built once) and
    // the CSR plan for them -- the multiply then runs the CSR kernels on (offsets, Aj, Ax) and never reads the row indices:
    // 12 nnz + 20 N bytes instead of 16 nnz + 16 N, storage-order sums, long rows handled as CSR handles them
    int32_t *coo_offsets = nullptr;
    cmi_plan *coo_csr_plan = nullptr;
    // CSR with cfg.kernel == CMI_CSR_STREAM_C16 (spmv_csr16.hip): per-tile smallest column and the 16-bit offsets from it
    int32_t *csr16_base = nullptr;  // device, one per tile of cfg.rows_per_block rows
    uint16_t *csr16_cols = nullptr; // device, nnz (+ padding) entries
    int32_t *wave_row_start = nullptr; // device, 2 (wave_tiles + 1) entries, (first row, first entry) per tile: CMI_CSR_STREAM_WAVE on IRREGULAR short rows -- wave tile t owns the rows whose first
                                       // entry lies in [t wave_q, (t + 1) wave_q) (plan.hip wave_partition); null: 64 rows per wave
    int64_t wave_tiles = 0;
    int wave_q = 0;
    // order-sensitive 64-bit checksums of the arrays the plan was made from (cmi_plan_validate): the index array (CSR row offsets, COO /
    // HYB-COO row indices) and -- when the plan owns data derived from them (the 16-bit copy) -- the CSR column indices
    uint64_t fp_index = 0, fp_columns = 0;
    bool has_fp_index = false, has_fp_columns = false;
    int csr16_wave_k = 0;           // > 0: the copy is tiled per WAVE (64 rows; cfg.rows_per_block == 64) and multiplied by the wave-tile kernel with this many entries per lane
    bool kernel_asked = false;      // the caller named a plan-only kernel (WAVE on a partition, WAVEV, WAVEX, WAVER, PACKED): its requirements are hard errors, not fall-backs
    // CMI_CSR_STREAM_WAVER / _PACKED (spmv_csr_runs.hip): the run-compressed column copy on wave tiles (wave_tiles, wave_q as above)
    int32_t *runs_start = nullptr;   // device, 4 (wave_tiles + 1) entries: {first row, first entry, first piece, packed offset / 16} per tile
    uint32_t *runs_pieces = nullptr; // device, runs_count (+ padding): (first column << 2) | (length - 1)
    int64_t runs_count = 0;
    int runs_cap = 4;                // entries per piece at most (3 where that costs no more pieces: no LDS bank conflict between pieces)
    unsigned char *runs_packed = nullptr; // device, _PACKED only: per tile [pieces | pad | values | pad]
    int64_t runs_packed_bytes = 0;
    unsigned char *csr16_packed = nullptr; // device, _PACKED on stencil-like rows (spmv_csr16.hip): fixed-stride wave tiles [head | row starts | 16-bit columns | values]
    int64_t csr16_packed_bytes = 0;
    uint64_t fp_values = 0;          // checksum of the values a _PACKED plan copied (cmi_plan_validate_values)
    bool has_fp_values = false;
};

namespace cmi {
constexpr int kHybTileRows = 256; // rows per workgroup of the one-launch HYB kernel = its block size
int hyb_tile_starts(int64_t rows, int64_t coo_entries, const int *coo_Ai, int32_t *tile_start, int *max_in_tile_dev, hipStream_t s);
// The one-launch kernel walks a tile's COO entries 256 at a time behind its ELL slots: right for a light COO part (the usual
// HYB: a few entries in some rows), wrong for a heavy one, where the entry-tiled COO kernel in a second launch is faster
// (tools/hyb_fuse_probe.py, archive/profiles/r02_hyb_one_vs_two_launches.txt: the crossover lies at 2.8-3.1 entries per row now that the second launch is the CSR kernel on the COO plan's row offsets).  One launch when the COO part averages at most
// kHybFusedMaxPerRow entries per row and no tile holds more than kHybFusedMaxInTile; $CMI_HYB_ONE_LAUNCH=0/1 forces it.
constexpr double kHybFusedMaxPerRow = 3.0;
constexpr int kHybFusedMaxInTile = 4096;
// spmv_csr16.hip: the plan's 16-bit column copy (built only if every tile qualifies) and the multiply that reads it
int wave_partition_build(cmi_plan *p, const int *Ap, int k, hipStream_t s, int q_override = 0); // spmv_csr.hip (q_override: entries per tile, csr_wavev)
int csr16_build(cmi_plan *p, const int *Ap, const int *Aj, hipStream_t s, int wave_k = 0);
// spmv_csr_runs.hip: the plan's run-compressed column copy (+ the packed tiles when `values` is given) and the multiply that reads it
int csr_runs_build(cmi_plan *p, const int *Ap, const int *Aj, int v, double min_mean_piece, const void *values, hipStream_t s, double *mean_piece, int cap = 0);
int csr_runs_multiply_f64(const cmi_plan *p, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, int accumulate, hipStream_t s,
                          const double *w, double *dot_partial, int *dot_partials, int cache_policy, int xcd_swizzle);
int csr_runs_multiply_f32(const cmi_plan *p, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, int accumulate, hipStream_t s,
                          const float *w, double *dot_partial, int *dot_partials, int cache_policy, int xcd_swizzle);
int csr16_pack(cmi_plan *p, const int *Ap, const void *values, hipStream_t s); // the wave-tiled 16-bit copy + the values -> packed wave tiles (CMI_CSR_STREAM_PACKED)
int csr16_multiply_f64(const cmi_plan *p, const int *Ap, const double *Ax, const double *x, double *y, int accumulate,
                       hipStream_t s, const double *w, double *dot_partial, int *dot_partials, int cache_policy, int xcd_swizzle);
int csr16_multiply_f32(const cmi_plan *p, const int *Ap, const float *Ax, const float *x, float *y, int accumulate,
                       hipStream_t s, const float *w, double *dot_partial, int *dot_partials, int cache_policy, int xcd_swizzle);
}
