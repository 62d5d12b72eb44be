// plan.hip -- cmi_plan: what a multiply needs to know about ONE matrix beyond its arrays, found once.
//
// SURVEY.md section 8(b) asked for `cmi_plan_create/destroy/select` owning "autotune table lookup, optional
// preprocessing"; round 1 instead measured a CSR matrix's row lengths inside its first multiply (hipMalloc + copy +
// stream synchronise, cached by pointer).  Here the measurement happens at plan creation, explicitly, and the
// multiply entry points (cmi_spmv_*_plan_*) neither allocate nor wait.  The reference keeps comparable state in
// function-local statics of its KTT path (cusp/system/cuda/ktt/csr_multiply.h:22-29,239-247: `row_starts`,
// `row_counter`, recomputed on the host for every call) -- there is no object to port.
//
//   CSR : launch shape from the table (or the caller's config), completed; the row-length profile (longest row, entries
//         in rows of 512+) decides between the row-tile kernel, its long-row instance and the merge-path kernel.
//   COO : "are the entries sorted by row?" -- sorted input gets the row offsets its row indices imply and runs the CSR
//         kernels on them (12 instead of 16 bytes per entry); anything else the order-agnostic atomics kernels.  An explicit
//         CMI_COO_TILE config keeps the COO tile kernel.
//   ELL / DIA : the resolved launch shape only (nothing to measure).
//   HYB (cmi_plan_create_hyb) : is the COO part sorted by row?  Then one int per 256 rows -- where that tile's COO entries
//         begin -- lets ONE kernel finish a row (ELL slots, then its COO entries) instead of two launches over y.
#include "common.h"
#include <cstdlib>
#include <new>

using namespace cmi;

// opt-in default for cmi_plan_create_csr with an AUTO kernel: try the 16-bit column copy ($CMI_COMPRESS_INDICES=1 at load,
// cmi_set_index_compression afterwards)
static int g_compress = -1;
static int compress_default()
{
    if (g_compress < 0) {
        const char *e = std::getenv("CMI_COMPRESS_INDICES");
        g_compress = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_compress;
}
CMI_API int cmi_set_index_compression(int on) { g_compress = on ? 1 : 0; return CMI_SUCCESS; }
CMI_API int cmi_get_index_compression(void) { return compress_default(); }

static int plan_create(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                       const int32_t *index_array, const int32_t *csr_columns, const cmi_config *cfg, void *stream, cmi_plan **plan_out, const void *csr_values = nullptr);

// ---- checksum of an index array (cmi_plan_validate) -------------------------------------------------------------------------
// sum over i of mix(i, a[i]) mod 2^64: position-dependent terms, combined by integer addition -- the same value whatever the
// launch shape or the order the workgroups finish in.  One pass over the array at streaming speed (10^7 row offsets: ~10 us).
namespace cmi {
__device__ __forceinline__ unsigned long long fp_mix(unsigned long long i, int v)
{
    unsigned long long h = ((unsigned long long)(unsigned int)v + 0x9E3779B97F4A7C15ull) * ((i << 1) | 1ull);
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}
__global__ void __launch_bounds__(256) fingerprint_kernel(int64_t n, const int *__restrict__ a, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long slots[256 / kWave];
    unsigned long long h = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) h += fp_mix((unsigned long long)i, a[i]);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) h += __shfl_down(h, o);
    if ((threadIdx.x & (kWave - 1)) == 0) slots[threadIdx.x / kWave] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; w++) h += slots[w];
        atomicAdd(out, h);
    }
}
// *fp <- checksum of a[0, n); synchronises the stream (set-up / validation call)
static int fingerprint(int64_t n, const int *a, hipStream_t s, uint64_t *fp)
{
    *fp = 0;
    if (n <= 0 || !a) return CMI_SUCCESS;
    unsigned long long *dev = nullptr, host = 0;
    CMI_HIP(hipMalloc((void **)&dev, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(host), s);
    if (e == hipSuccess) {
        int64_t blocks = ceil_div(n, 256 * 8);
        if (blocks > kCus * 8) blocks = kCus * 8;
        hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, n, a, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "plan fingerprint");
    *fp = (uint64_t)host;
    return CMI_SUCCESS;
}
// how many entries of the index array a plan of this format was made from
static int64_t index_length(const cmi_plan *p)
{
    switch (p->format) {
    case CMI_FORMAT_CSR: return p->rows > 0 ? p->rows + 1 : 0;
    case CMI_FORMAT_COO: return p->nnz;
    case CMI_FORMAT_HYB: return p->hyb_coo;
    default: return 0;
    }
}
} // namespace cmi

// Have the arrays this plan was made from changed?  Recomputes their checksum on the device (one streaming pass; synchronises
// `stream`) and compares it with the one taken at plan creation.  *valid_host = 1: same contents (up to a 2^-64 collision), 0: the
// structure was edited in place -- destroy the plan and make a new one.  column_indices: only compared when the plan owns data
// derived from them (CMI_CSR_STREAM_C16); may be NULL otherwise.  ELL / DIA plans hold nothing derived from the arrays: always valid.
CMI_API int cmi_plan_validate(const cmi_plan *plan, const int32_t *index_array, const int32_t *column_indices, void *stream, int *valid_host)
{
    if (!plan || !valid_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_validate: null plan or result");
    *valid_host = 1;
    hipStream_t s = as_stream(stream);
    if (plan->has_fp_index) {
        if (!index_array) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_validate: the plan was made from an index array; pass it");
        uint64_t fp = 0;
        if (int st = fingerprint(index_length(plan), index_array, s, &fp)) return st;
        if (fp != plan->fp_index) *valid_host = 0;
    }
    if (plan->has_fp_columns && *valid_host) {
        if (!column_indices) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_validate: the plan owns a copy of the column indices; pass them");
        uint64_t fp = 0;
        if (int st = fingerprint(plan->nnz, column_indices, s, &fp)) return st;
        if (fp != plan->fp_columns) *valid_host = 0;
    }
    return CMI_SUCCESS;
}

// csr_wave (CMI_CSR_STREAM_WAVE) instead of csr_stream: the longest row has 2..10 entries and the mean is within 7 % of it
// ($CMI_CSR_WAVE=0: never -- measurements of csr_stream on stencil matrices)
static int wave_env()
{
    static const int env = [] { const char *e = std::getenv("CMI_CSR_WAVE"); return e ? std::atoi(e) : 1; }();
    return env;
}
static bool wave_tiles_fit(int64_t rows, int64_t nnz, int64_t max_len)
{
    if (!wave_env() || rows <= 0 || max_len < 2 || max_len > kWaveTileMaxK) return false;
    return (double)nnz >= 0.93 * (double)max_len * (double)rows;
}
// Round 4 (the regret run, profiles/r04_auto_regret.txt, then the A/B of tools/stencil_tiles_probe.py, r04_stencil_tiles_ab.txt: one
// process, plans interleaved, five rounds that agree to 0.3 %): on f64 stencil rows of 5-7 entries csr_wavev with V = 1 -- wave tiles of
// 256 entries on a plan-built partition, the entry streams fetched as 16-byte vectors instead of csr_wave's lane-strided 4 + 8 bytes --
// takes 0.954 of csr_wave's time replayed (headline matrix 113.8 against 119.4 us = 0.878 against 0.837 of peak; 7-point 215^3 163.4
// against 171.2), 0.95-0.96 cold (125.9 against 130.9; 170.9 against 180.2) and 0.954 with the fused <y, w> of CG (123.1 against 129.0).
// Not so: 3 per row (replay 0.99, cold 1.05, fused dot 1.09), f32 (1.01), a cache-resident matrix (1000^2: 13.7 against 13.3 us), and
// rows of 9 (0.98 replayed, without the columns).  So: f64, the longest row 5..8, the streams beyond 1.25 x the Infinity Cache, and few
// enough tiles for the fused dot's partial list (one per 1024 entries) -- otherwise csr_wave as before.  $CMI_CSR_WAVE_VEC=0: never.
// f32 (sessions 22 / 24 / 25, three boxes): V = 1 gains nothing (0.99-1.01, fused dot 1.06), V = 2 -- tiles of 512 entries -- takes
// 0.971-0.981 of csr_wave's time replayed on the 5- and 7-point matrices, 0.96-1.00 cold, 0.98-1.00 with the fused dot: small, never a
// loss, taken.  Returns the index vectors per lane (0: csr_wave stays).
static int stencil_vector_tiles(int64_t nnz, const row_profile &prof, int dtype, size_t vbytes)
{
    static const int env = [] { const char *e = std::getenv("CMI_CSR_WAVE_VEC"); return e ? std::atoi(e) : 1; }();
    if (!env || prof.max_len < 5 || prof.max_len > 8 || prof.in_long > 0) return 0;
    if (nnz * (int64_t)(sizeof(int) + vbytes) <= kInfinityCacheBytes + kInfinityCacheBytes / 4) return 0;
    const int v = dtype == CMI_F64 ? 1 : 2;
    return nnz <= (int64_t)1024 * v * kPartialCapacity ? v : 0;
}
// ... and csr_wave on a plan-built partition (spmv_csr.hip wave_partition_build) for IRREGULAR short rows: K entries per lane with
// K = floor(mean + longest / 64), so that a wave tile of Q = 64 K - longest entries holds about Q / mean <= 64 rows; the longest row
// small enough for the tiles to fill 90 % of the wave's request slots.  OPT-IN: asked for per plan (cfg.kernel = CMI_CSR_STREAM_WAVE
// with rows_per_block < 0) -- measured against csr_stream it wins on FEM-like rows (thermal2-like 0.94 / 0.90 of its time, f64 / f32)
// and loses on large matrices with scattered columns (1.02-1.30x; archive/profiles/r02_wavep_ab.txt), so no plan selects it by itself
// ($CMI_CSR_WAVE=2: every AUTO plan whose rows qualify does -- measurements).
static int wave_partition_k(int64_t rows, int64_t nnz, const row_profile &prof, bool asked, int asked_k)
{
    if (!asked && (wave_env() < 2 || rows < 4096)) return 0;
    if (rows <= 0 || nnz <= 0 || prof.max_len < 1 || prof.in_long > 0) return 0;
    const double mean = (double)nnz / (double)rows;
    if (!asked && mean < 2.5) return 0;
    int k = asked_k > 0 ? asked_k : (int)std::floor(mean + (double)prof.max_len / 64.0);
    if (k < 2 && asked_k <= 0) k = 2;
    if (k < 2 || k > kWaveTileMaxK || (double)prof.max_len > 6.4 * k) return 0;
    return k;
}

// csr_wavev (CMI_CSR_STREAM_WAVEV): index vectors per lane V (a wave tile = 256 V slots), or 0 = not for this matrix.  Asked for (a plan
// made with that kernel): the caller's V or the rule's, refused when the longest row takes more than half of the tile.  AUTO plans:
// the rule below ($CMI_CSR_WAVEV=0: never; =1: whenever the rows qualify).
// AUTO plans take csr_wavev when ALL of (tools/wavev_ab.py, profiles/r03_wavev_ab.txt and r03_wavev_wavex_ab.txt: 24 matrices, f64 and f32,
// every variant bit-checked, interleaved rounds on one box):
//   * the index + value streams are beyond 0.75 x the Infinity Cache -- smaller matrices lose or tie (305 000 rows of 20: 1.13x; thermal2-like
//     at 104 MB even) because the partition's extra scalar hop is not hidden by anything there; at 240 MB (f32, 2..8 per row) it wins 0.79;
//   * fewer than 44 entries per row on average: 2-60 per row with columns anywhere in a +-2000..5000 band take 0.64-0.84 of csr_stream's
//     time (the wave keeps 16 gathers per lane in flight where csr_stream keeps 4-8: these matrices are gather-bound, 0.31-0.44 of peak),
//     27-point-like and nlpkkt120-like rows 0.95-0.97, columns scattered over the whole vector 0.98-1.00 (nothing helps those);
//     ldoor-like (45.6 per row) is 1.05x against the table's re-tuned csr_stream shape -- hence the bound;
//   * at least 4096 rows, no row of 512+ entries, the longest row at most half a wave tile (wavev_vectors below).
// Tile size: V = 4 (1024 request slots per wave) -- except f64 rows of fewer than 8 entries whose columns share x lines (see
// column_profile below: thermal2-like x6 110.0 us with V = 1 against 119.6), known only to plans made WITH the column indices.
// Stencil rows never get here (csr_wave is chosen first).  $CMI_CSR_WAVEV=0: never, =1: whenever the rows qualify.
static bool wavev_auto(int64_t rows, int64_t nnz, const row_profile &prof, int v, size_t vbytes)
{
    (void)prof;
    if (v != 4 || rows < 4096) return false;
    const double mean = (double)nnz / (double)rows;
    if (mean < 2.0 || mean >= 44.0) return false; // (2.0: until session 35 of round 4 2.5 -- rows of 1..4, mean 2.4999: 84.5 against 102.7 us, r04_auto_regret_set2_before.txt)
    return nnz * (int64_t)(sizeof(int) + vbytes) > kInfinityCacheBytes / 4 * 3;
}
// What a plan made with the column indices (cmi_plan_create_csr) adds: where the columns of a row lie.  `jumps` = share of entries 16+
// columns away from their predecessor in the row (each its own L1 lookup), `inside` = share within 1536 columns of the row's diagonal
// position (what an LDS x window around a workgroup's rows would serve).  Measured (r03_wavev_wavex_ab.txt):
//   jumps >= 0.6 and inside >= 0.25 (columns anywhere in a band): csr_wavex, the x window in LDS -- f32 0.71-0.92 of csr_wavev's time
//       (window 4096 from 15 entries per row, else 2048), f64 0.93-1.00 (window 2048); with jumps <= 0.31 (FEM blocks, stencils, sorted
//       meshes) the window only adds traffic: 1.3-1.8x SLOWER -- never selected there; inside = 0 (scattered): nothing helps.
//   f64, fewer than 8 entries per row, jumps < 0.5: V = 1.
//   inside < 0.05 with jumps >= 0.6 (scattered): csr_stream stays (wave tiles 0.97-1.06x: a wash).
//   f32 rows of 16+ entries with jumps < 0.5 -- or columns unknown -- : csr_stream stays (27-point-like / nlpkkt120-like f32: wave tiles 1.05-1.06x).
struct column_profile { double jumps = -1.0, inside = -1.0; }; // < 0: not measured
static int wavev_env()
{
    static const int env = [] { const char *e = std::getenv("CMI_CSR_WAVEV"); return e ? std::atoi(e) : -1; }();
    return env;
}
// Round 4: f64 rows of fewer than 8 entries whose columns share x lines take wave tiles with V = 1 BELOW the size gate
// too -- thermal2-like at 0.3 x / 1 x its size: 9.7 against 10.4 us, 20.8 against 21.8 (3 x: 63.6 against 70.6, already admitted),
// profiles/r04_size_gates_scale_sweep.txt, r04_thermal2_wave_tiles_v1.txt; V = 2 / 4 lose there.  From 200 000 rows; the column profile
// (jumps < 0.5) is checked by the caller, so only plans made WITH the columns qualify.
static bool short_f64_rows(int64_t rows, int64_t nnz, const row_profile &prof, int dtype, bool have_columns)
{
    if (!have_columns || dtype != CMI_F64 || rows < 200000 || nnz <= 0 || prof.max_len < 1 || prof.in_long > 0) return false;
    static const int env = [] { const char *e = std::getenv("CMI_CSR_WAVEV"); return e ? std::atoi(e) : -1; }();
    if (env == 0) return false;
    const double mean = (double)nnz / (double)rows;
    return mean >= 2.5 && mean < 8.0 && 2 * (prof.max_len + 3) <= 256;
}
// Round 4, session 35 (the regret table's sets 2 and 3, profiles/r04_auto_regret_set2_before.txt, r04_auto_regret_set3_size_gate.txt): on
// GATHER-BOUND band matrices (columns anywhere inside +-2000: every entry its own L1 lookup) the wave tiles win INSIDE the cache too --
// poisson(16) row lengths, 16 M entries: 47.0 against 53.5-56.3 us (f64), 29.0 against 33.7-40.5 (f32); 8 M: 26.5-28.1 against 29.0-30.6
// and 16.6-18.2 against 19.3-23.3; f32 at 4 M: 9.3-10.6 against 10.7-14.0; f64 at 4 M a tie -- where the size gate above (streams beyond
// 0.75 x the cache: made for rows whose columns share x lines, which lose or tie below it) keeps them out.  Such a matrix is admitted
// from 8 M entries (f64) / 4 M (f32) when its column profile says so (the caller measures it: plans made with the columns only); there
// the plain tiles do best, f64 with V = 2, f32 with V = 4 -- the x window pays only beyond the cache.
static bool band_candidate(int64_t rows, int64_t nnz, const row_profile &prof, int dtype, bool have_columns)
{
    if (!have_columns || rows < 4096 || nnz <= 0 || prof.max_len < 1 || prof.in_long > 0 || wavev_env() == 0) return false;
    const double mean = (double)nnz / (double)rows;
    const int v = dtype == CMI_F64 ? 2 : 4;
    return mean >= 2.0 && mean < 44.0 && nnz >= (dtype == CMI_F64 ? 8000000 : 4000000) && 2 * (prof.max_len + 3) <= 256 * v;
}
static int wavev_vectors(int64_t rows, int64_t nnz, const row_profile &prof, bool asked, int asked_v, size_t vbytes)
{
    if (rows <= 0 || nnz <= 0 || prof.max_len < 1 || prof.in_long > 0) return 0;
    const double mean = (double)nnz / (double)rows;
    int v = asked_v;
    if (v == 0) v = asked ? (mean >= 20.0 ? 4 : mean >= 8.0 ? 2 : 1) : 4; // (asked for: tile size by row length; AUTO plans: the measured V = 4)
    if (v != 1 && v != 2 && v != 4) return 0;
    while (v < 4 && !asked_v && 2 * (prof.max_len + 3) > 256 * v) v *= 2; // (the rule may widen the tile for a long row)
    if (2 * (prof.max_len + 3) > 256 * v) return 0;
    if (asked) return v;
    if (wavev_env() == 0) return 0;
    if (wavev_env() == 1) return rows >= 4096 ? v : 0;
    return wavev_auto(rows, nnz, prof, v, vbytes) ? v : 0;
}

// csr_waver (CMI_CSR_STREAM_WAVER): wave tiles on the run-compressed column copy.  Asked for: built whenever the tile can hold the longest
// row.  AUTO plans made with the columns ($CMI_CSR_WAVER=0: never, =1: whenever the rows qualify): at least 4096 rows of 8+ entries on
// average, no row of 512+, and the tuning table's "waver_rule" (tools/autotune_waver.py; tuning.hip has the built-in copy): at least
// min_entries entries (below: the partition's scalar hop is not hidden) and pieces of min_piece+ entries on average -- measured by
// building the copy, which is dropped again when they are shorter.  The rule also holds the launch shape (slots per lane, cap, XCD dealing).
static int waver_env()
{
    static const int env = [] { const char *e = std::getenv("CMI_CSR_WAVER"); return e ? std::atoi(e) : -1; }();
    return env;
}
static bool waver_try(cmi_plan *p, const int *Ap, const int *Aj, bool asked, int asked_v, int asked_cap, bool keep_policy, const void *values, hipStream_t s, int *st)
{
    const int64_t rows = p->rows, nnz = p->nnz;
    if (rows <= 0 || nnz <= 0 || p->prof.max_len < 1 || p->prof.in_long > 0 || p->cols < (p->dtype == CMI_F64 ? 2 : 4) || p->cols >= ((int64_t)1 << 30)) return false;
    // (f32, session 10 of round 4: ldoor-like 49.8 us against csr_stream's 70.9, nlpkkt120-like 101.3 against 156.8 -- the index stream is half of
    //  an f32 matrix's bytes, so compressing it pays more than for f64; same rule, profiles/r04_waver_f32_time.txt)
    cmi_waver_rule rule; // the persisted shape / gates (tuning table "waver_rule": tools/autotune_waver.py), else the built-in ones
    waver_rule(p->dtype, &rule);
    int v = asked_v ? asked_v : rule.items_per_thread;
    // (f64 rows of fewer than 12 entries: tiles of 512 slots, as for the plain wave tiles -- 5-point x 2 dof, 10 per row: 190.6 against
    //  211.3 us with V = 4; 9-point 143.7 against 144.4; in f32 V = 4 is ahead there, 125.1 against 133.1: r04_auto_regret_set4_before.txt)
    if (!asked && !asked_v && p->dtype == CMI_F64 && p->rows > 0 && (double)p->nnz < 12.0 * (double)p->rows && v == 4) v = 2; // (AUTO plans: an asked-for plan keeps the table's V)
    if (v != 1 && v != 2 && v != 4) return false;
    while (v < 4 && !asked_v && 2 * (p->prof.max_len + 3) > 256 * v) v *= 2;
    if (2 * (p->prof.max_len + 3) > 256 * v) return false;
    if (!asked) {
        if (waver_env() == 0) return false;
        const double mean = (double)nnz / (double)rows;
        if (rows < 4096) return false;
        // (size gate, profiles/r04_size_gates_scale_sweep.txt: at 10.8-10.9 M entries -- 130 MB of streams -- it takes 0.85-0.95 of the table
        //  kernel's time, at 5 M entries 1.13-1.22: smaller matrices keep csr_stream)
        //  f32 (r04_waver_f32_time.txt, f32 scale sweep): still 0.92 of the table kernel's time at 5.2 M entries -- gate at 5 M
        if (waver_env() != 1 && (mean < 8.0 || nnz < rule.min_entries)) return false;
    }
    double mean_piece = 0.0;
    *st = csr_runs_build(p, Ap, Aj, v, asked ? 0.0 : rule.min_piece, values, s, &mean_piece, asked && asked_cap ? asked_cap : rule.cap);
    if (*st != CMI_SUCCESS || !p->runs_start) return false;
    p->cfg.kernel = values ? CMI_CSR_STREAM_PACKED : CMI_CSR_STREAM_WAVER;
    p->cfg.block_size = 256;
    p->cfg.rows_per_block = 0;
    p->cfg.items_per_thread = v;
    p->cfg.threads_per_row = 0;
    p->cfg.nontemporal &= ~kPolStrided;
    p->cfg.xcd_swizzle = rule.xcd_swizzle; // (16: chunks of 16 workgroups = 64 wave tiles per XCD: 0.99 of launch order's time on both configs[3] matrices, twice, profiles/r04_waver_xcd_dealing.txt)
    if (!keep_policy) { // (a caller's policy bits are kept as given)
        if (nnz * (int64_t)(p->dtype == CMI_F64 ? 12 : 8) > kInfinityCacheBytes + kInfinityCacheBytes / 4) p->cfg.nontemporal |= kPolLoadNT;
        p->cfg.nontemporal |= kPolStoreNT;
    }
    return true;
}

// CMI_CSR_STREAM_PACKED on stencil-like rows: the wave-tiled 16-bit copy, then everything a wave needs packed into one span per tile
static bool packed16_try(cmi_plan *p, const int *Ap, const int *Aj, const void *values, bool keep_policy, hipStream_t s, int *st)
{
    const cmi_config before = p->cfg;
    *st = csr16_build(p, Ap, Aj, s, (int)p->prof.max_len); // granted only if every tile of 64 rows spans fewer than 65536 columns
    if (*st == CMI_SUCCESS && p->cfg.kernel == CMI_CSR_STREAM_C16 && p->csr16_wave_k > 0) *st = csr16_pack(p, Ap, values, s);
    if (*st != CMI_SUCCESS || !p->csr16_packed) { if (*st == CMI_SUCCESS) p->cfg = before; return false; }
    p->cfg.nontemporal &= ~kPolStrided;
    if (!keep_policy) {
        const int64_t vb = p->dtype == CMI_F64 ? 8 : 4;
        if (p->nnz * (2 + vb) > kInfinityCacheBytes + kInfinityCacheBytes / 4) p->cfg.nontemporal |= kPolLoadNT;
        p->cfg.nontemporal |= kPolStoreNT;
    }
    return true;
}

CMI_API int cmi_plan_create(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                            const int32_t *index_array, const cmi_config *cfg, void *stream, cmi_plan **plan_out)
{
    if (format == CMI_FORMAT_CSR && cfg && cfg->kernel == CMI_CSR_STREAM_C16)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_CSR_STREAM_C16 needs the column indices -- use cmi_plan_create_csr");
    return plan_create(format, dtype, num_rows, num_cols, num_entries, index_array, nullptr, cfg, stream, plan_out);
}

// CSR with both structure arrays: what cmi_plan_create(CMI_FORMAT_CSR, ...) does, plus -- asked for by
// cfg->kernel == CMI_CSR_STREAM_C16, or by the process-wide default with an AUTO kernel -- the 16-bit column copy.
CMI_API int cmi_plan_create_csr(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_offsets,
                                const int32_t *column_indices, const cmi_config *cfg, void *stream, cmi_plan **plan_out)
{
    if (num_entries > 0 && !column_indices) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_csr: null column indices");
    if (cfg && cfg->kernel == CMI_CSR_STREAM_PACKED) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_csr: CMI_CSR_STREAM_PACKED copies the values -- use cmi_plan_create_csr_values");
    return plan_create(CMI_FORMAT_CSR, dtype, num_rows, num_cols, num_entries, row_offsets, column_indices, cfg, stream, plan_out);
}

// COO with both index arrays: cmi_plan_create(CMI_FORMAT_COO, ...) plus, for ROW-SORTED entries, a CSR sub-plan made WITH the columns -- an FEM /
// KKT matrix held in COO then multiplies from the run-compressed column copy too (the reference's benchmark converts one matrix into every
// format, performance/spmv/spmv.cu:41-66: its COO line of such a matrix is this path).
CMI_API int cmi_plan_create_coo(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_indices,
                                const int32_t *column_indices, const cmi_config *cfg, void *stream, cmi_plan **plan_out)
{
    if (num_entries > 0 && !column_indices) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_coo: null column indices");
    return plan_create(CMI_FORMAT_COO, dtype, num_rows, num_cols, num_entries, row_indices, column_indices, cfg, stream, plan_out);
}

// CSR with the structure arrays AND the values: cmi_plan_create_csr plus -- asked for by cfg->kernel == CMI_CSR_STREAM_PACKED -- the packed
// per-tile copy of pieces and values (spmv_csr_runs.hip).  The values of any other plan stay the caller's.
CMI_API int cmi_plan_create_csr_values(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_offsets,
                                       const int32_t *column_indices, const void *values, const cmi_config *cfg, void *stream, cmi_plan **plan_out)
{
    if (num_entries > 0 && !column_indices) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_csr_values: null column indices");
    const bool packed = cfg && cfg->kernel == CMI_CSR_STREAM_PACKED;
    if (packed && num_entries > 0 && !values) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_csr_values: CMI_CSR_STREAM_PACKED needs the values");
    return plan_create(CMI_FORMAT_CSR, dtype, num_rows, num_cols, num_entries, row_offsets, column_indices, cfg, stream, plan_out, packed ? values : nullptr);
}

CMI_API int cmi_plan_validate_values(const cmi_plan *plan, const void *values, void *stream, int *valid_host)
{
    if (!plan || !valid_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_validate_values: null plan or result");
    *valid_host = 1;
    if (!plan->has_fp_values) return CMI_SUCCESS;
    if (!values) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_validate_values: the plan owns a copy of the values; pass them");
    uint64_t fp = 0;
    if (int st = fingerprint(plan->nnz * (plan->dtype == CMI_F64 ? 2 : 1), reinterpret_cast<const int *>(values), as_stream(stream), &fp)) return st;
    if (fp != plan->fp_values) *valid_host = 0;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_device_bytes(const cmi_plan *plan, int64_t *bytes)
{
    if (!plan || !bytes) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_device_bytes: null argument");
    int64_t b = 0;
    if (plan->hyb_tile_start) b += (ceil_div(plan->rows, kHybTileRows) + 1) * 4;
    if (plan->coo_offsets) b += (plan->rows + 1) * 4;
    if (plan->wave_row_start) b += (plan->wave_tiles + 1) * 8;
    if (plan->csr16_cols) b += (plan->nnz + 8) * 2;
    if (plan->csr16_base) b += ceil_div(plan->rows, plan->cfg.rows_per_block > 0 ? plan->cfg.rows_per_block : 1) * 4;
    if (plan->runs_start) b += (plan->wave_tiles + 1) * 16;
    if (plan->runs_pieces) b += (plan->runs_count + 64) * 4;
    if (plan->runs_packed) b += plan->runs_packed_bytes;
    if (plan->csr16_packed) b += plan->csr16_packed_bytes;
    int64_t sub = 0;
    if (plan->hyb_coo_plan && cmi_plan_device_bytes(plan->hyb_coo_plan, &sub) == CMI_SUCCESS) b += sub;
    if (plan->coo_csr_plan && cmi_plan_device_bytes(plan->coo_csr_plan, &sub) == CMI_SUCCESS) b += sub;
    *bytes = b;
    return CMI_SUCCESS;
}

static int plan_create(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                       const int32_t *index_array, const int32_t *csr_columns, const cmi_config *cfg, void *stream, cmi_plan **plan_out, const void *csr_values)
{
    if (!plan_out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: null result pointer");
    *plan_out = nullptr;
    if (format < 0 || format >= CMI_FORMAT_COUNT || dtype < 0 || dtype > 1)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: bad format or value type");
    if (num_rows < 0 || num_cols < 0 || num_entries < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: negative size");
    if (num_rows > INT32_MAX || num_cols > INT32_MAX) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: sizes exceed the int32 index type");
    const bool indexed = format == CMI_FORMAT_CSR || format == CMI_FORMAT_COO;
    if (indexed && !index_array && (format == CMI_FORMAT_CSR ? num_rows > 0 : num_entries > 0))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CSR / COO plans need the row offsets / row indices");
    cmi_plan *p = new (std::nothrow) cmi_plan;
    if (!p) return fail(CMI_ERROR_ALLOC, "cmi_plan_create: out of host memory");
    p->format = format;
    p->dtype = dtype;
    p->rows = num_rows;
    p->cols = num_cols;
    p->nnz = num_entries;
    p->prof = row_profile{};
    p->coo_sorted = -1;
    p->cfg_explicit = cfg && cfg->kernel != CMI_KERNEL_AUTO && !(cfg->kernel == CMI_CSR_STREAM_C16 && !(cfg->block_size || cfg->rows_per_block || cfg->items_per_thread));
    // CMI_CSR_STREAM_C16 is csr_stream's shape (the caller's launch-shape fields if any, else the table's) + the 16-bit copy
    cmi_config shape;
    bool want16 = false, table_shape = !cfg || cfg->kernel == CMI_KERNEL_AUTO;
    if (format == CMI_FORMAT_CSR && csr_columns) {
        if (cfg && cfg->kernel == CMI_CSR_STREAM_C16) {
            want16 = true;
            shape = *cfg;
            table_shape = !(cfg->block_size || cfg->rows_per_block || cfg->items_per_thread);
            shape.kernel = table_shape ? CMI_KERNEL_AUTO : CMI_CSR_STREAM;
            cfg = &shape;
        } else if (table_shape && compress_default())
            want16 = true;
    }
    // CMI_CSR_STREAM_WAVE with rows_per_block < 0: wave tiles on a partition the plan builds (irregular short rows); launch policy from
    // the table, entries per lane from the caller (items_per_thread) or the rule
    cmi_config part_shape;
    bool want_partition = false;
    int partition_k = 0;
    if (format == CMI_FORMAT_CSR && cfg && cfg->kernel == CMI_CSR_STREAM_WAVE && cfg->rows_per_block < 0) {
        want_partition = true;
        partition_k = cfg->items_per_thread;
        part_shape = *cfg;
        part_shape.kernel = CMI_KERNEL_AUTO;
        part_shape.rows_per_block = 0;
        part_shape.items_per_thread = 0;
        part_shape.block_size = 0;
        cfg = &part_shape;
        p->cfg_explicit = false;
    }
    // CMI_CSR_STREAM_WAVEV: wave-private tiles with the 16-byte-vector body on a partition the plan builds (rows of ~16-250 entries);
    // cache policy / XCD dealing from the caller's fields if set, else the table's; index vectors per lane from the caller or the rule
    cmi_config wavev_shape;
    bool want_wavev = false, want_wavex = false;
    int wavev_v = 0, wavex_window = 0;
    if (format == CMI_FORMAT_CSR && cfg && (cfg->kernel == CMI_CSR_STREAM_WAVEV || cfg->kernel == CMI_CSR_STREAM_WAVEX)) {
        want_wavev = true;
        want_wavex = cfg->kernel == CMI_CSR_STREAM_WAVEX; // the same partition; the multiply adds an x window in LDS (rows_per_block = its length)
        wavex_window = cfg->rows_per_block > 0 ? cfg->rows_per_block : 0;
        wavev_v = cfg->items_per_thread;
        if (want_wavex && wavev_v == 0) wavev_v = 4;
        if (want_wavex && wavev_v == 1) wavev_v = 2;
        wavev_shape = *cfg;
        wavev_shape.kernel = CMI_KERNEL_AUTO;
        wavev_shape.rows_per_block = 0;
        wavev_shape.items_per_thread = 0;
        wavev_shape.block_size = 0;
        wavev_shape.threads_per_row = 0;
        cfg = &wavev_shape;
        p->cfg_explicit = false;
    }
    // CMI_CSR_STREAM_WAVER / _PACKED: wave tiles on the run-compressed column copy (spmv_csr_runs.hip); policy / dealing as for WAVEV
    cmi_config waver_shape;
    bool want_waver = false, want_packed = false;
    int waver_v = 0, waver_cap = 0;
    if (format == CMI_FORMAT_CSR && cfg && (cfg->kernel == CMI_CSR_STREAM_WAVER || cfg->kernel == CMI_CSR_STREAM_PACKED)) {
        if (!csr_columns && num_entries > 0) { delete p; return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_CSR_STREAM_WAVER needs the column indices -- use cmi_plan_create_csr"); }
        want_waver = true;
        want_packed = cfg->kernel == CMI_CSR_STREAM_PACKED;
        waver_v = cfg->items_per_thread;
        waver_cap = cfg->threads_per_row; // (this kernel's use of the field: entries per piece at most, 0 = the rule, 3 or 4)
        waver_shape = *cfg;
        waver_shape.kernel = CMI_KERNEL_AUTO;
        waver_shape.rows_per_block = 0;
        waver_shape.items_per_thread = 0;
        waver_shape.block_size = 0;
        waver_shape.threads_per_row = 0;
        cfg = &waver_shape;
        p->cfg_explicit = false;
    }
    p->kernel_asked = want_partition || want_wavev || want_waver;
    // HYB's table key is its ELL part's (the COO part looks its own shape up per call)
    select_config(format == CMI_FORMAT_HYB ? CMI_FORMAT_ELL : format, dtype, num_rows, num_cols, num_entries, cfg, &p->cfg);
    hipStream_t s = as_stream(stream);
    const size_t vbytes = dtype == CMI_F64 ? 8 : 4;
    int st = CMI_SUCCESS;
    if (format == CMI_FORMAT_CSR && num_rows > 0 && num_entries > 0) {
        int64_t ends[2] = {0, num_entries};
        st = measure_row_lengths(num_rows, index_array, s, &p->prof.max_len, &p->prof.in_long, ends);
        // the kernels and the plan-owned arrays (wave partition, 16-bit copy) are sized from num_entries: row offsets that do not span
        // exactly [0, num_entries] are refused here, once, instead of being trusted by every multiply
        if (st == CMI_SUCCESS && (ends[0] != 0 || ends[1] != num_entries)) {
            set_error("cmi_plan_create: row offsets run from %lld to %lld, expected 0 to num_entries = %lld", (long long)ends[0], (long long)ends[1], (long long)num_entries);
            st = CMI_ERROR_INVALID_VALUE;
        }
        const bool auto_kernel = !cfg || cfg->kernel == CMI_KERNEL_AUTO;
        if (st == CMI_SUCCESS && auto_kernel && prefers_balanced(num_rows, num_entries, p->prof, vbytes, p->cfg.threads_per_row == 1)) {
            p->cfg.kernel = CMI_CSR_BALANCED;
            p->cfg.items_per_thread = 0; // the table's row-tile launch shape does not apply: balanced defaults
            p->cfg.blocks_per_cu = 0;
            p->cfg.xcd_swizzle = 0;      // (chunks in launch order: a skewed matrix has no x window worth dealing for)
        }
        if (st == CMI_SUCCESS && want16 && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1) {
            // the tiling is frozen into the copy, so the fused-dot instance cannot re-tile as spmv_csr.hip does for it: a
            // table-chosen shape takes whole waves of rows per tile here, for every multiply, when one LDS pass holds them
            const int tuned_rpb = p->cfg.rows_per_block;
            if (table_shape) {
                const int rpb = p->cfg.rows_per_block, up = (rpb + kWave - 1) / kWave * kWave;
                const double mean = (double)num_entries / (double)num_rows;
                if (up != rpb && up <= p->cfg.block_size && (double)up * mean + 3.0 <= (double)p->cfg.block_size * p->cfg.items_per_thread * 4) p->cfg.rows_per_block = up;
            }
            // (stencil-like rows and the table's shape: the copy is tiled for the wave-tile kernel instead -- 64 rows per tile)
            const int wave_k = table_shape && wave_tiles_fit(num_rows, num_entries, p->prof.max_len) ? (int)p->prof.max_len : 0;
            st = csr16_build(p, index_array, csr_columns, s, wave_k); // all tiles qualify -> cfg.kernel = CMI_CSR_STREAM_C16, else unchanged
            if (p->cfg.kernel != CMI_CSR_STREAM_C16) p->cfg.rows_per_block = tuned_rpb; // not granted: csr_stream as tuned
        }
        // Rows that all have (nearly) the same short length -- stencils -- take csr_stream's lane-strided body with wave-private
        // tiles (spmv_csr.hip csr_wave_kernel): 64 rows per wave, as many entries per lane as the longest row has, so every
        // tile fits and (mean within 7 % of the longest row) at least 93 % of the request lanes carry an entry.  Cache policy and
        // XCD dealing are the table's csr_stream entry's.  Not for a caller's explicit kernel, not over a granted 16-bit copy.
        // Equal short rows are a stencil's signature -- but only its columns make it one.  A plan made WITH the columns looks: where 75 %+ of
        // the entries sit 16+ columns from their predecessor in the row (a 5-point stencil: 40 %, 7-point: 57 %, 9-point: 33 %; columns drawn
        // anywhere inside a band: 97 %) the matrix is gather-bound and none of the stencil kernels below is right for it -- 6 / 9 entries per
        // row exactly, columns anywhere in +-2000: csr_wave / V = 1 tiles 161-167 us (f64), 94-95 (f32), where the general rule further down
        // (wave tiles V = 4, with the x window) gets 120-138 and 70-85 (profiles/r04_auto_regret_equal_lengths.txt).  Without the columns
        // the lengths decide, as before.
        bool gather_bound = false;
        if (st == CMI_SUCCESS && auto_kernel && !want_partition && !want_wavev && !want_waver && csr_columns && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
            wave_tiles_fit(num_rows, num_entries, p->prof.max_len)) {
            int64_t inside = 0, jumps = 0;
            st = measure_column_locality(num_rows, num_cols, index_array, csr_columns, 1536, s, &inside, &jumps);
            gather_bound = st == CMI_SUCCESS && (double)jumps >= 0.75 * (double)num_entries;
        }
        if (st == CMI_SUCCESS && !gather_bound && auto_kernel && !want_partition && !want_wavev && !want_waver && csr_columns && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
            wave_tiles_fit(num_rows, num_entries, p->prof.max_len) && waver_try(p, index_array, csr_columns, false, 0, 0, false, nullptr, s, &st)) {
            // stencil rows of 8+ entries whose columns come in runs (9-point: three runs of 3) and a plan made with the columns: the
            // run-compressed copy, tried BEFORE csr_wave (waver_try's own gates: the table's rule; pieces shorter than it asks -> nothing
            // is kept and csr_wave below runs).  9-point 3000^2: 143.9 us against csr_wave's 187.1 (f64), 89.8 against 134.0 (f32),
            // profiles/r04_auto_regret.txt -- until then every stencil-like matrix took csr_wave unseen.
        } else if (st == CMI_SUCCESS && !gather_bound && auto_kernel && !want_partition && !want_wavev && !want_waver && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
                   wave_tiles_fit(num_rows, num_entries, p->prof.max_len) && stencil_vector_tiles(num_entries, p->prof, dtype, vbytes) > 0) {
            // stencil rows of 5..8 entries beyond the Infinity Cache: wave tiles of 256 (f64) / 512 (f32) entries on a plan-built partition
            // (8 bytes per tile), entries fetched as 16-byte vectors (csr_wavev, V = 1 / 2) -- see stencil_vector_tiles above
            const int sv = stencil_vector_tiles(num_entries, p->prof, dtype, vbytes);
            st = wave_partition_build(p, index_array, sv, s, 256 * sv - (int)p->prof.max_len - 3);
            if (st == CMI_SUCCESS && p->wave_row_start) {
                p->cfg.kernel = CMI_CSR_STREAM_WAVEV;
                p->cfg.block_size = 256;
                p->cfg.rows_per_block = 0;
                p->cfg.items_per_thread = sv;
                p->cfg.threads_per_row = 0;
                p->cfg.nontemporal &= ~kPolStrided;
                p->cfg.nontemporal |= kPolLoadNT | kPolStoreNT; // (beyond the cache by the rule: every line of the streams is requested once)
            }
        } else if (st == CMI_SUCCESS && !gather_bound && auto_kernel && !want_wavev && !want_waver && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
            wave_tiles_fit(num_rows, num_entries, p->prof.max_len)) {
            p->cfg.kernel = CMI_CSR_STREAM_WAVE;
            p->cfg.block_size = 256;
            p->cfg.rows_per_block = 256;
            p->cfg.items_per_thread = (int)p->prof.max_len;
            p->cfg.threads_per_row = 0;
            p->cfg.nontemporal &= ~kPolStrided; // (the request shape is the kernel's own)
            // every line of its streams is requested exactly once: beyond the Infinity Cache the streaming hint is right whatever
            // the table's csr_stream entry of this bucket says (its 16-byte-vector shapes measured the other way) -- the top twelve of
            // 96 swept shapes on the headline matrix all carry it, a tridiagonal matrix of 10^7 rows 91.8 -> 83 us
            // (archive/profiles/r02_wave_shape_sweep.txt, r02_wave_ab.txt); below it, plain loads keep the matrix resident (tuning.hip)
            if (num_entries * (int64_t)(sizeof(int) + vbytes) > kInfinityCacheBytes + kInfinityCacheBytes / 4) p->cfg.nontemporal |= kPolLoadNT;
            p->cfg.nontemporal |= kPolStoreNT;
        } else if (st == CMI_SUCCESS && want_packed && csr_values && csr_columns && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
                   wave_tiles_fit(num_rows, num_entries, p->prof.max_len) && packed16_try(p, index_array, csr_columns, csr_values, waver_shape.nontemporal != 0, s, &st)) {
            // (stencil-like rows: packed wave tiles of the 16-bit copy, spmv_csr16.hip; p->cfg is set)
        } else if (st == CMI_SUCCESS && auto_kernel && !want_partition && !want_wavev && csr_columns && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
                   waver_try(p, index_array, csr_columns, want_waver, waver_v, waver_cap, want_waver && waver_shape.nontemporal != 0, want_packed ? csr_values : nullptr, s, &st)) {
            // (the run-compressed copy was built and pays: p->cfg is set; a caller's XCD dealing is kept as given)
            if (want_waver && waver_shape.xcd_swizzle != 0) p->cfg.xcd_swizzle = waver_shape.xcd_swizzle < 0 ? 0 : waver_shape.xcd_swizzle;
        } else if (st == CMI_SUCCESS && want_waver) {
            st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_CSR_STREAM_WAVER / _PACKED need f64 values, 2 <= columns < 2^30, items_per_thread 0, 1, 2 or 4, no row of 512+ entries and the longest row at most half of the 256 x items_per_thread slots of a wave tile");
        } else if (st == CMI_SUCCESS && auto_kernel && !want_partition && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1 &&
                   (wavev_vectors(num_rows, num_entries, p->prof, want_wavev, wavev_v, vbytes) > 0 ||
                    short_f64_rows(num_rows, num_entries, p->prof, dtype, csr_columns != nullptr && !want_wavev) ||
                    band_candidate(num_rows, num_entries, p->prof, dtype, csr_columns != nullptr && !want_wavev))) { // (a caller who asked for csr_wave on a partition gets that)
            int v = wavev_vectors(num_rows, num_entries, p->prof, want_wavev, wavev_v, vbytes);
            // v == 0 here: only the rules of round 4 for CACHE-RESIDENT matrices admitted this one -- f64 rows of fewer than 8 entries whose
            // columns share x lines run wave tiles with V = 1, gather-bound band matrices V = 2 (f64) / 4 (f32); else nothing changes
            const bool small_only = v == 0;
            const bool short_rule = small_only && short_f64_rows(num_rows, num_entries, p->prof, dtype, csr_columns != nullptr && !want_wavev);
            const bool band_rule = small_only && band_candidate(num_rows, num_entries, p->prof, dtype, csr_columns != nullptr && !want_wavev);
            if (small_only) v = 1;
            bool auto_wavex = false, keep_stream = false;
            if (!want_wavev) { // an AUTO plan: refine by the value type and -- made with the columns -- by where the columns lie
                const double mean = (double)num_entries / (double)num_rows;
                column_profile cp;
                if (csr_columns) {
                    int64_t inside = 0, jumps = 0;
                    st = measure_column_locality(num_rows, num_cols, index_array, csr_columns, 1536, s, &inside, &jumps);
                    if (st == CMI_SUCCESS) { cp.inside = (double)inside / (double)num_entries; cp.jumps = (double)jumps / (double)num_entries; }
                }
                const char *wx = std::getenv("CMI_CSR_WAVEX");
                if (small_only) {
                    if (band_rule && cp.jumps >= 0.6 && cp.inside >= 0.25) v = dtype == CMI_F64 ? 2 : 4; // (plain tiles: no window inside the cache)
                    else if (!(short_rule && cp.jumps >= 0.0 && cp.jumps < 0.5)) keep_stream = true;
                } else if (cp.jumps >= 0.6 && cp.inside >= 0.25 && !(wx && wx[0] == '0')) {
                    auto_wavex = true;
                    wavex_window = (dtype == CMI_F32 && mean >= 15.0) ? 4096 : 2048;
                } else if (cp.inside >= 0.0 && cp.inside < 0.05 && cp.jumps >= 0.6) {
                    keep_stream = true; // columns scattered over the whole vector: 0.08-0.11 of peak whatever runs; wave tiles 0.97-1.06x
                } else if (dtype == CMI_F32 && mean >= 16.0 && cp.jumps < 0.5) {
                    keep_stream = true; // f32 stencil / FEM-block rows of 16+ entries (or unknown columns): the table's csr_stream is 5-6 % faster
                } else if (mean < 12.0 && cp.jumps >= 0.0 && cp.jumps < (dtype == CMI_F64 ? 0.6 : 0.5)) {
                    // (f64 up to 0.6 since session 38: a 7-point stencil with 12 % of its entries removed -- rows of 1..7, 57 % jumps -- 118.3 us
                    //  with V = 1 against 127.0 with V = 4; in f32 V = 4 stays ahead there, 88.5 against 90.2 / 92.5)
                    // short rows whose columns share x lines: smaller tiles.  f64 rows of fewer than 8 entries, none longer than 16: V = 1
                    // (thermal2-like 20.4 / 63.4 us against 21.5 / 67.3 with V = 2); otherwise V = 2 -- against V = 4 (what every AUTO plan
                    // ran until the regret run of round 4, profiles/r04_auto_regret.txt): uniform 1..16 per row 74.9 against 78.3 us (f64) and
                    // 54.2 against 55.5 (f32), 4 / 40 per row 71.5 against 77.5 (V = 1: 78.6) and 44.9 against 46.3, thermal2-like x3 in f32
                    // 38.5 against 40.0.  Columns anywhere in a band (jumps >= 0.6) without the x window keep V = 4 (217.5 against 234.0).
                    // Either way only while the longest row still takes at most half of the smaller tile: with Q = 256 v - longest - 3
                    // entries per tile a few rows of 126-252 entries would leave Q = 127..1, i.e. up to one wave tile (8 bytes of plan
                    // memory, one wave) per handful of entries.  Widen back until the bound wavev_vectors() checked holds.
                    v = (dtype == CMI_F64 && mean < 8.0 && p->prof.max_len <= 16) ? 1 : 2;
                    while (v < 4 && 2 * (p->prof.max_len + 3) > 256 * v) v *= 2;
                }
            }
            if (keep_stream) {
                // (the table's csr_stream entry stays: nothing is built)
            } else if (st == CMI_SUCCESS) st = wave_partition_build(p, index_array, v, s, 256 * v - (int)p->prof.max_len - 3);
            if (st == CMI_SUCCESS && p->wave_row_start) {
                p->cfg.kernel = (want_wavex || auto_wavex) ? CMI_CSR_STREAM_WAVEX : CMI_CSR_STREAM_WAVEV;
                p->cfg.block_size = 256;
                p->cfg.rows_per_block = (want_wavex || auto_wavex) ? wavex_window : 0; // (the partition's; WAVEX: the window length, 0 = the default)
                p->cfg.items_per_thread = v;
                p->cfg.threads_per_row = 0;
                p->cfg.nontemporal &= ~kPolStrided;
                if (!(want_wavev && wavev_shape.nontemporal)) { // (a caller's policy bits are kept as given)
                    if (num_entries * (int64_t)(sizeof(int) + vbytes) > kInfinityCacheBytes + kInfinityCacheBytes / 4) p->cfg.nontemporal |= kPolLoadNT;
                    p->cfg.nontemporal |= kPolStoreNT;
                }
                // ... and so is a caller's XCD dealing (< 0: launch order) -- select_config takes only block size and policy from an AUTO-kernel
                // config, so until session 29 of round 4 an asked-for dealing was silently the table's (the "sweeps" of it measured nothing)
                if (want_wavev && wavev_shape.xcd_swizzle != 0) p->cfg.xcd_swizzle = wavev_shape.xcd_swizzle < 0 ? 0 : wavev_shape.xcd_swizzle;
            }
        } else if (st == CMI_SUCCESS && want_wavev) {
            st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_CSR_STREAM_WAVEV needs items_per_thread 0, 1, 2 or 4, no row of 512+ entries and the longest row at most half of the 256 x items_per_thread slots of a wave tile");
        } else if (st == CMI_SUCCESS && auto_kernel && p->cfg.kernel == CMI_CSR_STREAM && p->cfg.threads_per_row <= 1) {
            const int k = wave_partition_k(num_rows, num_entries, p->prof, want_partition, partition_k);
            if (want_partition && k == 0)
                st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: wave tiles on a row partition need 2..10 entries per lane and the longest row <= 6.4 x that");
            if (k > 0) st = wave_partition_build(p, index_array, k, s);
            if (st == CMI_SUCCESS && p->wave_row_start) {
                p->cfg.kernel = CMI_CSR_STREAM_WAVE;
                p->cfg.block_size = 256;
                p->cfg.rows_per_block = 0; // (the partition's: about Q / mean rows per wave)
                p->cfg.items_per_thread = k;
                p->cfg.threads_per_row = 0;
                p->cfg.nontemporal &= ~kPolStrided;
                if (!(want_partition && part_shape.nontemporal)) { // (a caller's policy bits are kept as given)
                    if (num_entries * (int64_t)(sizeof(int) + vbytes) > kInfinityCacheBytes + kInfinityCacheBytes / 4) p->cfg.nontemporal |= kPolLoadNT;
                    p->cfg.nontemporal |= kPolStoreNT;
                }
            }
        }
    } else if (format == CMI_FORMAT_COO) {
        int sorted = 1, long_runs = 0;
        if (num_entries > 0) st = coo_rows_sorted(num_rows, num_entries, index_array, s, &sorted, &long_runs);
        if (st == CMI_SUCCESS) {
            p->coo_sorted = sorted;
            const bool auto_kernel = !cfg || cfg->kernel == CMI_KERNEL_AUTO;
            // sorted entries: the table's CMI_TABLE_COO_SORTED key (the tile kernel with its tuned cache policy / XCD dealing)
            // (fewer than four entries: the tile kernel's vector loads have nothing to read -- the order-agnostic kernel stays)
            // (a row of more than 1024 entries: the tile kernel would walk its tail serially -- the order-agnostic kernel stays)
            if (sorted && !long_runs && auto_kernel && num_entries >= 4) select_config(CMI_TABLE_COO_SORTED, dtype, num_rows, num_cols, num_entries, cfg, &p->cfg);
            // Sorted entries and no kernel asked for: the row indices carry no more than row offsets do.  The plan builds the
            // offsets once (4 bytes per row of HBM it owns) and a CSR plan for them; every multiply then streams 12 bytes per
            // entry instead of 16 through the CSR kernels -- same products, storage-order sums, rows of any length.
            // ($CMI_COO_PLAN_OFFSETS=0: keep the COO kernels -- measurements of the tile kernel.)
            const char *off = std::getenv("CMI_COO_PLAN_OFFSETS");
            if (sorted && auto_kernel && num_entries >= 4 && num_rows > 0 && num_entries <= INT32_MAX - 65536 && !(off && off[0] == '0')) {
                hipError_t e = hipMalloc((void **)&p->coo_offsets, (size_t)(num_rows + 1) * sizeof(int32_t));
                if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create: COO row offsets");
                int sorted2 = 0;
                if (st == CMI_SUCCESS) st = cmi_coo_row_offsets(num_rows, num_entries, index_array, p->coo_offsets, &sorted2, stream);
                if (st == CMI_SUCCESS && sorted2)
                    st = plan_create(CMI_FORMAT_CSR, dtype, num_rows, num_cols, num_entries, p->coo_offsets, csr_columns /* cmi_plan_create_coo: the CSR plan may then hold the run-compressed copy */, nullptr, stream, &p->coo_csr_plan);
                if (st == CMI_SUCCESS && p->coo_csr_plan) p->cfg = p->coo_csr_plan->cfg; // what cmi_plan_config reports: the kernel that runs
            }
            // an explicit CMI_COO_TILE on unsorted entries would add rows up wrongly: refuse it here, where it is known
            if (!sorted && p->cfg.kernel == CMI_COO_TILE) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_COO_TILE needs row-sorted entries");
        }
    }
    if (st == CMI_SUCCESS && indexed && index_array && index_length(p) > 0) {
        st = fingerprint(index_length(p), index_array, s, &p->fp_index);
        p->has_fp_index = st == CMI_SUCCESS;
    }
    if (st == CMI_SUCCESS && (p->runs_packed || p->csr16_packed) && csr_values) {
        st = fingerprint(num_entries * (dtype == CMI_F64 ? 2 : 1), reinterpret_cast<const int *>(csr_values), s, &p->fp_values);
        p->has_fp_values = st == CMI_SUCCESS;
    }
    if (st == CMI_SUCCESS && p->coo_csr_plan && p->coo_csr_plan->has_fp_columns) { // (a COO plan made with the columns: its CSR sub-plan's copy is what cmi_plan_validate guards)
        p->fp_columns = p->coo_csr_plan->fp_columns;
        p->has_fp_columns = true;
    }
    if (st == CMI_SUCCESS && (p->csr16_cols || p->runs_pieces || p->csr16_packed) && csr_columns) {
        st = fingerprint(num_entries, csr_columns, s, &p->fp_columns);
        p->has_fp_columns = st == CMI_SUCCESS;
    }
    if (st != CMI_SUCCESS) { (void)cmi_plan_destroy(p); return st; }
    *plan_out = p;
    return CMI_SUCCESS;
}

// HYB: the ELL part's launch shape, the COO part's order and -- when it is sorted by row -- the per-tile entry ranges the
// one-launch kernel (spmv_coo_hyb.hip hyb_tile_kernel) reads (one int per 256 rows, owned by the plan); a heavy or unsorted COO
// part gets a COO plan of its own for the second launch.
CMI_API int cmi_plan_create_hyb(int dtype, int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row,
                                int64_t coo_entries, const int32_t *coo_row_indices, const cmi_config *cfg_ell,
                                const cmi_config *cfg_coo, void *stream, cmi_plan **plan_out)
{
    if (!plan_out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_hyb: null result pointer");
    *plan_out = nullptr;
    if (dtype < 0 || dtype > 1) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_hyb: bad value type");
    if (num_rows < 0 || num_cols < 0 || ell_entries_per_row < 0 || coo_entries < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_hyb: negative size");
    if (num_rows > INT32_MAX || num_cols > INT32_MAX || ell_entries_per_row > INT32_MAX)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_hyb: sizes exceed the int32 index type");
    if (coo_entries > 0 && !coo_row_indices) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create_hyb: the COO part's row indices are needed");
    cmi_plan *p = new (std::nothrow) cmi_plan;
    if (!p) return fail(CMI_ERROR_ALLOC, "cmi_plan_create_hyb: out of host memory");
    p->format = CMI_FORMAT_HYB;
    p->dtype = dtype;
    p->rows = num_rows;
    p->cols = num_cols;
    p->nnz = num_rows * ell_entries_per_row;
    p->prof = row_profile{};
    p->coo_sorted = -1;
    p->hyb_width = ell_entries_per_row;
    p->hyb_coo = coo_entries;
    select_config(CMI_FORMAT_ELL, dtype, num_rows, num_cols, p->nnz, cfg_ell, &p->cfg);
    select_config(CMI_FORMAT_COO, dtype, num_rows, num_cols, coo_entries, cfg_coo, &p->hyb_coo_cfg);
    hipStream_t s = as_stream(stream);
    int st = CMI_SUCCESS;
    if (coo_entries > 0 && num_rows > 0) {
        int sorted = 0, long_runs = 0;
        st = coo_rows_sorted(num_rows, coo_entries, coo_row_indices, s, &sorted, &long_runs);
        if (st == CMI_SUCCESS) p->coo_sorted = sorted;
        // sorted: the two-launch path's COO half is the tile kernel (storage-order sums on top of the ELL half: still one chain)
        if (st == CMI_SUCCESS && sorted && !long_runs && coo_entries >= 4 && (!cfg_coo || cfg_coo->kernel == CMI_KERNEL_AUTO))
            select_config(CMI_TABLE_COO_SORTED, dtype, num_rows, num_cols, coo_entries, cfg_coo, &p->hyb_coo_cfg);
        const char *force = std::getenv("CMI_HYB_ONE_LAUNCH"); // "0": never, "1": whenever the COO part is sorted, else: by its weight
        const bool never = force && force[0] == '0', always = force && force[0] == '1';
        if (st == CMI_SUCCESS && sorted && !never && coo_entries <= INT32_MAX - 4096 &&
            (always || (double)coo_entries <= kHybFusedMaxPerRow * (double)num_rows)) {
            const int64_t tiles = ceil_div(num_rows, kHybTileRows);
            int *max_dev = nullptr, max_in_tile = 0;
            hipError_t e = hipMalloc((void **)&p->hyb_tile_start, (size_t)(tiles + 1) * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc((void **)&max_dev, sizeof(int));
            if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create_hyb: tile ranges");
            if (st == CMI_SUCCESS) st = hyb_tile_starts(num_rows, coo_entries, coo_row_indices, p->hyb_tile_start, max_dev, s);
            if (st == CMI_SUCCESS) {
                e = hipMemcpyAsync(&max_in_tile, max_dev, sizeof(int), hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) st = hip_fail(e, "cmi_plan_create_hyb: tile ranges");
            }
            if (max_dev) (void)hipFree(max_dev);
            if (st == CMI_SUCCESS && !always && max_in_tile > kHybFusedMaxInTile) { // a few rows hold the COO part: two launches
                (void)hipFree(p->hyb_tile_start);
                p->hyb_tile_start = nullptr;
            }
        }
        // two launches: the COO half multiplies through a COO plan of its own (sorted entries: the CSR kernels on row offsets,
        // accumulating on top of the ELL half -- per row still the host chain)
        if (st == CMI_SUCCESS && !p->hyb_tile_start)
            st = plan_create(CMI_FORMAT_COO, dtype, num_rows, num_cols, coo_entries, coo_row_indices, nullptr, cfg_coo, stream, &p->hyb_coo_plan);
        if (st == CMI_SUCCESS) {
            st = fingerprint(coo_entries, coo_row_indices, s, &p->fp_index);
            p->has_fp_index = st == CMI_SUCCESS;
        }
    }
    if (st != CMI_SUCCESS) { (void)cmi_plan_destroy(p); return st; }
    *plan_out = p;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_hyb_launches(const cmi_plan *plan, int *launches)
{
    if (!plan || !launches || plan->format != CMI_FORMAT_HYB) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_hyb_launches: not a HYB plan");
    *launches = (plan->hyb_tile_start || plan->hyb_coo == 0) ? 1 : 2;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_destroy(cmi_plan *plan)
{
    if (plan && plan->hyb_tile_start) (void)hipFree(plan->hyb_tile_start);
    if (plan && plan->hyb_coo_plan) (void)cmi_plan_destroy(plan->hyb_coo_plan);
    if (plan && plan->coo_csr_plan) (void)cmi_plan_destroy(plan->coo_csr_plan);
    if (plan && plan->coo_offsets) (void)hipFree(plan->coo_offsets);
    if (plan && plan->wave_row_start) (void)hipFree(plan->wave_row_start);
    if (plan && plan->csr16_base) (void)hipFree(plan->csr16_base);
    if (plan && plan->csr16_cols) (void)hipFree(plan->csr16_cols);
    if (plan && plan->runs_start) (void)hipFree(plan->runs_start);
    if (plan && plan->runs_pieces) (void)hipFree(plan->runs_pieces);
    if (plan && plan->runs_packed) (void)hipFree(plan->runs_packed);
    if (plan && plan->csr16_packed) (void)hipFree(plan->csr16_packed);
    delete plan;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_config(const cmi_plan *plan, cmi_config *out)
{
    if (!plan || !out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_config: null argument");
    *out = plan->cfg;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_info(const cmi_plan *plan, int64_t *max_row_length, int64_t *entries_in_long_rows, int *coo_sorted,
                          int *storage_order_sums)
{
    if (!plan) return fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_info: null plan");
    if (max_row_length) *max_row_length = plan->format == CMI_FORMAT_CSR ? plan->prof.max_len : -1;
    if (entries_in_long_rows) *entries_in_long_rows = plan->format == CMI_FORMAT_CSR ? plan->prof.in_long : -1;
    if (coo_sorted) *coo_sorted = plan->coo_sorted;
    if (storage_order_sums) {
        int exact = 0;
        const cmi_config &c = plan->cfg;
        switch (plan->format) {
        case CMI_FORMAT_CSR:
            // scalar / pipe: always; stream: one lane per row and no row long enough for the cooperative path
            exact = c.kernel == CMI_CSR_SCALAR || c.kernel == CMI_CSR_STREAM_PIPE || c.kernel == CMI_CSR_STREAM_C16 || c.kernel == CMI_CSR_STREAM_WAVE || c.kernel == CMI_CSR_STREAM_WAVEV || c.kernel == CMI_CSR_STREAM_WAVEX || c.kernel == CMI_CSR_STREAM_WAVER || c.kernel == CMI_CSR_STREAM_PACKED ||
                    (c.kernel == CMI_CSR_STREAM && c.threads_per_row <= 1 && (c.threads_per_row == 1 || plan->prof.max_len < 512));
            break;
        case CMI_FORMAT_ELL: exact = ell_lanes_per_row(c, plan->rows, plan->rows > 0 ? plan->nnz / plan->rows : 0) == 1; break;
        case CMI_FORMAT_DIA: exact = 1; break;
        case CMI_FORMAT_COO:
            if (plan->coo_csr_plan) { // through the row offsets: the CSR kernel's class
                int sub = 0;
                (void)cmi_plan_info(plan->coo_csr_plan, nullptr, nullptr, nullptr, &sub);
                exact = sub;
            } else
                exact = c.kernel == CMI_COO_TILE;
            break;
        default: // HYB: one launch = one chain per row; two launches: the same chain when the COO half is the tile kernel, else atomics
            exact = plan->hyb_tile_start != nullptr;
            if (!exact && ell_lanes_per_row(c, plan->rows, plan->hyb_width) == 1) {
                int sub = plan->hyb_coo == 0;
                if (plan->hyb_coo_plan) (void)cmi_plan_info(plan->hyb_coo_plan, nullptr, nullptr, nullptr, &sub);
                exact = sub;
            }
            break;
        }
        *storage_order_sums = exact;
    }
    return CMI_SUCCESS;
}
