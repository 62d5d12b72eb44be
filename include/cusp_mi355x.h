/*
 * cusp_mi355x.h -- C-ABI of the MI355X-native SpMV engine that sits behind
 * cusp::multiply() for cusp::{csr,coo,ell,dia,hyb}_matrix<int, T, device_memory>.
 *
 * Plain pointers and sizes only: no torch, Thrust or C++ types cross this
 * boundary.  All pointers named Ap/Aj/Ax/x/y/... are DEVICE pointers (HBM);
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  Every
 * entry point returns a cmi_status (0 = success); launches are asynchronous on
 * `stream` like the reference's, but -- unlike the reference, which never checks
 * a launch (cusp/system/cuda/detail/multiply/csr_vector_spmv.h:204-208) -- launch
 * errors are reported.  Nothing here takes ownership of caller memory; no
 * cmi_spmv_* entry point allocates device memory or synchronises the stream
 * (what a multiply needs to know about a matrix beyond its arrays -- the
 * row-length profile that steers the CSR kernels, whether a COO matrix is
 * sorted by row -- is found ONCE by cmi_plan_create, which does synchronise,
 * and handed to the cmi_spmv_*_plan_* entry points).
 *
 * Each declaration cites the reference interface it replaces (paths relative to
 * the reference tree).  The header-only C++ layer in
 * cusp-autotuned_amd/include/cusp/ forwards cusp::multiply to these symbols; a
 * maintainer of the reference would bind them as shown in INTEGRATION.md.
 *
 * Index type is int32 (the reference's `int`), value types are f64 and f32.
 */
#ifndef CUSP_MI355X_H
#define CUSP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMI_VERSION 400 /* 0.4.0: run-compressed column copy (CMI_CSR_STREAM_WAVER), packed tiles (CMI_CSR_STREAM_PACKED), plan-less wave tiles; 0.3.0: communicator + collectives (RCCL) behind the boundary, cmi_plan_validate, CMI_CSR_STREAM_WAVEV */

typedef enum cmi_status {
    CMI_SUCCESS = 0,
    CMI_ERROR_INVALID_VALUE = 1, /* bad size / null pointer / bad config: cusp::invalid_input_exception */
    CMI_ERROR_HIP = 2,           /* a HIP runtime call or kernel launch failed: cusp::runtime_exception */
    CMI_ERROR_NOT_SUPPORTED = 3, /* config names a kernel variant that does not exist */
    CMI_ERROR_NO_DEVICE = 4,     /* no gfx950 device visible */
    CMI_ERROR_ALLOC = 5,         /* hipMalloc failed: std::bad_alloc */
    CMI_ERROR_IO = 6,            /* tuning table could not be read / written */
    CMI_ERROR_COMM = 7           /* RCCL could not be loaded, or a communicator call failed: cusp::runtime_exception */
} cmi_status;

/* Human-readable name of a status, and the message of the last failure on this thread. */
const char *cmi_status_string(int status);
const char *cmi_last_error(void);
int cmi_version(void);

/* ------------------------------------------------------------------------- */
/* Device + memory: what cusp::device_memory containers sit on.                */
/* Replaces thrust::device_malloc_allocator (cusp/detail/memory.inl:28-35) and  */
/* the H<->D copies done by cusp::array1d converting construction               */
/* (cusp/array1d.h:98-242).                                                     */
/* ------------------------------------------------------------------------- */
int cmi_device_count(int *count);
int cmi_set_device(int device);
int cmi_get_device(int *device);
/* name: caller buffer of name_len bytes; cus: compute units; hbm_bytes: total memory */
int cmi_device_info(int device, char *name, size_t name_len, int *cus, int64_t *hbm_bytes);

int cmi_malloc(void **ptr, size_t bytes);
int cmi_free(void *ptr);
int cmi_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int cmi_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int cmi_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int cmi_memset(void *dst, int byte_value, size_t bytes, void *stream);
/* Peer mapping for the row-block sharded SpMV (the reference has no multi-GPU  */
/* code).  One process per GPU: a rank exports the handle of a cmi_malloc'ed buffer (its slice of x  */
/* inside the full-length exchange buffer), its neighbours open it once, and before each multiply a  */
/* rank PULLS the boundary values its rows reference with ONE cmi_copy_ranges launch on its own      */
/* stream -- loads over xGMI, no collective.  Visibility is at kernel boundaries: the caller orders   */
/* the producing kernels of the peers before the pull (bench: barrier; CG: its all-reduces).         */
#define CMI_IPC_HANDLE_BYTES 64
#define CMI_MAX_COPY_RANGES 16
int cmi_device_can_access_peer(int device, int peer_device, int *can_access); /* same device: 1 */
int cmi_ipc_get_handle(void *dev_ptr, void *handle_out /* CMI_IPC_HANDLE_BYTES */);
int cmi_ipc_open_handle(const void *handle, void **peer_ptr);
int cmi_ipc_close_handle(void *peer_ptr);
/* dst[i][0, bytes[i]) <- src[i][0, bytes[i]) for i < count <= CMI_MAX_COPY_RANGES, one launch.      */
/* src / dst / bytes are HOST arrays of device pointers and lengths.                                  */
int cmi_copy_ranges(int count, const void *const *src, void *const *dst, const int64_t *bytes, void *stream);
/* Page-locked host memory and a device->host copy that is ordered on `stream` but NOT waited for  */
/* (pair with cmi_event_record / cmi_event_synchronize).  Used by cusp::krylov::cg for its one     */
/* host read per iteration (the monitor's residual norm, reference cusp/detail/monitor.inl:181-207) */
/* so that the next SpMV is already queued when the host blocks.                                    */
int cmi_malloc_host(void **ptr, size_t bytes);
int cmi_free_host(void *ptr);
int cmi_memcpy_d2h_async(void *dst_pinned, const void *src, size_t bytes, void *stream);
int cmi_stream_create(void **stream);
int cmi_stream_destroy(void *stream);
int cmi_stream_synchronize(void *stream);
int cmi_device_synchronize(void);

/* hipEvent-based timing on a stream (replaces performance/timer.h:23-54). */
int cmi_event_create(void **event);
int cmi_event_destroy(void *event);
int cmi_event_record(void *event, void *stream);
int cmi_event_synchronize(void *event);
int cmi_stream_wait_event(void *stream, void *event); /* work enqueued on `stream` afterwards waits for `event`; the host does not block */
int cmi_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */

/* ------------------------------------------------------------------------- */
/* Launch-shape / kernel-variant selection (replaces the KTT tuning           */
/* parameters of cusp/system/cuda/ktt/{csr,ell,dia,coo}_multiply.h and the      */
/* fixed selector of cusp/system/cuda/detail/multiply/csr_vector_spmv.h        */
/* :225-258).  A NULL config means: look the shape up in the persisted tuning   */
/* table (below), falling back to built-in heuristics.                          */
/* ------------------------------------------------------------------------- */
typedef enum cmi_format {
    CMI_FORMAT_CSR = 0,
    CMI_FORMAT_ELL = 1,
    CMI_FORMAT_DIA = 2,
    CMI_FORMAT_COO = 3,
    CMI_FORMAT_HYB = 4,
    CMI_FORMAT_COUNT = 5,     /* number of matrix formats */
    CMI_TABLE_COO_SORTED = 5, /* a key of the tuning table only, not a matrix format: the launch shape of a COO multiply
                                 whose plan found the entries sorted by row (CMI_COO_TILE and its cache policy / XCD
                                 dealing); the CMI_FORMAT_COO key stays with the order-agnostic kernels, which is what a
                                 plan-less call must run */
    CMI_TABLE_KEYS = 6
} cmi_format;

typedef enum cmi_dtype { CMI_F64 = 0, CMI_F32 = 1 } cmi_dtype;

typedef enum cmi_kernel {
    CMI_KERNEL_AUTO = 0,
    /* CSR */
    CMI_CSR_SCALAR = 1, /* one lane per row          (ref: csr_scalar.h:51-73)                   */
    CMI_CSR_VECTOR = 2, /* threads_per_row lanes/row (ref: csr_vector_spmv.h:71-161, THREADS_PER_ROW) */
    CMI_CSR_STREAM = 3, /* LDS-staged nnz tile, sequential per-row sum (bit-exact vs host order); threads_per_row
                           0: rows of 512+ entries are streamed by their whole workgroup instead (re-associated,
                           <= 1e-6; ~1-2 ns instead of ~12 ns per entry), 1: storage order for every row,
                           2..64: that many lanes per row (lane-strided partial sums)                    */
    CMI_CSR_STREAM_PIPE = 4, /* persistent, software-pipelined csr_stream (next tile's streams in flight) */
    CMI_CSR_BALANCED = 5,   /* merge-path split of row ends + entries: equal work per tile whatever the row
                               lengths (a few huge rows, power-law tails, runs of empty rows); replaces KTT's
                               csr_kernel_balanced (cuda/ktt/kernels/csr_kernel.h:316-375); block_size 512 (fixed);
                               items_per_thread = consecutive tiles per workgroup (0: 4), blocks_per_cu > 0:
                               persistent grid of CUs*blocks_per_cu workgroups instead              */
    CMI_CSR_STREAM_C16 = 6, /* OPT-IN, plans of cmi_plan_create_csr only: csr_stream's single-pass tile kernel reading a
                               16-bit copy of the column indices that the plan builds and owns (per tile: smallest column
                               + uint16 offsets; 2 bytes per entry of extra HBM) -- 10 nnz + 20 N bytes per multiply
                               instead of 12 nnz + 20 N, same products, same storage-order sums, same bits.  Granted only
                               if EVERY tile spans < 65536 columns and fits one LDS pass; otherwise the plan's config says
                               CMI_CSR_STREAM and nothing is built.  config fields: csr_stream's (0 = the table's).  With the
                               table's shape on stencil-like rows (where a plan would run CMI_CSR_STREAM_WAVE) the copy is tiled
                               per wave instead -- cmi_plan_config then reads block_size 256, rows_per_block 64 (the copy's
                               tile), items_per_thread = entries per lane -- and the wave-tile kernel reads it            */
    CMI_CSR_STREAM_WAVE = 7, /* csr_stream's lane-strided single-pass body with WAVE-PRIVATE tiles, for matrices whose rows all have
                               (nearly) the same short length (stencils): each 64-lane wave owns rows_per_block / (block_size / 64)
                               <= 64 consecutive rows, lane l requests entries l, l + 64, ... of the wave's tile, the products are
                               parked in the wave's own LDS region and every lane adds its row in storage order -- no workgroup
                               barrier, every lane owns a row.  items_per_thread = entries per lane (2..10): 64 x that many must
                               hold a wave's tile (a tile that does not fit is summed one lane per row from the arrays: correct,
                               slow).  Bit-exact.  A CSR plan selects it by itself when the longest row is <= 10 entries and the
                               mean within 7 % of it ($CMI_CSR_WAVE=0: never); never selected without a plan.
                               OPT-IN for irregular short rows: a plan made with this kernel and rows_per_block < 0 builds (and owns,
                               8 bytes per tile) a partition of the rows into wave tiles -- tile t = the rows whose first entry lies in
                               [t Q, (t + 1) Q), Q = 64 x items_per_thread - longest row -- and runs the same body on it
                               (items_per_thread 0: floor(mean + longest / 64)); cmi_plan_config then reads rows_per_block 0.  Faster than
                               csr_stream on FEM-like rows (thermal2-like: 0.90-0.94 of its time), slower on large matrices with
                               scattered columns: no plan selects it by itself                                         */
    CMI_CSR_STREAM_WAVEV = 8, /* plans only (round 3): WAVE-PRIVATE tiles with csr_stream's 16-byte-vector body, for rows of ~16-250 entries
                               (FEM / KKT matrices).  The plan partitions the rows into wave tiles -- tile t = the rows whose first entry
                               lies in [t Q, (t + 1) Q), Q = 256 x items_per_thread - longest row - 3; 8 bytes per tile of plan-owned
                               memory -- and each 64-lane wave streams its tile with items_per_thread (1, 2, 4; 0 = by the mean row
                               length) int4 index vectors + value vectors per lane, parks the products in its own LDS region and adds
                               every row in storage order.  No workgroup barrier (csr_stream's waves sit at theirs for most of their
                               life on such rows: profiles/r03_long_rows_pmc_csr_stream.json), every wave owns rows.  Bit-exact.  Needs
                               16-byte aligned Aj / Ax, no row of 512+ entries, longest row <= 128 x items_per_thread - 3.
                               Round 4: with items_per_thread = 1 (f32: 2) it is also what an AUTO plan runs on STENCIL rows of 5..8 entries in a
                               matrix beyond the Infinity Cache (the 5-point headline matrix: 0.95-0.96 of CMI_CSR_STREAM_WAVE's time replayed,
                               cold and with the fused dot; profiles/r04_stencil_tiles_ab.txt; $CMI_CSR_WAVE_VEC=0: never) -- and V = 2 instead
                               of 4 on irregular rows of fewer than 12 entries whose columns share x lines.                          */
    CMI_CSR_STREAM_WAVEX = 9, /* plans only (round 3): CMI_CSR_STREAM_WAVEV plus an x WINDOW in LDS -- the workgroup (four wave tiles) copies the
                               rows_per_block (0: 4096; the field carries the WINDOW LENGTH here, a multiple of 512 (f64) / 1024 (f32), at most
                               4096 / 8192) consecutive x entries around the diagonal position of its rows into LDS, coalesced, and gathers
                               from there; columns outside the window are gathered from memory as usual.  For GATHER-BOUND band matrices
                               (columns anywhere within a few thousand of the diagonal: every x entry is its own L1 lookup otherwise).
                               items_per_thread 2 or 4.  Bit-exact.  Needs 16-byte aligned Aj / Ax / x.                                   */
    CMI_CSR_STREAM_WAVER = 11, /* plans of cmi_plan_create_csr only (round 4), f64 and f32: CMI_CSR_STREAM_WAVEV's wave tiles reading a RUN-COMPRESSED copy of the
                               column indices that the plan builds and owns -- every row cut into pieces of 1..4 CONSECUTIVE columns, 4 bytes
                               per piece ((first column << 2) | (length - 1)), 16 bytes per wave tile -- instead of Aj: FEM (3 dof per node) and
                               KKT / 27-point matrices keep their columns in runs of 3 or more, so the index stream shrinks to about a third
                               (f64: 9.3-9.4 instead of 12 bytes per entry; f32: 5.3-5.4 instead of 8) and a piece's x values arrive with two
                               16-byte loads (f32: one) instead of one gather per entry.  The VALUES stay the caller's array (refreshing them in place is fine).  Same products,
                               storage-order sums: bit-exact.  An AUTO plan made with the columns selects it when the pieces average 2.2+ (f64) / 1.9+ (f32)
                               entries on a matrix the tuning table's "waver_rule" admits (cmi_tuning_waver_rule; also stencil rows of 8+ entries
                               with column runs -- 9-point -- where it is tried before CMI_CSR_STREAM_WAVE) ($CMI_CSR_WAVER=0: never, =1: whenever the rows
                               qualify); asked for explicitly it is refused only where the tile cannot hold the longest row.  items_per_thread
                               1, 2, 4 (0: 4) = 256 x that many slots per wave tile; threads_per_row = entries per piece at most, 3 or 4
                               (0: 3 where that costs at most 3 % more pieces than 4 -- pieces of three leave no LDS bank conflict between
                               them -- else 4).  Needs fewer than 2^30 columns (f32: at least 4), no row of 512+ entries,
                               16-byte aligned Ax.  cmi_plan_validate checks the column indices.                     */
    CMI_CSR_STREAM_PACKED = 12, /* OPT-IN, plans of cmi_plan_create_csr_values only (round 4): CMI_CSR_STREAM_WAVER with the pieces AND THE VALUES of
                               every wave tile laid side by side in one plan-owned buffer ([pieces | pad to 16 | values | pad to 16] per tile),
                               so that a wave's requests are ONE contiguous span of HBM.  The plan then owns a COPY OF THE VALUES: values
                               refreshed in place are NOT seen -- destroy the plan and make a new one (cmi_plan_validate_values tells).
                               Never selected by itself.  Bit-exact.                                                                   */
    /* ELL */
    CMI_ELL_ROW = 10, /* one lane per row            (ref: ell_spmv.h:55-93); threads_per_row 2,4,8,16: that many
                         lanes per row, each summing every 2nd / 4th / ... slot (ref: THREADS_PER_ROW of ktt
                         kernels/ell_kernel.h:102-109,165-173) -- re-associated sums, <= 1e-6; threads_per_row 0
                         picks width/16 lanes (<= 16) for matrices >= 32 slots wide with < 131072 rows (>= 64 wide:
                         < 524288 rows): 6x on 2000 x 512, 1.2x on 200000 x 64; threads_per_row 1 never          */
    /* DIA */
    CMI_DIA_ROW = 20, /* one lane per row, offsets in LDS (ref: dia_spmv.h:69-126)               */
    /* COO */
    CMI_COO_SEGMENTED = 30, /* wave segmented reduction + f64/f32 atomics at segment tails
                               (ref: ktt kernels/coo_kernel.h:289-369, coo_flat_spmv.h:231-311)  */
    CMI_COO_LANE4 = 31,     /* four consecutive entries per lane (16-byte vector loads), runs reduced in
                               registers, one wave scan per 256 entries (ref: ktt coo_direct_multi,
                               kernels/coo_kernel.h:64-106, VALUES_PER_THREAD)                     */
    CMI_COO_TILE = 32,      /* ROW-SORTED entries only (cmi_spmv_coo_plan_*, or an explicit config: then the
                               caller vouches for the order): 1024 entries per workgroup through LDS, row
                               boundaries from the row indices, storage-order sums, plain stores of whole
                               runs of y -- no zero fill, no atomics (ref: coo_flat_spmv.h:231-463 needs
                               three launches and two temporaries for the same contract)            */
} cmi_kernel;

typedef struct cmi_config {
    int32_t kernel;           /* cmi_kernel; CMI_KERNEL_AUTO = pick by heuristics                  */
    int32_t block_size;       /* threads per workgroup: 64..1024, multiple of 64; 0 = default      */
    int32_t threads_per_row;  /* CSR vector: 2,4,8,16,32,64; 0 = from mean row length.  CSR stream: lanes per
                                 row in the row-sum phase (0, 1, 2..64: see CMI_CSR_STREAM).  ELL: lanes per
                                 row (0 = auto, 1, 2..16: see CMI_ELL_ROW)                                  */
    int32_t rows_per_block;   /* CSR stream: rows per workgroup tile; 0 = from mean row length     */
    int32_t items_per_thread; /* CSR stream: 16-byte index vectors per lane per pass (1,2,4);
                                 ELL/DIA: rows per lane (1,2); COO: entries per lane; 0 = default   */
    int32_t nontemporal;      /* memory policy bits: 1 = once-read matrix streams loaded with the nt
                                 hint, 2 = y stored with the nt hint; 4 (CSR stream, one lane per row)
                                 = the index / value streams of a tile requested LANE-STRIDED -- lane l
                                 takes entries l, l + block, ... from the tile's first entry, so each load
                                 instruction of a wave is one contiguous span -- instead of as 16-byte
                                 vectors per lane: faster for short rows (about 5 per row: 128 -> 124 us
                                 on the headline matrix), slower from ~25 per row; same bits either way.
                                 A table entry's bit 1 is dropped for a CSR matrix whose streams fit the
                                 256 MiB Infinity Cache (it is then served from there when loaded plainly) */
    int32_t xcd_swizzle;      /* CSR stream, ELL, DIA, COO tile: 0 = tiles (a workgroup's rows / entries) in
                                 launch order, 1 = one contiguous eighth of the tiles per XCD, C >= 2 =
                                 chunks of C tiles dealt round the XCDs (a chunk's x window is fetched into
                                 ONE L2); CSR stream_pipe: != 0 = chunked tile schedule               */
    int32_t blocks_per_cu;    /* persistent kernels: workgroups per CU in the grid; 0 = default (8)  */
} cmi_config;

/* Persisted tuning table (replaces the in-process KTT tuner state,              */
/* cusp/ktt/detail/ktt.inl:29-62,130-142, which the reference never persists).  */
/* Keyed by (format, dtype, bucket of mean entries per row).                     */
int cmi_tuning_load(const char *path);  /* JSON written by tools/autotune; NULL = $CMI_TUNING_TABLE */
int cmi_tuning_save(const char *path);
int cmi_tuning_clear(void);             /* back to built-in heuristics (cusp::ktt::reset_tuning)   */
int cmi_tuning_set(int format, int dtype, double mean_entries_per_row, const cmi_config *cfg);
/* The config a NULL-config call with this shape would run. */
int cmi_tuning_select(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                      cmi_config *out);

/* HYB split rule = the ELL width cutoff, tuned offline (tools/autotune_hyb.py) and persisted in the same table file.
 * Two rule kinds, both functions of the row-length histogram alone:
 *   CMI_HYB_RULE_REFERENCE  the reference's cusp::compute_optimal_entries_per_row
 *       (cusp/system/detail/generic/format_utils.inl:281-325 with cusp/detail/functional.inl:114-132): the smallest k with
 *       relative_speed * #{rows longer than k} < num_rows  or  #{rows longer than k} < threshold.  The reference hard-wires
 *       (3.0, 4096), "chosen empirically for a GTX280" (csr_to_other.h:248-254); this is what applies without a table.
 *   CMI_HYB_RULE_COST  the width that minimises the modelled time of the two launches
 *       num_rows * k  +  [the COO part is not empty] * (threshold + relative_speed * coo_entries(k))      (in ELL slots)
 *       -- relative_speed = cost of a COO entry in ELL slots, threshold = fixed cost of the second launch in ELL slots.
 *       The reference's rule is this model's marginal test without the launch term.  Measured on MI355X
 *       (tools/autotune_hyb.py, archive/profiles/r02_autotune_hyb*), once per generation of the kernels behind a HYB multiply:
 *       two launches with the COO tile kernel (1.3, 5e6); one launch for light COO parts (2.0, 0); and heavy COO parts
 *       through a COO plan's row offsets + the CSR kernel: (1.0, 2e6) for f64, (1.3, 2e6) for f32 -- a COO entry costs what
 *       an ELL slot costs, irregular matrices get a narrow ELL part; geometric-mean regret over the tuning set 1.11 against
 *       1.18 for the reference's (3, 4096).  Shipped: CMI_HYB_RULE_COST2 below (1.05).                                      */
/*   CMI_HYB_RULE_COST2  the cost model with the two regimes a HYB plan has (cmi_plan_hyb_launches): while the COO part is light
 *       (at most 3 entries per row on average) the multiply is ONE launch and a COO entry costs `light_speed` ELL slots; beyond,
 *       it is ELL + the CSR kernel on the COO plan's row offsets:  threshold + relative_speed * coo_entries(k).  Offline on the
 *       sweep log (tools/autotune_hyb.py --refit): geometric-mean regret 1.03 (f64) / 1.06 (f32) against 1.11 for COST.
 *       THIS is the shipped rule (tuned/gfx950.json "hyb_rule"; the COST3 form printed by tools/autotune_hyb_holdout.py is an experiment
 *       of that tool, no better on its hold-out, and not a rule kind of the library).
 *       LIMIT OF THE RULE FORM: the width is a function of the row-length HISTOGRAM alone.  Held out (every 3-of-11 split,
 *       profiles/r03_autotune_hyb_holdout.txt) the regret is 1.09 in the geometric mean and up to 1.48 on one matrix; on the full fit
 *       f32 rows of uniform length 1..16 stay at 1.36 (the rule keeps one ELL slot where none is best).  No constants of this form close
 *       those cases -- what they miss (how the COO part's columns gather) is not in a histogram.  A caller who knows the matrix passes the
 *       width (cusp::hyb_matrix's num_entries_per_row / cmi_csr_to_hyb's width argument); the rule is the default, not a bound.          */
typedef enum cmi_hyb_rule_kind { CMI_HYB_RULE_REFERENCE = 0, CMI_HYB_RULE_COST = 1, CMI_HYB_RULE_COST2 = 2 } cmi_hyb_rule_kind;
int cmi_tuning_hyb_light_speed(int dtype, double *light_speed);      /* COST2's fourth parameter (persisted as "light_speed") */
int cmi_tuning_set_hyb_light_speed(int dtype, double light_speed);
int cmi_tuning_hyb_rule(int dtype, int *kind, double *relative_speed, int64_t *threshold);
int cmi_tuning_set_hyb_rule(int dtype, int kind, double relative_speed, int64_t threshold);
/* The width a rule gives for the CSR matrix with these row offsets: histogram of the row lengths on the device, rule on
 * the host.  kind < 0: the tuned rule (then relative_speed / threshold are ignored).  Set-up call: allocates scratch and
 * synchronises `stream`.  Row lengths beyond 4096 count as 4096 (no width beyond that is returned). */
int cmi_hyb_entries_per_row(int dtype, int64_t num_rows, const int32_t *Ap, int kind, double relative_speed,
                            int64_t threshold, int64_t *width_host, void *stream);

/* The run-compressed column copy's launch shape and AUTO gates (CMI_CSR_STREAM_WAVER, round 4), tuned offline (tools/autotune_waver.py:
 * items_per_thread x cap x xcd_swizzle swept on FEM / KKT matrices, every shape validated before it is timed) and persisted in the same
 * table file as "waver_rule" -- what the KTT tuner's per-kernel parameter space (cuda/ktt/csr_multiply.h:239-247) becomes for this kernel.
 * An AUTO plan made with the column indices takes the copy when the matrix has at least min_entries entries and its pieces of consecutive
 * columns average at least min_piece entries; it then runs 256 x items_per_thread slots per wave tile, pieces cut at `cap` (0: 3 where that
 * costs at most 3 % more pieces than 4), tiles dealt to the XCDs in chunks of xcd_swizzle workgroups (0: launch order).            */
typedef struct cmi_waver_rule {
    int32_t items_per_thread; /* 1, 2, 4 */
    int32_t cap;              /* 0, 3, 4 */
    int32_t xcd_swizzle;      /* >= 0 */
    int32_t reserved;
    double min_piece;         /* >= 1 */
    int64_t min_entries;      /* >= 0 */
} cmi_waver_rule;
int cmi_tuning_waver_rule(int dtype, cmi_waver_rule *rule);           /* the table's rule, else the built-in one */
int cmi_tuning_set_waver_rule(int dtype, const cmi_waver_rule *rule); /* layered on top of the shipped table; cmi_tuning_save writes it */

/* ------------------------------------------------------------------------- */
/* Plans (cmi_plan_create / destroy / select).        */
/* A plan holds what the library learns about ONE matrix before its first      */
/* multiply, so that no multiply has to: the launch shape (tuning table or the  */
/* caller's config, completed), for CSR the row-length profile (longest row,    */
/* entries in rows of 512+: picks the long-row instance of csr_stream or the     */
/* merge-path kernel), for COO whether the entries are sorted by row (then the   */
/* plan builds the row offsets they imply -- 4 bytes per row that it owns -- and   */
/* every multiply runs the CSR kernels on them, never reading the row indices:      */
/* 12 instead of 16 bytes per entry, storage-order sums, no zero fill, no atomics;  */
/* an explicit CMI_COO_TILE config keeps the COO tile kernel).                      */
/* The reference has no equivalent object: its KTT path keeps such state in       */
/* function-local statics keyed by nothing (cuda/ktt/csr_multiply.h:22-29,239-247) */
/* and recomputes `row_starts` on the host per call.                               */
/*   cmi_plan_create reads the index array on the device and SYNCHRONISES `stream`  */
/*   (one small kernel + a 16-byte read-back); everything after it is asynchronous.  */
/*   A plan owns little device memory -- a HYB plan's tile ranges, a wave partition, a sorted-COO plan's row offsets, the       */
/*   run-compressed / 16-bit column copies, the opt-in packed tiles: cmi_plan_device_bytes; all freed by cmi_plan_destroy -- and */
/*   does not keep the pointers: the arrays are   */
/*   passed again at every multiply; they must be the ones the plan was made for     */
/*   (same sizes are checked; contents are the caller's promise).                     */
/*   CONTRACT: the index arrays a plan was made from -- CSR row offsets, COO / HYB-COO   */
/*   row indices, and for a plan made by cmi_plan_create_csr* the column indices --  MUST NOT     */
/*   CHANGE IN PLACE while the plan is used.  A plan caches structure derived from     */
/*   them (a sorted-COO plan its row offsets, a C16 plan the 16-bit columns, a WAVER plan the pieces of   */
/*   consecutive columns, a wave partition the tile bounds): after an in-place edit the multiplies read that stale  */
/*   structure and y is WRONG (never a fault: every kernel bounds its LDS tile by what   */
/*   it reads).  Destroy the plan and make a new one when the structure changes;         */
/*   cmi_plan_validate (below) tells whether that has happened.  cmi_plan_create also      */
/*   refuses CSR row offsets that do not run from 0 to num_entries.                        */
/*   Thread-safe once created (read-only).                                                */
/* ------------------------------------------------------------------------- */
typedef struct cmi_plan cmi_plan;
/* index_array: CSR row offsets (num_rows + 1), COO row indices (num_entries), NULL for ELL / DIA / HYB  */
/* (their plan is the resolved launch shape only).  `num_entries`: entries for CSR / COO, slots          */
/* (num_rows * width or * diagonals) for ELL / DIA -- what the tuning table is keyed by.  cfg may be NULL. */
int cmi_plan_create(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                    const int32_t *index_array, const cmi_config *cfg, void *stream, cmi_plan **plan);
/* CSR with both structure arrays: cmi_plan_create(CMI_FORMAT_CSR, ...) plus, when asked for, the 16-bit column copy  */
/* of CMI_CSR_STREAM_C16 -- asked for by cfg->kernel == CMI_CSR_STREAM_C16, or for every AUTO-kernel plan after             */
/* cmi_set_index_compression(1) (initial value: $CMI_COMPRESS_INDICES).  cmi_plan_config tells whether it was granted.      */
int cmi_plan_create_csr(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_offsets,
                        const int32_t *column_indices, const cmi_config *cfg, void *stream, cmi_plan **plan);
/* COO with both index arrays (round 4): cmi_plan_create(CMI_FORMAT_COO, ...) whose CSR sub-plan -- row-sorted entries -- is made WITH   */
/* the columns, so that a matrix whose columns come in runs multiplies from the run-compressed copy (CMI_CSR_STREAM_WAVER) in COO too.  */
/* cmi_plan_validate then wants both arrays.  Replaces the per-call temporaries of coo_flat_spmv.h:387-463 like cmi_plan_create does.    */
int cmi_plan_create_coo(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_indices,
                        const int32_t *column_indices, const cmi_config *cfg, void *stream, cmi_plan **plan);
/* ... and with the VALUES (device pointer, `dtype` elements): what cmi_plan_create_csr does, plus -- asked for by                    */
/* cfg->kernel == CMI_CSR_STREAM_PACKED -- the packed per-tile copy of pieces and values.  Any other config: the values are ignored.  */
int cmi_plan_create_csr_values(int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *row_offsets,
                               const int32_t *column_indices, const void *values, const cmi_config *cfg, void *stream, cmi_plan **plan);
/* Have the VALUES a CMI_CSR_STREAM_PACKED plan copied changed since?  (*valid_host = 1 for every other plan: nothing of the values  */
/* is kept.)  One streaming pass, synchronises `stream`.                                                                             */
int cmi_plan_validate_values(const cmi_plan *plan, const void *values, void *stream, int *valid_host);
/* Bytes of device memory the plan owns (partitions, offsets, column copies, packed tiles).                                           */
int cmi_plan_device_bytes(const cmi_plan *plan, int64_t *bytes);
int cmi_set_index_compression(int on);
int cmi_get_index_compression(void);
/* HYB: launch shapes of both parts (cfg_* may be NULL) and, when the COO part's row indices are sorted (what every     */
/* conversion produces: csr_to_other.h:229-306) and the part is light (cmi_plan_hyb_launches), the per-tile entry ranges   */
/* that let cmi_spmv_hyb_plan_* run the whole multiply as ONE launch.  That array (one int per 256 rows) is the only device memory a plan owns; cmi_plan_destroy     */
/* frees it.  Synchronises `stream`.                                                                                       */
int cmi_plan_create_hyb(int dtype, int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row, int64_t coo_entries,
                        const int32_t *coo_row_indices, const cmi_config *cfg_ell, const cmi_config *cfg_coo, void *stream,
                        cmi_plan **plan);
/* How many kernels a multiply through this HYB plan launches: 1 (hyb_tile: COO part sorted and light -- at most 3       */
/* entries per row on average, no 256-row tile holding more than 4096 -- or empty) or 2 (ELL kernel, then a COO kernel        */
/* accumulating: heavy or unsorted COO parts).  $CMI_HYB_ONE_LAUNCH=0/1 at plan creation overrides the weight rule.           */
int cmi_plan_hyb_launches(const cmi_plan *plan, int *launches);
int cmi_plan_destroy(cmi_plan *plan);
/* Have the arrays the plan was made from changed since?  One streaming pass over them on the device (an order-sensitive    */
/* 64-bit checksum, compared with the one taken at creation); SYNCHRONISES `stream`.  *valid_host: 1 same contents, 0 edited  */
/* in place -> make a new plan.  index_array as for cmi_plan_create (HYB: the COO part's row indices); column_indices only     */
/* for plans that own a copy derived from them (CMI_CSR_STREAM_C16, _WAVER, _PACKED), NULL otherwise.  ELL / DIA plans: always valid. */
int cmi_plan_validate(const cmi_plan *plan, const int32_t *index_array, const int32_t *column_indices, void *stream,
                      int *valid_host);
/* The launch shape the plan's multiplies run (SURVEY's cmi_plan_select): kernel CMI_CSR_BALANCED means   */
/* the profile switched kernels.                                                                            */
int cmi_plan_config(const cmi_plan *plan, cmi_config *out);
/* What was measured: longest row and entries sitting in rows of 512+ (CSR; -1 otherwise), row-sortedness   */
/* (COO: 1/0; -1 otherwise), and whether every result is the storage-order sum (bit-identical to the host   */
/* loop: 1) or some rows are re-associated (within 1e-6: 0).                                                  */
int cmi_plan_info(const cmi_plan *plan, int64_t *max_row_length, int64_t *entries_in_long_rows, int *coo_sorted,
                  int *storage_order_sums);

/* ------------------------------------------------------------------------- */
/* SpMV: y = A*x (accumulate == 0; the reference's 3-argument cusp::multiply,  */
/* cusp/multiply.h:40,101 -> generic/multiply.inl:98-111) or y = y + A*x        */
/* (accumulate != 0; initialize = identity, as hyb's COO half uses,             */
/* generic/multiply/spmv.h:275-290).  x has num_cols elements, y num_rows.      */
/* ------------------------------------------------------------------------- */

/* Replaces cuda::detail::multiply(csr) (csr_vector_spmv.h:225-258), spmv_csr_scalar
 * (csr_scalar.h:82-109) and the KTT csr_spmv kernel (ktt/kernels/csr_kernel.h:378-410).
 * Host-order oracle: sequential/multiply/csr_spmv.h:42-74. */
int cmi_spmv_csr_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                     const int32_t *Aj, const double *Ax, const double *x, double *y, int accumulate,
                     const cmi_config *cfg, void *stream);
int cmi_spmv_csr_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                     const int32_t *Aj, const float *Ax, const float *x, float *y, int accumulate,
                     const cmi_config *cfg, void *stream);
/* The same multiply steered by a plan (no table lookup, no profile measurement, no allocation, no sync). */
int cmi_spmv_csr_plan_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                          const double *x, double *y, int accumulate, void *stream);
int cmi_spmv_csr_plan_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                          const float *x, float *y, int accumulate, void *stream);
/* The same fusion for ELL (ELLR with row_lengths) and DIA: one lane per row owns y[row], so <y, w> costs one  */
/* extra coalesced read of w and one partial per workgroup.                                                    */
int cmi_spmv_ell_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                         const int32_t *ell_Aj, const double *ell_Ax, const int32_t *row_lengths, const double *x,
                         double *y, const double *w, double *dot_dev, void *workspace, const cmi_config *cfg,
                         void *stream);
int cmi_spmv_dia_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                         const int32_t *diagonal_offsets, const double *values, const double *x, double *y,
                         const double *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream);
/* float matrices: the same one-pass forms; the scalar stays a double in device memory */
int cmi_spmv_ell_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                         const int32_t *Aj, const float *Ax, const int32_t *row_lengths, const float *x,
                         float *y, const float *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream);
int cmi_spmv_dia_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                         const int32_t *diagonal_offsets, const float *values, const float *x, float *y,
                         const float *w, double *dot_dev, void *workspace, const cmi_config *cfg, void *stream);
/* Row-length profile.  cmi_spmv_csr_* WITHOUT a plan runs the table's row-tile kernel whatever the row    */
/* lengths (correct for every matrix; a row of 10^5 entries is then summed by one lane).  A plan measures  */
/* the longest row once and switches to the long-row instance of csr_stream or to CMI_CSR_BALANCED when    */
/* that row alone would cost more than the whole multiply.  cmi_csr_max_row_length is the same measurement  */
/* for hosts that want to decide themselves (synchronises the stream).                                      */
int cmi_csr_max_row_length(int64_t num_rows, const int32_t *Ap, int64_t *max_length_host, void *stream);
/* y <- A x AND *dot_dev <- <y, w> (w: num_rows values; w may be x).  The CG step                  */
/* `y = A p; alpha = rz / dot(y, p)` (reference cusp/krylov/detail/cg.inl:80-83) in ONE pass: the   */
/* csr_stream workgroups leave per-tile partial sums in `workspace` (cmi_blas_workspace_bytes()),   */
/* folded by a fixed tree -- deterministic, no atomics.  y is bit-identical to cmi_spmv_csr_f64's.  */
/* When the selected kernel cannot fuse the dot it runs cmi_spmv_csr_f64 + cmi_blas_dot_f64.        */
int cmi_spmv_csr_dot_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                         const int32_t *Aj, const double *Ax, const double *x, double *y, const double *w,
                         double *dot_dev, void *workspace, const cmi_config *cfg, void *stream);
int cmi_spmv_csr_dot_plan_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                              const double *x, double *y, const double *w, double *dot_dev, void *workspace,
                              void *stream);
/* f32 matrix and vectors, <y, w> accumulated and returned as a DOUBLE (the scalars of the float CG stay doubles). */
int cmi_spmv_csr_dot_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ap,
                         const int32_t *Aj, const float *Ax, const float *x, float *y, const float *w,
                         double *dot_dev, void *workspace, const cmi_config *cfg, void *stream);
int cmi_spmv_csr_dot_plan_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                              const float *x, float *y, const float *w, double *dot_dev, void *workspace,
                              void *stream);

/* Replaces cuda::detail::multiply(ell) (ell_spmv.h:103-155) and ktt_ell_kernel / ktt_ellr_kernel
 * (ktt/kernels/ell_kernel.h:181-213).  Column-major num_rows x num_entries_per_row arrays with
 * leading dimension `pitch` (element (i,n) at n*pitch+i); padding slots have column -1.
 * row_lengths may be NULL; when given (the fork's ELLR, cusp/ktt/ellr_matrix.h:17-90) row i has
 * exactly row_lengths[i] leading valid slots.  Oracle: sequential/multiply/ell_spmv.h:41-76. */
int cmi_spmv_ell_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                     const int32_t *Aj, const double *Ax, const int32_t *row_lengths, const double *x,
                     double *y, int accumulate, const cmi_config *cfg, void *stream);
int cmi_spmv_ell_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries_per_row, int64_t pitch,
                     const int32_t *Aj, const float *Ax, const int32_t *row_lengths, const float *x,
                     float *y, int accumulate, const cmi_config *cfg, void *stream);

/* Replaces cuda::detail::multiply(dia) (dia_spmv.h:136-188) and ktt_dia_vector_kernel
 * (ktt/kernels/dia_kernel.h:236-252).  values column-major num_rows x num_diagonals, leading
 * dimension `pitch`.  Oracle: sequential/multiply/dia_spmv.h:43-82. */
int cmi_spmv_dia_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                     const int32_t *diagonal_offsets, const double *values, const double *x, double *y,
                     int accumulate, const cmi_config *cfg, void *stream);
int cmi_spmv_dia_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch,
                     const int32_t *diagonal_offsets, const float *values, const float *x, float *y,
                     int accumulate, const cmi_config *cfg, void *stream);

/* Replaces the COO flat trio (coo_flat_spmv.h:387-463, coo_serial.h:38-54), the Thrust
 * reduce_by_key fallback that device COO actually runs on modern Thrust
 * (generic/multiply/spmv.h:185-238) and the KTT coo_spmv composite (ktt/kernels/coo_kernel.h:372-392).
 * Entries may be in any order (sorted by row is fastest).  No scratch allocation.
 * Oracle: sequential/multiply/coo_spmv.h:42-68. */
int cmi_spmv_coo_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ai,
                     const int32_t *Aj, const double *Ax, const double *x, double *y, int accumulate,
                     const cmi_config *cfg, void *stream);
int cmi_spmv_coo_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, const int32_t *Ai,
                     const int32_t *Aj, const float *Ax, const float *x, float *y, int accumulate,
                     const cmi_config *cfg, void *stream);
/* With a plan that found the entries sorted by row (the reference's contract for coo_matrix,
 * cusp/coo_matrix.h:72) the tile kernel CMI_COO_TILE runs: a workgroup owns the rows that START in its
 * 1024 entries, finds the row boundaries from the row indices in LDS, adds each row's products in STORAGE
 * ORDER (bit-identical to sequential/multiply/coo_spmv.h:42-68) and stores whole runs of y: no zero fill,
 * no atomics; rows without entries are zeroed (or left, when accumulating) by the tile that owns the gap.
 * A plan that found them unsorted runs the order-agnostic kernels above. */
int cmi_spmv_coo_plan_f64(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const double *Ax,
                          const double *x, double *y, int accumulate, void *stream);
int cmi_spmv_coo_plan_f32(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const float *Ax,
                          const float *x, float *y, int accumulate, void *stream);
/* ... and with <y, w> (a double) in the same pass where the plan runs the CSR kernel on its row offsets (sorted entries).      */
int cmi_spmv_coo_dot_plan_f64(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const double *Ax, const double *x,
                              double *y, const double *w, double *dot_dev, void *workspace, void *stream);
int cmi_spmv_coo_dot_plan_f32(const cmi_plan *plan, const int32_t *Ai, const int32_t *Aj, const float *Ax, const float *x,
                              float *y, const float *w, double *dot_dev, void *workspace, void *stream);

/* HYB = ELL part (caller's accumulate) then COO part accumulating on top
 * (generic/multiply/spmv.h:275-290; oracle sequential/multiply/hyb_spmv.h:42-57).
 * cfg_ell / cfg_coo may be NULL. */
int cmi_spmv_hyb_f64(int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row, int64_t ell_pitch,
                     const int32_t *ell_Aj, const double *ell_Ax, int64_t coo_entries,
                     const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x,
                     double *y, int accumulate, const cmi_config *cfg_ell, const cmi_config *cfg_coo,
                     void *stream);
int cmi_spmv_hyb_f32(int64_t num_rows, int64_t num_cols, int64_t ell_entries_per_row, int64_t ell_pitch,
                     const int32_t *ell_Aj, const float *ell_Ax, int64_t coo_entries,
                     const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x,
                     float *y, int accumulate, const cmi_config *cfg_ell, const cmi_config *cfg_coo,
                     void *stream);
/* HYB through a plan of cmi_plan_create_hyb.  COO part sorted by row: ONE launch -- a workgroup owns 256 rows, a lane   */
/* walks its row's ELL slots and then adds the row's COO entries (staged through LDS) to the same accumulator, in entry  */
/* order: per row exactly the chain of sequential/multiply/hyb_spmv.h:55-56, so the result has the host loops' bits, y is  */
/* written once and nothing is zero-filled or accumulated with atomics.  Otherwise: the two launches of cmi_spmv_hyb_*.    */
int cmi_spmv_hyb_plan_f64(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                          const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x, double *y,
                          int accumulate, void *stream);
int cmi_spmv_hyb_plan_f32(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                          const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x, float *y,
                          int accumulate, void *stream);
/* ... and with <y, w> (a double) in the same pass (the CG step <A p, p>): fused where the plan runs one launch, else the       */
/* multiply followed by the library's dot.                                                                                      */
int cmi_spmv_hyb_dot_plan_f64(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                              const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax, const double *x, double *y,
                              const double *w, double *dot_dev, void *workspace, void *stream);
int cmi_spmv_hyb_dot_plan_f32(const cmi_plan *plan, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                              const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax, const float *x, float *y,
                              const float *w, double *dot_dev, void *workspace, void *stream);

/* ------------------------------------------------------------------------- */
/* On-device builders of the benchmark inputs.      */
/* cusp::gallery::poisson5pt (cusp/gallery/detail/poisson.inl:29-47,            */
/* stencil.inl:143-206) followed by the conversion to the target format         */
/* (conversions/dia_to_other.h:109-163, csr_to_other.h:56-70,155-227), done     */
/* directly in HBM so 1e7..1e8-row inputs do not go through the host.            */
/* Sizes: N = m*n rows, nnz = 5mn - 2m - 2n.                                     */
/* ------------------------------------------------------------------------- */
int64_t cmi_poisson5pt_num_entries(int64_t m, int64_t n);
/* Rows [row_begin, row_end) of the global matrix with GLOBAL column indices (row-block shard,
 * SURVEY.md section 8(e)); Ap has (row_end-row_begin)+1 entries starting at 0.
 * Pass 0, m*n for the whole matrix.  Aj/Ax must hold cmi_poisson5pt_shard_entries(). */
int64_t cmi_poisson5pt_shard_entries(int64_t m, int64_t n, int64_t row_begin, int64_t row_end);
int cmi_poisson5pt_csr_f64(int64_t m, int64_t n, int64_t row_begin, int64_t row_end, int32_t *Ap,
                           int32_t *Aj, double *Ax, void *stream);
int cmi_poisson5pt_csr_f32(int64_t m, int64_t n, int64_t row_begin, int64_t row_end, int32_t *Ap,
                           int32_t *Aj, float *Ax, void *stream);
/* DIA: offsets[5] = {-m,-1,0,1,m}; values 5 columns of leading dimension pitch (>= m*n). */
int cmi_poisson5pt_dia_f64(int64_t m, int64_t n, int64_t pitch, int32_t *offsets, double *values,
                           void *stream);
int cmi_poisson5pt_dia_f32(int64_t m, int64_t n, int64_t pitch, int32_t *offsets, float *values,
                           void *stream);

/* CSR -> ELL (width slots, leading dimension pitch, padding -1 / 0) keeping within-row order;
 * entries at within-row index >= width are dropped (they belong to HYB's COO part, below).
 * (csr_to_other.h:155-227) */
int cmi_csr_to_ell_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                       int64_t width, int64_t pitch, int32_t *ell_Aj, double *ell_Ax, void *stream);
int cmi_csr_to_ell_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                       int64_t width, int64_t pitch, int32_t *ell_Aj, float *ell_Ax, void *stream);
/* Number of explicit zeros among `n` values: the reference's CSR -> ELL reports num_entries without them
 * (csr_to_other.h:188, thrust::count(values, 0)).  Set-up call: synchronises the stream. */
int cmi_count_zeros_f64(int64_t n, const double *values, int64_t *count_host, void *stream);
int cmi_count_zeros_f32(int64_t n, const float *values, int64_t *count_host, void *stream);
/* CSR -> HYB, COO part (csr_to_other.h:229-306): the entries at within-row index >= width, in CSR order.
 * coo_offsets[i] = number of such entries in rows [0, i) -- an exclusive scan of max(0, len_i - width),
 * a function of the row offsets alone, supplied by the caller (num_rows entries).  The ELL part is
 * cmi_csr_to_ell with the same width. */
int cmi_csr_to_hyb_coo_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax, int64_t width,
                           const int32_t *coo_offsets, int32_t *coo_Ai, int32_t *coo_Aj, double *coo_Ax,
                           void *stream);
int cmi_csr_to_hyb_coo_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax, int64_t width,
                           const int32_t *coo_offsets, int32_t *coo_Ai, int32_t *coo_Aj, float *coo_Ax,
                           void *stream);
/* CSR -> DIA on the device (reference conversions/csr_to_other.h:73-153, there with Thrust sorts):   */
/* cmi_csr_diagonals flags the occupied diagonals in slot_map (num_rows + num_cols ints of scratch)  */
/* and lists their offsets (col - row) UNORDERED in diag_list (at most `capacity`; the true count    */
/* comes back in *num_diagonals_host -- larger than capacity means "too much fill-in, give up").     */
/* The caller sorts the few offsets ascending (the reference's order), uploads them, and             */
/* cmi_csr_to_dia_* zeroes `values` (pitch x num_diagonals, column-major) and scatters the entries.  */
int cmi_csr_diagonals(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, int32_t *slot_map,
                      int32_t *diag_list, int64_t capacity, int64_t *num_diagonals_host, void *stream);
int cmi_csr_to_dia_f64(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                       int64_t num_diagonals, int64_t pitch, const int32_t *offsets, int32_t *slot_map,
                       double *values, void *stream);
int cmi_csr_to_dia_f32(int64_t num_rows, int64_t num_cols, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                       int64_t num_diagonals, int64_t pitch, const int32_t *offsets, int32_t *slot_map,
                       float *values, void *stream);
/* Set-up helpers of the row-block sharded operator (no reference equivalent): the column window a row block gathers    */
/* from (smallest / largest column index; no entries: 0, -1; synchronises), and the row offsets of a block cut out of a larger       */
/* The rows of a row block that need nothing but the block's OWN slice of x (sharded multiply: they can be multiplied while the
 * halo is in flight -- SURVEY.md 8(f).4).  Rows whose columns all lie in [col_lo, col_hi) are interior; boundary rows of a banded
 * block cluster at its two ends, so the answer is ONE range: *first_host = 1 + the last row of the block's first half with a
 * column outside, *last_host = the first such row of the second half (num_rows if none): rows [first, last) are interior.
 * One pass over the column indices, synchronises the stream.                                                              */
int cmi_csr_interior_rows(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, int64_t col_lo, int64_t col_hi,
                          int64_t *first_host, int64_t *last_host, void *stream);
/* matrix (out[i] = Ap[i] - base for i <= num_rows).                                                                                  */
int cmi_csr_column_span(int64_t num_entries, const int32_t *Aj, int32_t *min_host, int32_t *max_host, void *stream);
int cmi_csr_rebase_offsets(int64_t num_rows, const int32_t *Ap, int32_t base, int32_t *out, void *stream);
/* CSR -> COO row indices (offsets_to_indices, csr_to_other.h:56-70). */
int cmi_csr_row_indices(int64_t num_rows, const int32_t *Ap, int32_t *Ai, void *stream);
/* The way back for ROW-SORTED entries (coo -> csr on the device; reference: cusp/system/detail/generic/conversions/
 * coo_to_other.h, which sorts first and then builds the offsets with lower_bound): Ap[num_rows + 1] from the row indices
 * in one pass, with the order checked on the way.  *sorted_host == 0 afterwards: the entries were not sorted by row (or a
 * row index was out of range) and Ap is unspecified -- the caller sorts, or converts on the host.  Synchronises the stream. */
int cmi_coo_row_offsets(int64_t num_rows, int64_t num_entries, const int32_t *Ai, int32_t *Ap, int *sorted_host, void *stream);
/* The COO container's ordering on the device (reference: cusp/sort.h:231 sort_by_row, :302 sort_by_row_and_column behind
 * coo_matrix::sort_by_row[_and_column], cusp/coo_matrix.h:208-224; is_sorted_by_row[_and_column]: same lines).  The
 * reference's device multiply REQUIRES row-sorted entries (cuda/detail/multiply/coo_flat_spmv.h:139-145): a caller with
 * entries in any order sorts ONCE with this call and then multiplies through a plan (sorted entries -> the CSR kernels on
 * plan-built row offsets, bit-exact) instead of paying the atomics kernels on every multiply.  In place, STABLE: the
 * entries of a row keep their storage order (and_column == 0), so the sorted matrix's row sums are the chains the host loop
 * forms on the unsorted one.  and_column != 0: by (row, column).  Entries already in the order asked for: one checking
 * pass, nothing moves.  Allocates scratch (about 12 + sizeof(value) bytes per entry; and_column: 20 + ...), synchronises
 * the stream.  A row index outside [0, num_rows): CMI_ERROR_INVALID_VALUE, arrays untouched.                              */
int cmi_coo_sort_by_row_f64(int64_t num_rows, int64_t num_cols, int64_t num_entries, int32_t *Ai, int32_t *Aj, double *Ax,
                            int and_column, void *stream);
int cmi_coo_sort_by_row_f32(int64_t num_rows, int64_t num_cols, int64_t num_entries, int32_t *Ai, int32_t *Aj, float *Ax,
                            int and_column, void *stream);
/* *sorted_host = 1 when the entries are ordered by row (and_column: by (row, column)); Aj may be NULL when and_column == 0.
 * One pass, synchronises the stream.                                                                                    */
int cmi_coo_is_sorted(int64_t num_rows, int64_t num_entries, const int32_t *Ai, const int32_t *Aj, int and_column,
                      int *sorted_host, void *stream);
/* ELL -> CSR and DIA -> CSR on the device (reference conversions/ell_to_other.h and dia_to_other.h:107-160: the entries
 * with a valid column -- DIA: and a non-zero value -- in row-major order): count per row, exclusive scan, scatter.
 * Two-call protocol: first with Aj == Ax == NULL -- Ap[num_rows + 1] and *num_entries_host are filled, size the arrays --
 * then with the arrays and their capacity (in entries).  Synchronise the stream. */
int cmi_ell_to_csr_f64(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj, const double *ell_Ax,
                       int32_t *Ap, int32_t *Aj, double *Ax, int64_t capacity, int64_t *num_entries_host, void *stream);
int cmi_ell_to_csr_f32(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj, const float *ell_Ax,
                       int32_t *Ap, int32_t *Aj, float *Ax, int64_t capacity, int64_t *num_entries_host, void *stream);
int cmi_dia_to_csr_f64(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch, const int32_t *offsets,
                       const double *values, int32_t *Ap, int32_t *Aj, double *Ax, int64_t capacity,
                       int64_t *num_entries_host, void *stream);
int cmi_dia_to_csr_f32(int64_t num_rows, int64_t num_cols, int64_t num_diagonals, int64_t pitch, const int32_t *offsets,
                       const float *values, int32_t *Ap, int32_t *Aj, float *Ax, int64_t capacity,
                       int64_t *num_entries_host, void *stream);
/* HYB -> CSR, same protocol: a row's ELL entries, then its COO entries (the COO part row-sorted, as hyb_matrix keeps it;
 * CMI_ERROR_NOT_SUPPORTED if it is not -- convert on the host then). */
int cmi_hyb_to_csr_f64(int64_t num_rows, int64_t ell_width, int64_t ell_pitch, const int32_t *ell_Aj, const double *ell_Ax,
                       int64_t coo_entries, const int32_t *coo_Ai, const int32_t *coo_Aj, const double *coo_Ax,
                       int32_t *Ap, int32_t *Aj, double *Ax, int64_t capacity, int64_t *num_entries_host, void *stream);
int cmi_hyb_to_csr_f32(int64_t num_rows, int64_t ell_width, int64_t ell_pitch, const int32_t *ell_Aj, const float *ell_Ax,
                       int64_t coo_entries, const int32_t *coo_Ai, const int32_t *coo_Aj, const float *coo_Ax,
                       int32_t *Ap, int32_t *Aj, float *Ax, int64_t capacity, int64_t *num_entries_host, void *stream);
/* ELL -> per-row length of the leading valid run (cusp/ktt/detail/ellr_matrix.inl:16-53). */
int cmi_ell_row_lengths(int64_t num_rows, int64_t width, int64_t pitch, const int32_t *ell_Aj,
                        int32_t *row_lengths, void *stream);

/* ------------------------------------------------------------------------- */
/* BLAS-1 on device vectors (f64 and f32): the routines cusp::krylov::cg calls */
/* (cusp/krylov/detail/cg.inl:63-105; generic/blas.h:175-220,283-340).          */
/* dot / nrm2 write their scalar to a DEVICE double (*result_dev) without a     */
/* host sync; the caller copies it back when it needs the value.                */
/* `workspace` is a device buffer of cmi_blas_workspace_bytes() bytes.          */
/* ------------------------------------------------------------------------- */
size_t cmi_blas_workspace_bytes(void);
int cmi_blas_axpy_f64(int64_t n, double alpha, const double *x, double *y, void *stream);  /* y += a x */
int cmi_blas_axpy_f32(int64_t n, float alpha, const float *x, float *y, void *stream);
int cmi_blas_axpby_f64(int64_t n, double alpha, const double *x, double beta, const double *y,
                       double *z, void *stream);                                             /* z = a x + b y */
int cmi_blas_axpby_f32(int64_t n, float alpha, const float *x, float beta, const float *y, float *z,
                       void *stream);
int cmi_blas_copy_f64(int64_t n, const double *x, double *y, void *stream);
int cmi_blas_copy_f32(int64_t n, const float *x, float *y, void *stream);
int cmi_blas_fill_f64(int64_t n, double value, double *y, void *stream);
int cmi_blas_fill_f32(int64_t n, float value, float *y, void *stream);
/* f32: products and partial sums are accumulated in double, the result is rounded to float once */
int cmi_blas_dot_f64(int64_t n, const double *x, const double *y, double *result_dev, void *workspace,
                     void *stream);
int cmi_blas_dot_f32(int64_t n, const float *x, const float *y, float *result_dev, void *workspace,
                     void *stream);
int cmi_blas_nrm2_f64(int64_t n, const double *x, double *result_dev, void *workspace, void *stream);
int cmi_blas_nrm2_f32(int64_t n, const float *x, float *result_dev, void *workspace, void *stream);
/* The rest of the reference's BLAS-1 set (cusp/blas/blas.h scal, xmy, axpbypcz, nrm1, nrmmax, amax -> cusp/system/detail/generic/blas.h)
 * for the multiply's OTHER callers (a Jacobi-preconditioned cg, bicgstab, cr): x <- alpha x; z <- x .* y (z may alias x or y);
 * out <- alpha x + beta y + gamma z; *result_dev <- sum |x_i| (nrm1); *value_dev <- max |x_i| (nrmmax) and *index_dev <- the FIRST position
 * holding it (amax; either pointer may be NULL).  Reductions: deterministic (fixed grid and tree, no atomics), accumulated in double,
 * result in DEVICE memory, `workspace` as above.  An empty vector: 0 and position 0.                                                   */
int cmi_blas_scal_f64(int64_t n, double alpha, double *x, void *stream);
int cmi_blas_scal_f32(int64_t n, float alpha, float *x, void *stream);
int cmi_blas_xmy_f64(int64_t n, const double *x, const double *y, double *z, void *stream);
int cmi_blas_xmy_f32(int64_t n, const float *x, const float *y, float *z, void *stream);
int cmi_blas_axpbypcz_f64(int64_t n, double alpha, const double *x, double beta, const double *y, double gamma, const double *z, double *out, void *stream);
int cmi_blas_axpbypcz_f32(int64_t n, float alpha, const float *x, float beta, const float *y, float gamma, const float *z, float *out, void *stream);
int cmi_blas_asum_f64(int64_t n, const double *x, double *result_dev, void *workspace, void *stream);
int cmi_blas_asum_f32(int64_t n, const float *x, float *result_dev, void *workspace, void *stream);
int cmi_blas_amax_f64(int64_t n, const double *x, double *value_dev, int64_t *index_dev, void *workspace, void *stream);
int cmi_blas_amax_f32(int64_t n, const float *x, float *value_dev, int64_t *index_dev, void *workspace, void *stream);
/* Jacobi-preconditioned CG's two vector passes with the scalars in device memory (the twin of cmi_cg_update_* / cmi_cg_direction_x_* for
 * M = cusp::precond::diagonal; reference cusp/krylov/detail/cg.inl:83-103 with that M: 5 passes and 2 host reads).  dinv = 1 / diag(A).
 *   update:     alpha = *rz_dev / *yp_dev;  r <- r - alpha y;  *rz_new_dev <- <r, dinv .* r>;  *rr_dev <- <r, r> (also written to
 *               rr_host_mirror when that is not NULL: page-locked memory the device can write, for the monitor)
 *   direction:  alpha = *rz_old_dev / *yp_dev;  beta = *rz_new_dev / *rz_old_dev;  x <- x + alpha p;  p <- dinv .* r + beta p
 * Deterministic two-stage reductions in double; `workspace`: cmi_blas_workspace_bytes().                                                 */
int cmi_pcg_update_jacobi_f64(int64_t n, const double *rz_dev, const double *yp_dev, const double *y, double *r, const double *dinv, double *rz_new_dev,
                              double *rr_dev, double *rr_host_mirror, void *workspace, void *stream);
int cmi_pcg_update_jacobi_f32(int64_t n, const double *rz_dev, const double *yp_dev, const float *y, float *r, const float *dinv, double *rz_new_dev,
                              double *rr_dev, double *rr_host_mirror, void *workspace, void *stream);
int cmi_pcg_direction_x_jacobi_f64(int64_t n, const double *rz_new_dev, const double *rz_old_dev, const double *yp_dev, const double *r, const double *dinv,
                                   double *p, double *x, void *stream);
int cmi_pcg_direction_x_jacobi_f32(int64_t n, const double *rz_new_dev, const double *rz_old_dev, const double *yp_dev, const float *r, const float *dinv,
                                   float *p, float *x, void *stream);
/* BiCGstab's three vector passes with the scalars in device memory (reference cusp/krylov/detail/bicgstab.inl:78-125, identity preconditioner:
 * ~9 passes and 6 host reads around its two multiplies).  rho = <r*, r>, d1 = <r*, A p>, d2 = <A s, s>, d3 = <A s, A s> are device doubles:
 *   s:   alpha = rho / d1;  s <- r - alpha A p;  *ss_dev <- <s, s>  (+ host mirror when not NULL)
 *   xr:  omega = d2 / d3;  x <- x + alpha p + omega s;  r <- s - omega A s;  *rho_new_dev <- <r*, r>;  *rr_dev <- <r, r>  (+ host mirror)
 *   p:   beta = (rho_new / rho) (alpha / omega);  p <- r + beta (p - omega A p)
 * cmi_blas_axpy_ratio_*: y <- y + (*num_dev / *den_dev) x  (the half step x <- x + alpha p of the early exit).                                */
int cmi_bicgstab_s_f64(int64_t n, const double *rho_dev, const double *d1_dev, const double *r, const double *AMp, double *s, double *ss_dev, double *ss_host_mirror,
                       void *workspace, void *stream);
int cmi_bicgstab_s_f32(int64_t n, const double *rho_dev, const double *d1_dev, const float *r, const float *AMp, float *s, double *ss_dev, double *ss_host_mirror,
                       void *workspace, void *stream);
int cmi_bicgstab_xr_f64(int64_t n, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const double *p, const double *s,
                        const double *AMs, const double *r_star, double *x, double *r, double *rho_new_dev, double *rr_dev, double *rr_host_mirror, void *workspace,
                        void *stream);
int cmi_bicgstab_xr_f32(int64_t n, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const float *p, const float *s,
                        const float *AMs, const float *r_star, float *x, float *r, double *rho_new_dev, double *rr_dev, double *rr_host_mirror, void *workspace,
                        void *stream);
int cmi_bicgstab_p_f64(int64_t n, const double *rho_new_dev, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const double *r,
                       const double *AMp, double *p, void *stream);
int cmi_bicgstab_p_f32(int64_t n, const double *rho_new_dev, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const float *r,
                       const float *AMp, float *p, void *stream);
/* diag[i] <- the sum of row i's entries in column i (0 when none is stored); reciprocal != 0: 1 / that -- the Jacobi preconditioner's set-up on the
 * device (reference cusp/format_utils.h:184 extract_diagonal, precond/detail/diagonal.inl).  One pass over the column indices, no synchronisation.  */
int cmi_csr_diagonal_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax, double *diag, int reciprocal, void *stream);
int cmi_csr_diagonal_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax, float *diag, int reciprocal, void *stream);
/* Conjugate residuals' two vector passes with the scalars in device memory (reference cusp/krylov/detail/cr.inl:83-124, identity preconditioner:
 * 7 passes and 3 host reads around its multiply).  rz = <r, A r>, yy = <A p, A p> are device doubles:
 *   xr:  alpha = rz / yy;  x <- x + alpha p;  update_r != 0: r <- r - alpha y (y = A p) and *rr_dev <- <r, r> (+ host mirror)
 *   py:  beta = rz_new / rz;  p <- r + beta p;  y <- A r + beta y;  *yy_new_dev <- <y, y>                                                   */
int cmi_cr_xr_f64(int64_t n, const double *rz_dev, const double *yy_dev, const double *p, const double *y, double *x, double *r, int update_r, double *rr_dev,
                  double *rr_host_mirror, void *workspace, void *stream);
int cmi_cr_xr_f32(int64_t n, const double *rz_dev, const double *yy_dev, const float *p, const float *y, float *x, float *r, int update_r, double *rr_dev,
                  double *rr_host_mirror, void *workspace, void *stream);
int cmi_cr_py_f64(int64_t n, const double *rz_new_dev, const double *rz_dev, const double *r, const double *Ar, double *p, double *y, double *yy_new_dev, void *workspace,
                  void *stream);
int cmi_cr_py_f32(int64_t n, const double *rz_new_dev, const double *rz_dev, const float *r, const float *Ar, float *p, float *y, double *yy_new_dev, void *workspace,
                  void *stream);
/* One step of GMRES's modified Gram-Schmidt with the coefficient in device memory (reference gmres.inl:145-152: a dotc -- a host read -- and an
 * axpy per basis vector): w <- w - (*h_dev) v, then *out_dev <- <w, u> in the same pass (u = the next basis vector, or u = w: the norm's square
 * behind the last axpy).  h_dev == NULL: the dot alone.  Deterministic two-stage reduction in double.                                        */
int cmi_blas_axpy_dot_f64(int64_t n, const double *h_dev, const double *v, double *w, const double *u, double *out_dev, void *workspace, void *stream);
int cmi_blas_axpy_dot_f32(int64_t n, const double *h_dev, const float *v, float *w, const float *u, double *out_dev, void *workspace, void *stream);
int cmi_blas_axpy_ratio_f64(int64_t n, const double *num_dev, const double *den_dev, const double *x, double *y, void *stream);
int cmi_blas_axpy_ratio_f32(int64_t n, const double *num_dev, const double *den_dev, const float *x, float *y, void *stream);

/* Fused steps of unpreconditioned CG (identity M, so z == r), scalars taken from DEVICE memory:
 * replaces dotc -> host -> axpy -> axpy -> copy -> dotc -> host -> axpby of
 * cusp/krylov/detail/cg.inl:83-103 (seven vector passes, three host syncs per iteration) by
 *   cmi_cg_update:    alpha = *rz / *yp;  x += alpha p (skipped when x == NULL);  r -= alpha y;  *rr = <r, r>   (one pass)
 *   cmi_cg_direction: beta = *rr_new / *rr_old;  p = r + beta p                         (one pass)
 * The element arithmetic is the reference's (one multiply and one add per update, unfused).
 * rr_host_mirror (may be NULL): cmi_malloc_host memory that also receives <r, r> -- written by the
 * reduction's last workgroup, so the convergence check needs no copy; read it after an event
 * recorded behind this call has been synchronised. */
int cmi_cg_update_f64(int64_t n, const double *rz_dev, const double *yp_dev, const double *p, const double *y,
                      double *x, double *r, double *rr_dev, double *rr_host_mirror, void *workspace, void *stream);
int cmi_cg_direction_f64(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *r,
                         double *p, void *stream);
/* The 8-pass split of the same iteration: cmi_cg_update_f64 with x == NULL leaves x alone (r and <r,r> only; p is then
 * not read), and cmi_cg_direction_x applies x += alpha p (alpha = *rr_old / *yp, the value cmi_cg_update used) with the
 * OLD p before it forms p = r + beta p -- the direction pass holds p in registers anyway, so p is read once less per
 * iteration.  Same expressions, same bits as the two calls above. */
int cmi_cg_direction_x_f64(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *yp_dev,
                           const double *r, double *p, double *x, void *stream);
/* The same steps on float vectors.  The scalars stay DOUBLES in device memory (every reduction here accumulates in
 * double); alpha and beta are rounded to float once per kernel and the vectors are updated in float, as the reference's
 * float CG does (cg.inl with ValueType = float).  cmi_blas_dotd_f32: <x, y> of float vectors as such a double. */
int cmi_cg_update_f32(int64_t n, const double *rz_dev, const double *yp_dev, const float *p, const float *y,
                      float *x, float *r, double *rr_dev, double *rr_host_mirror, void *workspace, void *stream);
int cmi_cg_direction_f32(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const float *r,
                         float *p, void *stream);
int cmi_cg_direction_x_f32(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *yp_dev,
                           const float *r, float *p, float *x, void *stream);

/* Fold-ahead forms of the fused iteration (measured -1.4 % per iteration on the headline matrix; the header layer uses them  */
/* only with $CMI_CG_FOLD_AHEAD=1): the two reductions' folds ride at the FRONT of the kernels that need them, so an              */
/* iteration is three launches (SpMV + partials, update, direction) instead of five.  The first workgroups of the consumer   */
/* fold the partial list exactly as the fold kernel does (same tree: bit-identical scalars) and publish the sum; every other   */
/* workgroup requests its vectors, then polls for it (bounded; the producer leaves the slot "pending").  `workspace`:            */
/* cmi_blas_workspace_bytes() bytes, two fold areas -- <y,p> partials in the first, <r,r> partials in the second.                */
/*   cmi_spmv_csr_dot_plan_partials_*: y <- A x; *npartials per-tile partials of <y, w> left in the workspace (0: this plan's    */
/*       kernel cannot fuse the dot -- y is computed, run cmi_blas_dot_* and the plain steps)                                     */
/*   cmi_cg_update_fold_*: yp <- fold (stored to *yp_dev too); r <- r - (rz/yp) y; *npartials_rr partials of <r, r> left           */
/*   cmi_cg_direction_x_fold_*: rr_new <- fold (stored to *rr_new_dev and, if given, the page-locked *rr_host_mirror);              */
/*       x <- x + (rr_old/yp) p; p <- r + (rr_new/rr_old) p                                                                         */
int cmi_spmv_csr_dot_plan_partials_f64(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                                       const double *x, double *y, const double *w, void *workspace, int *npartials, void *stream);
int cmi_spmv_csr_dot_plan_partials_f32(const cmi_plan *plan, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                                       const float *x, float *y, const float *w, void *workspace, int *npartials, void *stream);
int cmi_cg_update_fold_f64(int64_t n, const double *rz_dev, double *yp_dev, int npartials_yp, const double *y, double *r,
                           void *workspace, int *npartials_rr, void *stream);
int cmi_cg_update_fold_f32(int64_t n, const double *rz_dev, double *yp_dev, int npartials_yp, const float *y, float *r,
                           void *workspace, int *npartials_rr, void *stream);
int cmi_cg_direction_x_fold_f64(int64_t n, double *rr_new_dev, double *rr_host_mirror, int npartials_rr, const double *rr_old_dev,
                                const double *yp_dev, const double *r, double *p, double *x, void *workspace, void *stream);
int cmi_cg_direction_x_fold_f32(int64_t n, double *rr_new_dev, double *rr_host_mirror, int npartials_rr, const double *rr_old_dev,
                                const double *yp_dev, const float *r, float *p, float *x, void *workspace, void *stream);
int cmi_blas_dotd_f32(int64_t n, const float *x, const float *y, double *result_dev, void *workspace, void *stream);

/* ------------------------------------------------------------------------- */
/* Multi-GPU: communicator and collectives of the row-block sharded SpMV / CG */
/* (cmi_allgather_f64, cmi_allreduce_f64; 8(e): one process per   */
/* GPU, matrices sharded by row blocks with GLOBAL column indices, an all-gather of x over   */
/* xGMI before each multiply, all-reduced <.,.> scalars inside cusp::krylov::cg).  The          */
/* reference has no distributed code (cusp/ktt/detail/ktt.inl:34-35 pins device 0; the caller */
/* this serves is cusp/krylov/detail/cg.inl:41-107).  Backed by RCCL, bound at run time at the  */
/* first cmi_comm_* call (librccl.so.1; $CMI_RCCL_LIBRARY overrides): single-GPU users never    */
/* load it.  One process per GPU: call cmi_set_device first; every rank calls cmi_comm_create     */
/* with the SAME 128-byte id, which rank 0 makes with cmi_comm_unique_id and hands to the others   */
/* out of band (the header layer's cusp::distributed::communicator does it over a TCP socket to     */
/* MASTER_ADDR; a Python host may use any store).  Collectives are ENQUEUED on the caller's stream  */
/* and return at once; buffers are device memory; every rank must issue the same sequence.           */
/* ------------------------------------------------------------------------- */
#define CMI_COMM_ID_BYTES 128
typedef struct cmi_comm cmi_comm;
typedef enum cmi_reduce_op { CMI_OP_SUM = 0, CMI_OP_MAX = 1, CMI_OP_MIN = 2 } cmi_reduce_op;
int cmi_comm_unique_id(void *id_out /* CMI_COMM_ID_BYTES */);
int cmi_comm_create(const void *unique_id, int rank, int world, cmi_comm **comm); /* collective; synchronises */
int cmi_comm_destroy(cmi_comm *comm);
int cmi_comm_rank(const cmi_comm *comm, int *rank, int *world);
int cmi_comm_library_version(int *version); /* ncclGetVersion of the RCCL that was bound */
/* recv[r * count, (r + 1) * count) <- rank r's send[0, count): the north-star exchange (ncclAllGather).  In place when      */
/* send == recv + rank * count: the sharded operator keeps every rank's slice INSIDE the full-length x buffer, so nothing is   */
/* copied before or after.                                                                                                       */
int cmi_allgather_f64(cmi_comm *comm, const double *send, double *recv, int64_t count, void *stream);
int cmi_allgather_f32(cmi_comm *comm, const float *send, float *recv, int64_t count, void *stream);
/* Row blocks of different lengths (balanced by entries): recv[displs[r], displs[r] + counts[r]) <- rank r's         */
/* send[0, counts[r]); counts / displs are HOST arrays of `world` element counts.  algorithm 0: one ncclBroadcast per rank in ONE   */
/* group; 1: direct exchange, a grouped ncclSend / ncclRecv pair per peer -- on the point-to-point xGMI mesh every link then carries */
/* one peer's piece each way, the fabric's lower bound for an all-gather.  Same result either way.                                    */
int cmi_allgatherv_f64(cmi_comm *comm, const double *send, double *recv, const int64_t *counts, const int64_t *displs, int algorithm,
                       void *stream);
int cmi_allgatherv_f32(cmi_comm *comm, const float *send, float *recv, const int64_t *counts, const int64_t *displs, int algorithm,
                       void *stream);
/* Two-sided halo exchange inside ONE full-length buffer indexed by global column (banded matrices: 5-point Poisson needs 2 m     */
/* values per rank instead of N): for peer i, send x_full[send_lo[i], +send_count[i]) -- part of this rank's slice -- and receive     */
/* x_full[recv_lo[i], +recv_count[i]); one group = one launch.  HOST arrays of npeers entries.  The one-sided alternative is          */
/* cmi_ipc_* + cmi_copy_ranges above (a pull over xGMI, no collective).                                                               */
int cmi_halo_exchange_f64(cmi_comm *comm, double *x_full, int npeers, const int *peers, const int64_t *send_lo, const int64_t *send_count,
                          const int64_t *recv_lo, const int64_t *recv_count, void *stream);
int cmi_halo_exchange_f32(cmi_comm *comm, float *x_full, int npeers, const int *peers, const int64_t *send_lo, const int64_t *send_count,
                          const int64_t *recv_lo, const int64_t *recv_count, void *stream);
/* recv[i] <- op over the ranks of send[i]; device buffers, in place when send == recv.  CG's <y,p> and <r,r> (cg.inl:83,97): one or */
/* two doubles that never leave device memory.  Same inputs, same world size -> same bits on every rank and every run.               */
int cmi_allreduce_f64(cmi_comm *comm, const double *send, double *recv, int64_t count, int op, void *stream);
/* Every rank has reached this call and everything queued on `stream` before it has completed everywhere.  Synchronises.            */
int cmi_comm_barrier(cmi_comm *comm, void *stream);
/* Set-up convenience for small HOST records (IPC handles, column spans, row counts): recv_host[r * bytes, +bytes) <- rank r's          */
/* send_host[0, bytes); bytes x world <= ~4000.  Synchronises.                                                                      */
int cmi_comm_allgather_host(cmi_comm *comm, const void *send_host, void *recv_host, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUSP_MI355X_H */
