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
//   COO : launch shape + "are the entries sorted by row?" -- sorted input runs the tile kernel (plain stores, storage-
//         order sums), anything else the order-agnostic atomics kernels.
//   ELL / DIA / HYB : the resolved launch shape only (nothing to measure).
#include "common.h"
#include <new>

using namespace cmi;

CMI_API int cmi_plan_create(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                            const int32_t *index_array, const cmi_config *cfg, void *stream, cmi_plan **plan_out)
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
    // HYB's table key is its ELL part's (the COO part looks its own shape up per call)
    select_config(format == CMI_FORMAT_HYB ? CMI_FORMAT_ELL : format, dtype, num_rows, num_cols, num_entries, cfg, &p->cfg);
    hipStream_t s = as_stream(stream);
    const size_t vbytes = dtype == CMI_F64 ? 8 : 4;
    int st = CMI_SUCCESS;
    if (format == CMI_FORMAT_CSR && num_rows > 0 && num_entries > 0) {
        st = measure_row_lengths(num_rows, index_array, s, &p->prof.max_len, &p->prof.in_long);
        const bool auto_kernel = !cfg || cfg->kernel == CMI_KERNEL_AUTO;
        if (st == CMI_SUCCESS && auto_kernel && prefers_balanced(num_rows, num_entries, p->prof, vbytes, p->cfg.threads_per_row == 1)) {
            p->cfg.kernel = CMI_CSR_BALANCED;
            p->cfg.items_per_thread = 0; // the table's row-tile launch shape does not apply: balanced defaults
            p->cfg.blocks_per_cu = 0;
            p->cfg.xcd_swizzle = 0;      // (chunks in launch order: a skewed matrix has no x window worth dealing for)
        }
    } else if (format == CMI_FORMAT_COO) {
        int sorted = 1;
        if (num_entries > 0) st = coo_rows_sorted(num_rows, num_entries, index_array, s, &sorted);
        if (st == CMI_SUCCESS) {
            p->coo_sorted = sorted;
            const bool auto_kernel = !cfg || cfg->kernel == CMI_KERNEL_AUTO;
            // sorted entries: the table's CMI_TABLE_COO_SORTED key (the tile kernel with its tuned cache policy / XCD dealing)
            // (fewer than four entries: the tile kernel's vector loads have nothing to read -- the order-agnostic kernel stays)
            if (sorted && auto_kernel && num_entries >= 4) select_config(CMI_TABLE_COO_SORTED, dtype, num_rows, num_cols, num_entries, cfg, &p->cfg);
            // an explicit CMI_COO_TILE on unsorted entries would add rows up wrongly: refuse it here, where it is known
            if (!sorted && p->cfg.kernel == CMI_COO_TILE) st = fail(CMI_ERROR_INVALID_VALUE, "cmi_plan_create: CMI_COO_TILE needs row-sorted entries");
        }
    }
    if (st != CMI_SUCCESS) { delete p; return st; }
    *plan_out = p;
    return CMI_SUCCESS;
}

CMI_API int cmi_plan_destroy(cmi_plan *plan)
{
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
            exact = c.kernel == CMI_CSR_SCALAR || c.kernel == CMI_CSR_STREAM_PIPE ||
                    (c.kernel == CMI_CSR_STREAM && c.threads_per_row <= 1 && (c.threads_per_row == 1 || plan->prof.max_len < 512));
            break;
        case CMI_FORMAT_ELL:
        case CMI_FORMAT_DIA: exact = 1; break;
        case CMI_FORMAT_COO: exact = c.kernel == CMI_COO_TILE; break;
        default: exact = 0; break; // HYB: its COO half accumulates with atomics unless planned separately
        }
        *storage_order_sums = exact;
    }
    return CMI_SUCCESS;
}
