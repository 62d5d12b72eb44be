// cusp/convert.h -- cusp::convert(src, dst): any format / memory space -> any format / memory space
// (reference cusp/convert.h, cusp/system/detail/generic/convert.inl and conversions/*_to_other.h).
//
// Same-format conversions are plain array copies (H<->D included).  Format changes run on the host
// through CSR, producing exactly the layouts of the reference:
//   * -> CSR : COO is (stably) sorted by row when it is not already; DIA / ELL / dense walk
//              row-major and drop zeros / padding (conversions/dia_to_other.h:109-163)
//   CSR -> COO : row indices expanded from the offsets                 (csr_to_other.h:56-70)
//   CSR -> ELL : entry k of row i at k*pitch+i, padding -1 / 0, refuses a fill-in of > 3x and > 1e6
//                slots with format_conversion_exception               (csr_to_other.h:155-227)
//   CSR -> DIA : occupied diagonals ascending, same fill-in guard      (csr_to_other.h:73-153)
//   CSR -> HYB : ELL width from compute_optimal_entries_per_row(3.0, 4096), the rest in COO in CSR
//                order                              (csr_to_other.h:229-306, format_utils.inl:281-325)
// Device CSR -> device ELL / COO / HYB / DIA use the C-ABI's on-device builders (no host round trip), which is
// what makes 1e7-row format sweeps practical.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>

#include "detail/matrices.h"

namespace cusp {

// ---- format_utils (reference cusp/format_utils.h) on host arrays --------------------------------
template <typename OffsetArray, typename IndexArray>
void offsets_to_indices(const OffsetArray &offsets, IndexArray &indices)
{
    typedef typename IndexArray::value_type I;
    array1d<typename OffsetArray::value_type, host_memory> off(offsets);
    const size_t rows = off.size() ? off.size() - 1 : 0;
    array1d<I, host_memory> out(rows ? off[rows] : 0);
    for (size_t i = 0; i < rows; i++)
        for (I jj = off[i]; jj < off[i + 1]; jj++) out[jj] = static_cast<I>(i);
    indices = out;
}

template <typename IndexArray, typename OffsetArray>
void indices_to_offsets(const IndexArray &indices, OffsetArray &offsets)
{
    typedef typename OffsetArray::value_type I;
    array1d<typename IndexArray::value_type, host_memory> idx(indices);
    array1d<I, host_memory> off(offsets.size(), I(0));
    for (size_t n = 0; n < idx.size(); n++) off[idx[n] + 1]++;
    for (size_t i = 1; i < off.size(); i++) off[i] += off[i - 1];
    offsets = off;
}

template <typename OffsetArray> size_t compute_max_entries_per_row(const OffsetArray &row_offsets)
{
    array1d<typename OffsetArray::value_type, host_memory> off(row_offsets);
    size_t m = 0;
    for (size_t i = 0; i + 1 < off.size(); i++) m = std::max<size_t>(m, off[i + 1] - off[i]);
    return m;
}

// reference cusp/system/detail/generic/format_utils.inl:281-325 + cusp/detail/functional.inl:114-132
template <typename OffsetArray>
size_t compute_optimal_entries_per_row(const OffsetArray &row_offsets, float relative_speed = 3.0f, size_t breakeven_threshold = 4096)
{
    array1d<typename OffsetArray::value_type, host_memory> off(row_offsets);
    const size_t num_rows = off.size() ? off.size() - 1 : 0;
    const size_t max_len = compute_max_entries_per_row(off);
    std::vector<size_t> hist(max_len + 2, 0);
    for (size_t i = 0; i < num_rows; i++) hist[off[i + 1] - off[i]]++;
    size_t cum = 0;
    for (size_t k = 0; k < max_len; k++) {
        cum += hist[k]; // rows no longer than k
        const size_t longer = num_rows - cum;
        if (relative_speed * static_cast<float>(longer) < static_cast<float>(num_rows) || longer < breakeven_threshold) return k;
    }
    return max_len;
}

namespace detail {

template <typename I, typename V> using host_csr = csr_matrix<I, V, host_memory>;

// ---- anything -> host CSR ----------------------------------------------------------------------
template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, csr_format)
{
    out.resize(src.num_rows, src.num_cols, src.num_entries);
    out.row_offsets = src.row_offsets;
    out.column_indices = src.column_indices;
    out.values = src.values;
}

template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, coo_format)
{
    array1d<I, host_memory> ri(src.row_indices), ci(src.column_indices);
    array1d<V, host_memory> va(src.values);
    const size_t nnz = ri.size();
    bool sorted = true;
    for (size_t k = 1; k < nnz && sorted; k++) sorted = ri[k - 1] <= ri[k];
    out.resize(src.num_rows, src.num_cols, nnz);
    if (!sorted) { // stable sort by row, as coo_matrix::sort_by_row
        std::vector<size_t> perm(nnz);
        std::iota(perm.begin(), perm.end(), size_t(0));
        std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return ri[a] < ri[b]; });
        for (size_t k = 0; k < nnz; k++) { out.column_indices[k] = ci[perm[k]]; out.values[k] = va[perm[k]]; }
        array1d<I, host_memory> rs(nnz);
        for (size_t k = 0; k < nnz; k++) rs[k] = ri[perm[k]];
        indices_to_offsets(rs, out.row_offsets);
    } else {
        out.column_indices = ci;
        out.values = va;
        indices_to_offsets(ri, out.row_offsets);
    }
}

template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, ell_format)
{
    array1d<I, host_memory> cj(src.column_indices.values);
    array1d<V, host_memory> va(src.values.values);
    const size_t rows = src.num_rows, width = src.column_indices.num_cols, pitch = src.column_indices.pitch;
    size_t nnz = 0;
    for (size_t i = 0; i < rows; i++)
        for (size_t k = 0; k < width; k++) nnz += cj[k * pitch + i] != I(-1);
    out.resize(rows, src.num_cols, nnz);
    size_t p = 0;
    for (size_t i = 0; i < rows; i++) {
        out.row_offsets[i] = static_cast<I>(p);
        for (size_t k = 0; k < width; k++)
            if (cj[k * pitch + i] != I(-1)) { out.column_indices[p] = cj[k * pitch + i]; out.values[p] = va[k * pitch + i]; p++; }
    }
    out.row_offsets[rows] = static_cast<I>(p);
}

template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, dia_format)
{
    array1d<I, host_memory> off(src.diagonal_offsets);
    array1d<V, host_memory> va(src.values.values);
    const size_t rows = src.num_rows, nd = off.size(), pitch = src.values.pitch;
    const long long cols = static_cast<long long>(src.num_cols);
    auto keep = [&](size_t i, size_t d) {
        const long long j = static_cast<long long>(i) + off[d];
        return j >= 0 && j < cols && va[d * pitch + i] != V(0);
    };
    size_t nnz = 0;
    for (size_t i = 0; i < rows; i++)
        for (size_t d = 0; d < nd; d++) nnz += keep(i, d);
    out.resize(rows, src.num_cols, nnz);
    size_t p = 0;
    for (size_t i = 0; i < rows; i++) {
        out.row_offsets[i] = static_cast<I>(p);
        for (size_t d = 0; d < nd; d++)
            if (keep(i, d)) { out.column_indices[p] = static_cast<I>(i + off[d]); out.values[p] = va[d * pitch + i]; p++; }
    }
    out.row_offsets[rows] = static_cast<I>(p);
}

template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, hyb_format)
{
    // rows of the ELL part followed, per row, by that row's COO entries (the layout CSR -> HYB splits)
    host_csr<I, V> e;
    to_host_csr(src.ell, e, ell_format());
    coo_matrix<I, V, host_memory> c;
    c.resize(src.coo.num_rows, src.coo.num_cols, src.coo.num_entries);
    c.row_indices = src.coo.row_indices; c.column_indices = src.coo.column_indices; c.values = src.coo.values;
    host_csr<I, V> cc;
    to_host_csr(c, cc, coo_format());
    const size_t rows = src.num_rows;
    out.resize(rows, src.num_cols, e.num_entries + cc.num_entries);
    size_t p = 0;
    for (size_t i = 0; i < rows; i++) {
        out.row_offsets[i] = static_cast<I>(p);
        for (I jj = e.row_offsets[i]; jj < e.row_offsets[i + 1]; jj++) { out.column_indices[p] = e.column_indices[jj]; out.values[p] = e.values[jj]; p++; }
        if (cc.num_rows)
            for (I jj = cc.row_offsets[i]; jj < cc.row_offsets[i + 1]; jj++) { out.column_indices[p] = cc.column_indices[jj]; out.values[p] = cc.values[jj]; p++; }
    }
    out.row_offsets[rows] = static_cast<I>(p);
}

template <typename Src, typename I, typename V> void to_host_csr(const Src &src, host_csr<I, V> &out, array2d_format)
{
    typedef typename Src::value_type SV;
    array2d<SV, host_memory, typename Src::orientation> h(src);
    size_t nnz = 0;
    for (size_t i = 0; i < h.num_rows; i++)
        for (size_t j = 0; j < h.num_cols; j++) nnz += h(i, j) != SV(0);
    out.resize(h.num_rows, h.num_cols, nnz);
    size_t p = 0;
    for (size_t i = 0; i < h.num_rows; i++) {
        out.row_offsets[i] = static_cast<I>(p);
        for (size_t j = 0; j < h.num_cols; j++)
            if (h(i, j) != SV(0)) { out.column_indices[p] = static_cast<I>(j); out.values[p] = static_cast<V>(h(i, j)); p++; }
    }
    out.row_offsets[h.num_rows] = static_cast<I>(p);
}

// ---- host CSR -> anything ----------------------------------------------------------------------
inline void check_fill(const char *what, size_t slots, size_t entries)
{
    // reference csr_to_other.h:97-103,178-184: max_fill 3.0, threshold 1e6 (float arithmetic)
    const float size = static_cast<float>(slots);
    const float fill_ratio = size / std::max(1.0f, static_cast<float>(entries));
    if (3.0f < fill_ratio && size > 1e6f) throw cusp::format_conversion_exception(std::string(what) + " fill-in would exceed maximum tolerance");
}

template <typename I, typename V, typename Dst> void from_host_csr(const host_csr<I, V> &csr, Dst &dst, csr_format)
{
    dst.resize(csr.num_rows, csr.num_cols, csr.num_entries);
    dst.row_offsets = csr.row_offsets;
    dst.column_indices = csr.column_indices;
    dst.values = csr.values;
}

template <typename I, typename V, typename Dst> void from_host_csr(const host_csr<I, V> &csr, Dst &dst, coo_format)
{
    dst.resize(csr.num_rows, csr.num_cols, csr.num_entries);
    if (csr.num_entries == 0) return;
    array1d<I, host_memory> ri;
    offsets_to_indices(csr.row_offsets, ri);
    dst.row_indices = ri;
    dst.column_indices = csr.column_indices;
    dst.values = csr.values;
}

template <typename I, typename V, typename IA, typename VA>
void fill_ell_host(const host_csr<I, V> &csr, size_t width, size_t pitch, IA &cj, VA &va)
{
    std::fill(cj.begin(), cj.end(), I(-1));
    std::fill(va.begin(), va.end(), V(0));
    for (size_t i = 0; i < csr.num_rows; i++)
        for (I jj = csr.row_offsets[i]; jj < csr.row_offsets[i + 1]; jj++) {
            const size_t k = jj - csr.row_offsets[i];
            if (k < width) { cj[k * pitch + i] = csr.column_indices[jj]; va[k * pitch + i] = csr.values[jj]; }
        }
}

template <typename I, typename V, typename Dst>
void from_host_csr(const host_csr<I, V> &csr, Dst &dst, ell_format, size_t num_entries_per_row = 0, size_t alignment = 32)
{
    if (csr.num_entries == 0) { dst.resize(csr.num_rows, csr.num_cols, 0, num_entries_per_row, alignment); return; }
    if (num_entries_per_row == 0) {
        const size_t max_len = compute_max_entries_per_row(csr.row_offsets);
        check_fill("ell_matrix", max_len * csr.num_rows, csr.num_entries);
        num_entries_per_row = max_len;
    }
    size_t zeros = 0; // reference :188: num_entries excludes explicit zeros
    for (size_t k = 0; k < csr.num_entries; k++) zeros += csr.values[k] == V(0);
    dst.resize(csr.num_rows, csr.num_cols, csr.num_entries - zeros, num_entries_per_row, alignment);
    const size_t pitch = dst.column_indices.pitch;
    array1d<I, host_memory> cj(pitch * num_entries_per_row);
    array1d<V, host_memory> va(pitch * num_entries_per_row);
    fill_ell_host(csr, num_entries_per_row, pitch, cj, va);
    dst.column_indices.values = cj;
    dst.values.values = va;
}

template <typename I, typename V, typename Dst>
void from_host_csr(const host_csr<I, V> &csr, Dst &dst, dia_format, size_t alignment = 32)
{
    if (csr.num_entries == 0) { dst.resize(csr.num_rows, csr.num_cols, 0, 0); return; }
    std::vector<char> occupied(csr.num_rows + csr.num_cols, 0);
    for (size_t i = 0; i < csr.num_rows; i++)
        for (I jj = csr.row_offsets[i]; jj < csr.row_offsets[i + 1]; jj++) occupied[csr.column_indices[jj] - static_cast<long long>(i) + csr.num_rows] = 1;
    std::vector<long long> diag;
    std::vector<int> slot(occupied.size(), -1);
    for (size_t s = 0; s < occupied.size(); s++)
        if (occupied[s]) { slot[s] = static_cast<int>(diag.size()); diag.push_back(static_cast<long long>(s) - static_cast<long long>(csr.num_rows)); }
    check_fill("dia_matrix", diag.size() * csr.num_rows, csr.num_entries);
    dst.resize(csr.num_rows, csr.num_cols, csr.num_entries, diag.size(), alignment);
    const size_t pitch = dst.values.pitch;
    array1d<I, host_memory> off(diag.size());
    for (size_t d = 0; d < diag.size(); d++) off[d] = static_cast<I>(diag[d]);
    array1d<V, host_memory> va(pitch * diag.size(), V(0));
    for (size_t i = 0; i < csr.num_rows; i++)
        for (I jj = csr.row_offsets[i]; jj < csr.row_offsets[i + 1]; jj++)
            va[static_cast<size_t>(slot[csr.column_indices[jj] - static_cast<long long>(i) + csr.num_rows]) * pitch + i] = csr.values[jj];
    dst.diagonal_offsets = off;
    dst.values.values = va;
}

template <typename I, typename V, typename Dst>
void from_host_csr(const host_csr<I, V> &csr, Dst &dst, hyb_format, size_t num_entries_per_row = std::numeric_limits<size_t>::max(), size_t alignment = 32)
{
    if (csr.num_entries == 0) { dst.resize(csr.num_rows, csr.num_cols, 0, 0, num_entries_per_row == std::numeric_limits<size_t>::max() ? 0 : num_entries_per_row); return; }
    if (num_entries_per_row == std::numeric_limits<size_t>::max()) num_entries_per_row = compute_optimal_entries_per_row(csr.row_offsets, 3.0f, 4096);
    size_t n_coo = 0;
    for (size_t i = 0; i < csr.num_rows; i++) {
        const size_t len = csr.row_offsets[i + 1] - csr.row_offsets[i];
        if (len > num_entries_per_row) n_coo += len - num_entries_per_row;
    }
    dst.resize(csr.num_rows, csr.num_cols, csr.num_entries - n_coo, n_coo, num_entries_per_row, alignment);
    const size_t pitch = dst.ell.column_indices.pitch;
    array1d<I, host_memory> cj(pitch * num_entries_per_row);
    array1d<V, host_memory> va(pitch * num_entries_per_row);
    fill_ell_host(csr, num_entries_per_row, pitch, cj, va);
    dst.ell.column_indices.values = cj;
    dst.ell.values.values = va;
    array1d<I, host_memory> ri(n_coo), ci(n_coo);
    array1d<V, host_memory> cv(n_coo);
    size_t p = 0;
    for (size_t i = 0; i < csr.num_rows; i++)
        for (I jj = csr.row_offsets[i]; jj < csr.row_offsets[i + 1]; jj++)
            if (static_cast<size_t>(jj - csr.row_offsets[i]) >= num_entries_per_row) { ri[p] = static_cast<I>(i); ci[p] = csr.column_indices[jj]; cv[p] = csr.values[jj]; p++; }
    dst.coo.row_indices = ri; dst.coo.column_indices = ci; dst.coo.values = cv;
}

template <typename I, typename V, typename Dst> void from_host_csr(const host_csr<I, V> &csr, Dst &dst, array2d_format)
{
    typedef typename Dst::value_type DV;
    array2d<DV, host_memory, typename Dst::orientation> h(csr.num_rows, csr.num_cols, DV(0));
    for (size_t i = 0; i < csr.num_rows; i++)
        for (I jj = csr.row_offsets[i]; jj < csr.row_offsets[i + 1]; jj++) h(i, csr.column_indices[jj]) = static_cast<DV>(csr.values[jj]);
    dst = Dst(h);
}

// ---- same format: plain array copies -------------------------------------------------------------
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, csr_format)
{
    d.resize(s.num_rows, s.num_cols, s.num_entries);
    d.row_offsets = s.row_offsets; d.column_indices = s.column_indices; d.values = s.values;
}
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, coo_format)
{
    d.resize(s.num_rows, s.num_cols, s.num_entries);
    d.row_indices = s.row_indices; d.column_indices = s.column_indices; d.values = s.values;
}
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, ell_format)
{
    d.num_rows = s.num_rows; d.num_cols = s.num_cols; d.num_entries = s.num_entries;
    d.column_indices = typename Dst::column_indices_array_type(s.column_indices);
    d.values = typename Dst::values_array_type(s.values);
}
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, dia_format)
{
    d.num_rows = s.num_rows; d.num_cols = s.num_cols; d.num_entries = s.num_entries;
    d.diagonal_offsets = s.diagonal_offsets;
    d.values = typename Dst::values_array_type(s.values);
}
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, hyb_format)
{
    d.num_rows = s.num_rows; d.num_cols = s.num_cols; d.num_entries = s.num_entries;
    copy_same(s.ell, d.ell, ell_format());
    copy_same(s.coo, d.coo, coo_format());
}
template <typename Src, typename Dst> void copy_same(const Src &s, Dst &d, array2d_format)
{
    typedef typename Dst::value_type DV;
    if (std::is_same<typename Src::orientation, typename Dst::orientation>::value) {
        d.num_rows = s.num_rows; d.num_cols = s.num_cols; d.num_entries = s.num_entries; d.pitch = s.pitch;
        d.values = s.values;
    } else {
        array2d<typename Src::value_type, host_memory, typename Src::orientation> hs(s);
        array2d<DV, host_memory, typename Dst::orientation> hd(s.num_rows, s.num_cols);
        for (size_t i = 0; i < s.num_rows; i++)
            for (size_t j = 0; j < s.num_cols; j++) hd(i, j) = static_cast<DV>(hs(i, j));
        d = Dst(hd);
    }
}

// ---- device fast paths: device CSR -> device ELL / COO through the C-ABI builders ----------------
inline int csr_to_ell_device(int64_t rows, const int *Ap, const int *Aj, const double *Ax, int64_t w, int64_t pitch, int *eAj, double *eAx)
{ return cmi_csr_to_ell_f64(rows, Ap, Aj, Ax, w, pitch, eAj, eAx, nullptr); }
inline int csr_to_ell_device(int64_t rows, const int *Ap, const int *Aj, const float *Ax, int64_t w, int64_t pitch, int *eAj, float *eAx)
{ return cmi_csr_to_ell_f32(rows, Ap, Aj, Ax, w, pitch, eAj, eAx, nullptr); }

inline int count_zeros_device(int64_t n, const double *v, int64_t *c) { return cmi_count_zeros_f64(n, v, c, nullptr); }
inline int count_zeros_device(int64_t n, const float *v, int64_t *c) { return cmi_count_zeros_f32(n, v, c, nullptr); }

template <typename Src, typename Dst, typename SF, typename DF> struct device_fast_path {
    static bool run(const Src &, Dst &) { return false; }
};
template <typename V>
struct device_fast_path<csr_matrix<int, V, device_memory>, ell_matrix<int, V, device_memory>, csr_format, ell_format> {
    static bool run(const csr_matrix<int, V, device_memory> &s, ell_matrix<int, V, device_memory> &d)
    {
        if (s.num_entries == 0) return false;
        const size_t width = compute_max_entries_per_row(s.row_offsets); // one D2H copy of the offsets
        check_fill("ell_matrix", width * s.num_rows, s.num_entries);
        int64_t zeros = 0; // reference csr_to_other.h:188: the ELL matrix reports num_entries without the explicit zeros
        check(count_zeros_device(s.num_entries, s.values.data(), &zeros));
        d.resize(s.num_rows, s.num_cols, s.num_entries - static_cast<size_t>(zeros), width);
        check(csr_to_ell_device(s.num_rows, s.row_offsets.data(), s.column_indices.data(), s.values.data(), width,
                                d.column_indices.pitch, d.column_indices.values.data(), d.values.values.data()));
        check(cmi_stream_synchronize(nullptr));
        return true;
    }
};
template <typename V>
struct device_fast_path<csr_matrix<int, V, device_memory>, coo_matrix<int, V, device_memory>, csr_format, coo_format> {
    static bool run(const csr_matrix<int, V, device_memory> &s, coo_matrix<int, V, device_memory> &d)
    {
        d.resize(s.num_rows, s.num_cols, s.num_entries);
        if (s.num_entries == 0) return true;
        check(cmi_csr_row_indices(s.num_rows, s.row_offsets.data(), d.row_indices.data(), nullptr));
        d.column_indices = s.column_indices;
        d.values = s.values;
        check(cmi_stream_synchronize(nullptr));
        return true;
    }
};

// device COO (row-sorted, the container's contract: cusp/coo_matrix.h:72) -> device CSR: the offsets from the row indices
// in one pass with the order checked on the way; unsorted entries are sorted on the device first (a copy)
template <typename V>
struct device_fast_path<coo_matrix<int, V, device_memory>, csr_matrix<int, V, device_memory>, coo_format, csr_format> {
    static bool run(const coo_matrix<int, V, device_memory> &s, csr_matrix<int, V, device_memory> &d)
    {
        array1d<int, device_memory> offsets(s.num_rows + 1);
        int sorted = 0;
        check(cmi_coo_row_offsets(s.num_rows, s.num_entries, s.row_indices.data(), offsets.data(), &sorted, nullptr));
        if (!sorted) { // any order: a copy, sorted by row on the device (stable, coo_matrix::sort_by_row), then its offsets
            coo_matrix<int, V, device_memory> t(s);
            try { t.sort_by_row(); } catch (const std::exception &) { return false; } // (a row index outside the matrix: the general path's to deal with)
            check(cmi_coo_row_offsets(t.num_rows, t.num_entries, t.row_indices.data(), offsets.data(), &sorted, nullptr));
            if (!sorted) return false;
            d.resize(t.num_rows, t.num_cols, t.num_entries);
            d.row_offsets = offsets;
            d.column_indices.swap(t.column_indices);
            d.values.swap(t.values);
            return true;
        }
        d.resize(s.num_rows, s.num_cols, s.num_entries);
        d.row_offsets = offsets;
        d.column_indices = s.column_indices;
        d.values = s.values;
        check(cmi_stream_synchronize(nullptr));
        return true;
    }
};

inline int ell_to_csr_device(int64_t rows, int64_t w, int64_t pitch, const int *eAj, const double *eAx, int *Ap, int *Aj, double *Ax, int64_t cap, int64_t *n)
{ return cmi_ell_to_csr_f64(rows, w, pitch, eAj, eAx, Ap, Aj, Ax, cap, n, nullptr); }
inline int ell_to_csr_device(int64_t rows, int64_t w, int64_t pitch, const int *eAj, const float *eAx, int *Ap, int *Aj, float *Ax, int64_t cap, int64_t *n)
{ return cmi_ell_to_csr_f32(rows, w, pitch, eAj, eAx, Ap, Aj, Ax, cap, n, nullptr); }
inline int dia_to_csr_device(int64_t rows, int64_t cols, int64_t nd, int64_t pitch, const int *off, const double *va, int *Ap, int *Aj, double *Ax, int64_t cap, int64_t *n)
{ return cmi_dia_to_csr_f64(rows, cols, nd, pitch, off, va, Ap, Aj, Ax, cap, n, nullptr); }
inline int dia_to_csr_device(int64_t rows, int64_t cols, int64_t nd, int64_t pitch, const int *off, const float *va, int *Ap, int *Aj, float *Ax, int64_t cap, int64_t *n)
{ return cmi_dia_to_csr_f32(rows, cols, nd, pitch, off, va, Ap, Aj, Ax, cap, n, nullptr); }

// device ELL / DIA -> device CSR: count per row, exclusive scan, scatter (two calls: sizes first, then the arrays)
template <typename V>
struct device_fast_path<ell_matrix<int, V, device_memory>, csr_matrix<int, V, device_memory>, ell_format, csr_format> {
    static bool run(const ell_matrix<int, V, device_memory> &s, csr_matrix<int, V, device_memory> &d)
    {
        array1d<int, device_memory> offsets(s.num_rows + 1);
        int64_t n = 0;
        const int64_t w = s.column_indices.num_cols, pitch = s.column_indices.pitch;
        check(ell_to_csr_device(s.num_rows, w, pitch, s.column_indices.values.data(), s.values.values.data(), offsets.data(), nullptr,
                                static_cast<V *>(nullptr), 0, &n));
        d.resize(s.num_rows, s.num_cols, static_cast<size_t>(n));
        if (n > 0)
            check(ell_to_csr_device(s.num_rows, w, pitch, s.column_indices.values.data(), s.values.values.data(), d.row_offsets.data(),
                                    d.column_indices.data(), d.values.data(), n, &n));
        else
            d.row_offsets = offsets;
        return true;
    }
};
template <typename V>
struct device_fast_path<dia_matrix<int, V, device_memory>, csr_matrix<int, V, device_memory>, dia_format, csr_format> {
    static bool run(const dia_matrix<int, V, device_memory> &s, csr_matrix<int, V, device_memory> &d)
    {
        array1d<int, device_memory> offsets(s.num_rows + 1);
        int64_t n = 0;
        const int64_t nd = s.values.num_cols, pitch = s.values.pitch;
        check(dia_to_csr_device(s.num_rows, s.num_cols, nd, pitch, s.diagonal_offsets.data(), s.values.values.data(), offsets.data(), nullptr,
                                static_cast<V *>(nullptr), 0, &n));
        d.resize(s.num_rows, s.num_cols, static_cast<size_t>(n));
        if (n > 0)
            check(dia_to_csr_device(s.num_rows, s.num_cols, nd, pitch, s.diagonal_offsets.data(), s.values.values.data(), d.row_offsets.data(),
                                    d.column_indices.data(), d.values.data(), n, &n));
        else
            d.row_offsets = offsets;
        return true;
    }
};

inline int hyb_to_csr_device(int64_t rows, int64_t w, int64_t pitch, const int *eAj, const double *eAx, int64_t nc, const int *cAi, const int *cAj, const double *cAx, int *Ap, int *Aj, double *Ax, int64_t cap, int64_t *n)
{ return cmi_hyb_to_csr_f64(rows, w, pitch, eAj, eAx, nc, cAi, cAj, cAx, Ap, Aj, Ax, cap, n, nullptr); }
inline int hyb_to_csr_device(int64_t rows, int64_t w, int64_t pitch, const int *eAj, const float *eAx, int64_t nc, const int *cAi, const int *cAj, const float *cAx, int *Ap, int *Aj, float *Ax, int64_t cap, int64_t *n)
{ return cmi_hyb_to_csr_f32(rows, w, pitch, eAj, eAx, nc, cAi, cAj, cAx, Ap, Aj, Ax, cap, n, nullptr); }

// device HYB -> device CSR: a row's ELL entries, then its COO entries; a COO part that is not row-sorted takes the general path
template <typename V>
struct device_fast_path<hyb_matrix<int, V, device_memory>, csr_matrix<int, V, device_memory>, hyb_format, csr_format> {
    static bool run(const hyb_matrix<int, V, device_memory> &s, csr_matrix<int, V, device_memory> &d)
    {
        array1d<int, device_memory> offsets(s.num_rows + 1);
        int64_t n = 0;
        const int64_t w = s.ell.column_indices.num_cols, pitch = s.ell.column_indices.pitch;
        int st = hyb_to_csr_device(s.num_rows, w, pitch, s.ell.column_indices.values.data(), s.ell.values.values.data(), s.coo.num_entries,
                                   s.coo.row_indices.data(), s.coo.column_indices.data(), s.coo.values.data(), offsets.data(), nullptr,
                                   static_cast<V *>(nullptr), 0, &n);
        if (st == CMI_ERROR_NOT_SUPPORTED) return false;
        check(st);
        d.resize(s.num_rows, s.num_cols, static_cast<size_t>(n));
        if (n > 0)
            check(hyb_to_csr_device(s.num_rows, w, pitch, s.ell.column_indices.values.data(), s.ell.values.values.data(), s.coo.num_entries,
                                    s.coo.row_indices.data(), s.coo.column_indices.data(), s.coo.values.data(), d.row_offsets.data(),
                                    d.column_indices.data(), d.values.data(), n, &n));
        else
            d.row_offsets = offsets;
        return true;
    }
};

inline int csr_to_hyb_coo_device(int64_t rows, const int *Ap, const int *Aj, const double *Ax, int64_t w, const int *off, int *cAi, int *cAj, double *cAx)
{ return cmi_csr_to_hyb_coo_f64(rows, Ap, Aj, Ax, w, off, cAi, cAj, cAx, nullptr); }
inline int csr_to_hyb_coo_device(int64_t rows, const int *Ap, const int *Aj, const float *Ax, int64_t w, const int *off, int *cAi, int *cAj, float *cAx)
{ return cmi_csr_to_hyb_coo_f32(rows, Ap, Aj, Ax, w, off, cAi, cAj, cAx, nullptr); }

// device CSR -> device HYB: only the row offsets visit the host (width heuristic + overflow offsets)
template <typename V>
struct device_fast_path<csr_matrix<int, V, device_memory>, hyb_matrix<int, V, device_memory>, csr_format, hyb_format> {
    static bool run(const csr_matrix<int, V, device_memory> &s, hyb_matrix<int, V, device_memory> &d)
    {
        if (s.num_entries == 0) return false;
        array1d<int, host_memory> off(s.row_offsets);
        // the ELL width cutoff: the rule MEASURED on MI355X (tools/autotune_hyb.py, persisted in the tuning table: the width that
        // minimises the modelled time of the ELL + COO launches; cusp::ktt::reset_tuning() / no table: the reference's
        // compute_optimal_entries_per_row(3.0, 4096)).  Host conversions keep the reference's rule.
        int64_t tuned_width = 0;
        check(cmi_hyb_entries_per_row(detail::dtype_code<V>::value, s.num_rows, s.row_offsets.data(), -1, 0.0, 0, &tuned_width, nullptr));
        const size_t width = static_cast<size_t>(tuned_width);
        array1d<int, host_memory> coo_off(s.num_rows);
        size_t n_coo = 0;
        for (size_t i = 0; i < s.num_rows; i++) {
            coo_off[i] = static_cast<int>(n_coo);
            const size_t len = off[i + 1] - off[i];
            if (len > width) n_coo += len - width;
        }
        d.resize(s.num_rows, s.num_cols, s.num_entries - n_coo, n_coo, width);
        if (width > 0)
            check(csr_to_ell_device(s.num_rows, s.row_offsets.data(), s.column_indices.data(), s.values.data(), width,
                                    d.ell.column_indices.pitch, d.ell.column_indices.values.data(), d.ell.values.values.data()));
        if (n_coo) {
            array1d<int, device_memory> dev_off(coo_off);
            check(csr_to_hyb_coo_device(s.num_rows, s.row_offsets.data(), s.column_indices.data(), s.values.data(), width, dev_off.data(),
                                        d.coo.row_indices.data(), d.coo.column_indices.data(), d.coo.values.data()));
            check(cmi_stream_synchronize(nullptr));
        }
        check(cmi_stream_synchronize(nullptr));
        return true;
    }
};

inline int csr_to_dia_device(int64_t rows, int64_t cols, const int *Ap, const int *Aj, const double *Ax, int64_t nd, int64_t pitch, const int *off, int *map, double *va)
{ return cmi_csr_to_dia_f64(rows, cols, Ap, Aj, Ax, nd, pitch, off, map, va, nullptr); }
inline int csr_to_dia_device(int64_t rows, int64_t cols, const int *Ap, const int *Aj, const float *Ax, int64_t nd, int64_t pitch, const int *off, int *map, float *va)
{ return cmi_csr_to_dia_f32(rows, cols, Ap, Aj, Ax, nd, pitch, off, map, va, nullptr); }

// device CSR -> device DIA: only the (few) diagonal offsets visit the host, to be sorted ascending
template <typename V>
struct device_fast_path<csr_matrix<int, V, device_memory>, dia_matrix<int, V, device_memory>, csr_format, dia_format> {
    static bool run(const csr_matrix<int, V, device_memory> &s, dia_matrix<int, V, device_memory> &d)
    {
        if (s.num_entries == 0 || s.num_rows == 0) return false;
        // no more diagonals than the fill-in guard lets through (csr_to_other.h:97-103)
        size_t capacity = std::max<size_t>(static_cast<size_t>(3.0 * s.num_entries / s.num_rows) + 1, static_cast<size_t>(1e6) / s.num_rows + 1);
        capacity = std::min(capacity, s.num_rows + s.num_cols);
        array1d<int, device_memory> slot_map(s.num_rows + s.num_cols), list(capacity);
        int64_t nd = 0;
        check(cmi_csr_diagonals(s.num_rows, s.num_cols, s.row_offsets.data(), s.column_indices.data(), slot_map.data(), list.data(),
                                static_cast<int64_t>(capacity), &nd, nullptr));
        check_fill("dia_matrix", static_cast<size_t>(nd) * s.num_rows, s.num_entries);
        if (static_cast<size_t>(nd) > capacity) throw cusp::format_conversion_exception("dia_matrix fill-in would exceed maximum tolerance");
        array1d<int, host_memory> off(list);
        off.resize(static_cast<size_t>(nd));
        std::sort(off.begin(), off.end());
        d.resize(s.num_rows, s.num_cols, s.num_entries, static_cast<size_t>(nd));
        d.diagonal_offsets = off;
        check(csr_to_dia_device(s.num_rows, s.num_cols, s.row_offsets.data(), s.column_indices.data(), s.values.data(), nd, d.values.pitch,
                                d.diagonal_offsets.data(), slot_map.data(), d.values.values.data()));
        check(cmi_stream_synchronize(nullptr));
        return true;
    }
};

template <typename Src, typename Dst> void convert_impl(const Src &src, Dst &dst, std::true_type /*same format*/)
{
    copy_same(src, dst, typename Dst::format());
}
template <typename Src, typename Dst> void convert_impl(const Src &src, Dst &dst, std::false_type)
{
    if (device_fast_path<Src, Dst, typename Src::format, typename Dst::format>::run(src, dst)) return;
    typedef typename Dst::index_type I;
    typedef typename Dst::value_type V;
    // two device matrices, neither of them CSR (ELL -> DIA, COO -> HYB, ...): pivot through a CSR matrix that stays in HBM
    if constexpr (std::is_same<typename Src::memory_space, device_memory>::value && std::is_same<typename Dst::memory_space, device_memory>::value &&
                  std::is_same<typename Src::index_type, int>::value && std::is_same<I, int>::value &&
                  std::is_same<typename Src::value_type, V>::value && !std::is_same<typename Src::format, csr_format>::value &&
                  !std::is_same<typename Dst::format, csr_format>::value && !std::is_same<typename Src::format, array2d_format>::value &&
                  !std::is_same<typename Dst::format, array2d_format>::value) {
        csr_matrix<int, V, device_memory> pivot;
        if (device_fast_path<Src, csr_matrix<int, V, device_memory>, typename Src::format, csr_format>::run(src, pivot) &&
            device_fast_path<csr_matrix<int, V, device_memory>, Dst, csr_format, typename Dst::format>::run(pivot, dst))
            return;
    }
    host_csr<I, V> csr;
    to_host_csr(src, csr, typename Src::format());
    from_host_csr(csr, dst, typename Dst::format());
}

} // namespace detail

template <typename Src, typename Dst> void convert(const Src &src, Dst &dst)
{
    detail::convert_impl(src, dst, typename std::is_same<typename Src::format, typename Dst::format>::type());
}

// array2d(const SparseMatrix&) declared in cusp/array2d.h
template <typename T, typename M, typename O>
template <typename Matrix, typename, typename>
array2d<T, M, O>::array2d(const Matrix &m) : array2d()
{
    cusp::convert(m, *this);
}

namespace ktt {
template <typename I, typename V, typename M> void ellr_matrix<I, V, M>::compute_row_lengths()
{
    array1d<I, host_memory> cj(this->column_indices.values);
    const size_t rows = this->num_rows, width = this->column_indices.num_cols, pitch = this->column_indices.pitch;
    array1d<I, host_memory> len(rows);
    for (size_t i = 0; i < rows; i++) {
        I l = 0;
        while (static_cast<size_t>(l) < width && cj[l * pitch + i] >= 0) l++;
        len[i] = l;
    }
    row_lengths = len;
}
} // namespace ktt

} // namespace cusp
