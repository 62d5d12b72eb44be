// cusp/detail/matrices.h -- the five sparse containers and their views, with the public data members
// of the reference:
//   csr_matrix  (cusp/csr_matrix.h:107-208)   row_offsets, column_indices, values
//   coo_matrix  (cusp/coo_matrix.h:116-224)   row_indices, column_indices, values, sort_by_row*, is_sorted_*
//   ell_matrix  (cusp/ell_matrix.h:119-228)   column_indices, values (column-major array2d), invalid_index = -1
//   dia_matrix  (cusp/dia_matrix.h:120-226)   diagonal_offsets, values (column-major array2d)
//   hyb_matrix  (cusp/hyb_matrix.h:142-246)   ell, coo
//   base        (cusp/detail/matrix_base.h:29-68) num_rows, num_cols, num_entries + the typedefs
// plus cusp::ktt::ellr_matrix (cusp/ktt/ellr_matrix.h:17-90): ELL + row_lengths.
// Included through cusp/{csr,coo,ell,dia,hyb}_matrix.h, which also pull in cusp/convert.h so that the
// converting constructors / operator= (any format, any memory space) are complete.
#pragma once
#include <algorithm>
#include <memory>
#include <numeric>

#include "../array1d.h"
#include "../array2d.h"

namespace cusp {

template <typename Src, typename Dst> void convert(const Src &src, Dst &dst); // cusp/convert.h

namespace detail {

template <typename IndexType, typename ValueType, typename MemorySpace, typename Format> class matrix_base {
public:
    typedef IndexType index_type;
    typedef ValueType value_type;
    typedef MemorySpace memory_space;
    typedef Format format;

    size_t num_rows, num_cols, num_entries;

    matrix_base() : num_rows(0), num_cols(0), num_entries(0) {}
    matrix_base(size_t r, size_t c, size_t n) : num_rows(r), num_cols(c), num_entries(n) {}
    template <typename M> explicit matrix_base(const M &m) : num_rows(m.num_rows), num_cols(m.num_cols), num_entries(m.num_entries) {}
    void resize(size_t r, size_t c, size_t n) { num_rows = r; num_cols = c; num_entries = n; }
    void swap(matrix_base &o) { std::swap(num_rows, o.num_rows); std::swap(num_cols, o.num_cols); std::swap(num_entries, o.num_entries); }
};

// What the engine learnt about a device-resident CSR / COO matrix before its first multiply (cmi_plan, cusp_mi355x.h):
// launch shape, the CSR row-length profile, COO row-sortedness.  Owned by the CONTAINER (views have none and multiply
// through the plan-less entry points); made at the first cusp::multiply -- that one call synchronises the stream -- or
// ahead of time by A.plan(); re-made when the index array or the sizes change; `invalidate_plan()` after editing the
// structure -- row offsets OR column indices: a CSR plan is made from both and may hold a copy derived from the columns (the
// run-compressed pieces of CMI_CSR_STREAM_WAVER, round 4) -- in place.  Values are never cached: refresh them freely.  The reference
// has no such object (its KTT path recomputes `row_starts` on the host per call, cusp/system/cuda/ktt/csr_multiply.h:239-247).
struct plan_slot {
    std::shared_ptr<cmi_plan> plan;
    const void *index_ptr = nullptr, *columns_ptr = nullptr;
    size_t rows = 0, cols = 0, entries = 0;
    int compress = -1; // CSR: -1 follow cmi_set_index_compression / $CMI_COMPRESS_INDICES, 0 never, 1 ask for CMI_CSR_STREAM_C16
    void reset() { plan.reset(); index_ptr = nullptr; columns_ptr = nullptr; }
    // `columns`: CSR only -- with them the plan may hold the 16-bit column copy (cmi_plan_create_csr)
    const cmi_plan *get(int format, int dtype, size_t r, size_t c, size_t n, const int *index, void *stream, const int *columns = nullptr)
    {
        if (!plan || index_ptr != index || columns_ptr != columns || rows != r || cols != c || entries != n) {
            cmi_plan *p = nullptr;
            if (format == CMI_FORMAT_CSR && columns && compress != 0) {
                cmi_config want = {};
                want.kernel = CMI_CSR_STREAM_C16;
                check(cmi_plan_create_csr(dtype, (int64_t)r, (int64_t)c, (int64_t)n, index, columns, compress == 1 ? &want : nullptr, stream, &p));
            } else if (format == CMI_FORMAT_COO && columns) // (round 4: the sorted-COO plan's CSR sub-plan made with the columns too)
                check(cmi_plan_create_coo(dtype, (int64_t)r, (int64_t)c, (int64_t)n, index, columns, nullptr, stream, &p));
            else
                check(cmi_plan_create(format, dtype, (int64_t)r, (int64_t)c, (int64_t)n, index, nullptr, stream, &p));
            plan.reset(p, [](cmi_plan *q) { cmi_plan_destroy(q); });
            index_ptr = index; columns_ptr = columns; rows = r; cols = c; entries = n;
        }
        return plan.get();
    }
};
// The same for a HYB matrix (cmi_plan_create_hyb): keyed by the COO part's row indices; with that part sorted by row the
// plan holds the per-tile entry ranges of the one-launch HYB kernel.
struct hyb_plan_slot {
    std::shared_ptr<cmi_plan> plan;
    const void *index_ptr = nullptr;
    size_t rows = 0, cols = 0, width = 0, coo = 0;
    void reset() { plan.reset(); index_ptr = nullptr; }
    const cmi_plan *get(int dtype, size_t r, size_t c, size_t w, size_t n, const int *index, void *stream)
    {
        if (!plan || index_ptr != index || rows != r || cols != c || width != w || coo != n) {
            cmi_plan *p = nullptr;
            check(cmi_plan_create_hyb(dtype, (int64_t)r, (int64_t)c, (int64_t)w, (int64_t)n, index, nullptr, nullptr, stream, &p));
            plan.reset(p, [](cmi_plan *q) { cmi_plan_destroy(q); });
            index_ptr = index; rows = r; cols = c; width = w; coo = n;
        }
        return plan.get();
    }
};
template <typename V> struct dtype_code;
template <> struct dtype_code<double> { static const int value = CMI_F64; };
template <> struct dtype_code<float> { static const int value = CMI_F32; };

// true for cusp containers / views (anything with a format tag)
template <typename M, typename = void> struct has_format { static const bool value = false; };
template <typename M> struct has_format<M, typename std::enable_if<!std::is_void<typename M::format>::value>::type> { static const bool value = true; };

} // namespace detail

// ---------------------------------------------------------------------------------------------
// CSR
// ---------------------------------------------------------------------------------------------
template <typename I, typename V, typename M> class csr_matrix_view;

template <typename IndexType, typename ValueType, typename MemorySpace>
class csr_matrix : public detail::matrix_base<IndexType, ValueType, MemorySpace, csr_format> {
    typedef detail::matrix_base<IndexType, ValueType, MemorySpace, csr_format> Parent;
public:
    typedef array1d<IndexType, MemorySpace> row_offsets_array_type;
    typedef array1d<IndexType, MemorySpace> column_indices_array_type;
    typedef array1d<ValueType, MemorySpace> values_array_type;
    typedef csr_matrix<IndexType, ValueType, MemorySpace> container;
    typedef csr_matrix_view<IndexType, ValueType, MemorySpace> view;
    typedef csr_matrix_view<const IndexType, const ValueType, MemorySpace> const_view;
    template <typename Space> struct rebind { typedef csr_matrix<IndexType, ValueType, Space> type; };

    row_offsets_array_type row_offsets;
    column_indices_array_type column_indices;
    values_array_type values;

    csr_matrix() {}
    csr_matrix(size_t rows, size_t cols, size_t entries) : Parent(rows, cols, entries), row_offsets(rows + 1), column_indices(entries), values(entries) {}
    csr_matrix(const csr_matrix &) = default;
    csr_matrix(csr_matrix &&) = default;
    csr_matrix &operator=(const csr_matrix &) = default;
    csr_matrix &operator=(csr_matrix &&) = default;
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    csr_matrix(const Matrix &m) { cusp::convert(m, *this); }
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    csr_matrix &operator=(const Matrix &m) { cusp::convert(m, *this); return *this; }

    void resize(size_t rows, size_t cols, size_t entries)
    {
        Parent::resize(rows, cols, entries);
        row_offsets.resize(rows + 1);
        column_indices.resize(entries);
        values.resize(entries);
        plan_.reset(); // every writer of structure (convert, gallery, readers, operator=) resizes first: a device array1d keeps its
                       // buffer when the new size fits, so the plan's (pointer, sizes) key alone would survive a different matrix
    }
    void swap(csr_matrix &o) { Parent::swap(o); row_offsets.swap(o.row_offsets); column_indices.swap(o.column_indices); values.swap(o.values); std::swap(plan_, o.plan_); }

    // device_memory, int indices, float / double values: the engine's plan for this matrix (see detail::plan_slot)
    const cmi_plan *plan(void *stream = nullptr) const
    {
        return plan_.get(CMI_FORMAT_CSR, detail::dtype_code<ValueType>::value, this->num_rows, this->num_cols, this->num_entries,
                         reinterpret_cast<const int *>(row_offsets.data()), stream, reinterpret_cast<const int *>(column_indices.data()));
    }
    void invalidate_plan() const { plan_.reset(); }
    // Opt in to (or out of) the plan's 16-bit copy of the column indices for THIS matrix (CMI_CSR_STREAM_C16: 2 bytes per
    // entry of extra HBM, 10 instead of 12 bytes per entry read by every multiply, same bits); granted only when every row
    // tile spans < 65536 columns -- cmi_plan_config(A.plan()) tells.  Without a call the process-wide default applies.
    void compress_indices(bool on) const { plan_.compress = on ? 1 : 0; plan_.reset(); }
private:
    mutable detail::plan_slot plan_;
};

template <typename I, typename V, typename M>
class csr_matrix_view : public detail::matrix_base<typename std::remove_const<I>::type, typename std::remove_const<V>::type, M, csr_format> {
    typedef detail::matrix_base<typename std::remove_const<I>::type, typename std::remove_const<V>::type, M, csr_format> Parent;
public:
    typedef csr_matrix<typename Parent::index_type, typename Parent::value_type, M> container;
    typedef csr_matrix_view view;
    array1d_view<I, M> row_offsets, column_indices;
    array1d_view<V, M> values;

    csr_matrix_view() {}
    csr_matrix_view(size_t rows, size_t cols, size_t entries, array1d_view<I, M> ro, array1d_view<I, M> ci, array1d_view<V, M> v)
        : Parent(rows, cols, entries), row_offsets(ro), column_indices(ci), values(v) {}
    csr_matrix_view(container &m) : Parent(m), row_offsets(m.row_offsets), column_indices(m.column_indices), values(m.values) {}
    csr_matrix_view(const container &m) : Parent(m), row_offsets(m.row_offsets), column_indices(m.column_indices), values(m.values) {}
};

template <typename I, typename V, typename M>
csr_matrix_view<I, V, M> make_csr_matrix_view(size_t rows, size_t cols, size_t entries, array1d_view<I, M> ro, array1d_view<I, M> ci, array1d_view<V, M> v)
{
    return csr_matrix_view<I, V, M>(rows, cols, entries, ro, ci, v);
}
// reference cusp/csr_matrix.h make_csr_matrix_view(matrix | const matrix | view) (testing/csr_matrix_view.cu:95-194)
template <typename I, typename V, typename M> csr_matrix_view<I, V, M> make_csr_matrix_view(csr_matrix<I, V, M> &m) { return csr_matrix_view<I, V, M>(m); }
template <typename I, typename V, typename M> csr_matrix_view<const I, const V, M> make_csr_matrix_view(const csr_matrix<I, V, M> &m) { return csr_matrix_view<const I, const V, M>(m); }
template <typename I, typename V, typename M> csr_matrix_view<I, V, M> make_csr_matrix_view(const csr_matrix_view<I, V, M> &v) { return v; }

// ---------------------------------------------------------------------------------------------
// COO
// ---------------------------------------------------------------------------------------------
template <typename I, typename V, typename M> class coo_matrix_view;

template <typename IndexType, typename ValueType, typename MemorySpace>
class coo_matrix : public detail::matrix_base<IndexType, ValueType, MemorySpace, coo_format> {
    typedef detail::matrix_base<IndexType, ValueType, MemorySpace, coo_format> Parent;
public:
    typedef array1d<IndexType, MemorySpace> row_indices_array_type;
    typedef array1d<IndexType, MemorySpace> column_indices_array_type;
    typedef array1d<ValueType, MemorySpace> values_array_type;
    typedef coo_matrix container;
    typedef coo_matrix_view<IndexType, ValueType, MemorySpace> view;
    typedef coo_matrix_view<const IndexType, const ValueType, MemorySpace> const_view;
    template <typename Space> struct rebind { typedef coo_matrix<IndexType, ValueType, Space> type; };

    row_indices_array_type row_indices;
    column_indices_array_type column_indices;
    values_array_type values;

    coo_matrix() {}
    coo_matrix(size_t rows, size_t cols, size_t entries) : Parent(rows, cols, entries), row_indices(entries), column_indices(entries), values(entries) {}
    coo_matrix(const coo_matrix &) = default;
    coo_matrix(coo_matrix &&) = default;
    coo_matrix &operator=(const coo_matrix &) = default;
    coo_matrix &operator=(coo_matrix &&) = default;
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    coo_matrix(const Matrix &m) { cusp::convert(m, *this); }
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    coo_matrix &operator=(const Matrix &m) { cusp::convert(m, *this); return *this; }

    void resize(size_t rows, size_t cols, size_t entries)
    {
        Parent::resize(rows, cols, entries);
        row_indices.resize(entries);
        column_indices.resize(entries);
        values.resize(entries);
        plan_.reset(); // as csr_matrix::resize: the cached row offsets of a sorted-COO plan belong to the OLD row indices
    }
    void swap(coo_matrix &o) { Parent::swap(o); row_indices.swap(o.row_indices); column_indices.swap(o.column_indices); values.swap(o.values); std::swap(plan_, o.plan_); }

    // device_memory: the engine's plan (row-sortedness checked once: sorted entries run the tile kernel -- storage-order
    // sums, no atomics; see detail::plan_slot)
    const cmi_plan *plan(void *stream = nullptr) const
    {
        return plan_.get(CMI_FORMAT_COO, detail::dtype_code<ValueType>::value, this->num_rows, this->num_cols, this->num_entries,
                         reinterpret_cast<const int *>(row_indices.data()), stream, reinterpret_cast<const int *>(column_indices.data()));
    }
    void invalidate_plan() const { plan_.reset(); }

    // reference cusp/coo_matrix.h: sort_by_row / sort_by_row_and_column / is_sorted_by_row[_and_column] (stable, like the
    // reference's stable_sort_by_key).  device_memory with int indices and float / double values: on the device
    // (cmi_coo_sort_by_row_*, cmi_coo_is_sorted); everything else: on the host.
    void sort_by_row() { sort_impl(false); }
    void sort_by_row_and_column() { sort_impl(true); }
    bool is_sorted_by_row() const { return sorted_impl(false); }
    bool is_sorted_by_row_and_column() const { return sorted_impl(true); }

private:
    mutable detail::plan_slot plan_;
    static int device_sort(int64_t r, int64_t c, int64_t n, int *ai, int *aj, double *ax, int ac) { return cmi_coo_sort_by_row_f64(r, c, n, ai, aj, ax, ac, nullptr); }
    static int device_sort(int64_t r, int64_t c, int64_t n, int *ai, int *aj, float *ax, int ac) { return cmi_coo_sort_by_row_f32(r, c, n, ai, aj, ax, ac, nullptr); }
    template <typename A, typename B, typename C> static int device_sort(int64_t, int64_t, int64_t, A *, B *, C *, int) { return -1; } // no device sort for these types
    static const bool on_device = std::is_same<MemorySpace, device_memory>::value && std::is_same<IndexType, int>::value &&
                                  (std::is_same<ValueType, double>::value || std::is_same<ValueType, float>::value);
    void sort_impl(bool and_column)
    {
        plan_.reset();
        if (on_device) {
            detail::check(device_sort((int64_t)this->num_rows, (int64_t)this->num_cols, (int64_t)row_indices.size(), row_indices.data(),
                                      column_indices.data(), values.data(), and_column ? 1 : 0));
            return;
        }
        array1d<IndexType, host_memory> ri(row_indices), ci(column_indices);
        array1d<ValueType, host_memory> va(values);
        std::vector<size_t> perm(ri.size());
        bool in_range = true;
        for (size_t k = 0; k < ri.size() && in_range; k++) in_range = ri[k] >= IndexType(0) && static_cast<size_t>(ri[k]) < this->num_rows;
        if (in_range) { // stable counting sort by row (O(n)), then -- by (row, column) -- a stable sort of each row's few entries
            std::vector<size_t> start(this->num_rows + 1, 0);
            for (size_t k = 0; k < ri.size(); k++) start[static_cast<size_t>(ri[k]) + 1]++;
            for (size_t r = 0; r < this->num_rows; r++) start[r + 1] += start[r];
            std::vector<size_t> next(start.begin(), start.end() - 1);
            for (size_t k = 0; k < ri.size(); k++) perm[next[static_cast<size_t>(ri[k])]++] = k;
            if (and_column)
                for (size_t r = 0; r < this->num_rows; r++) {
                    auto lo = perm.begin() + static_cast<std::ptrdiff_t>(start[r]), hi = perm.begin() + static_cast<std::ptrdiff_t>(start[r + 1]);
                    if (hi - lo > 1 && !std::is_sorted(lo, hi, [&](size_t a, size_t b) { return ci[a] < ci[b]; }))
                        std::stable_sort(lo, hi, [&](size_t a, size_t b) { return ci[a] < ci[b]; });
                }
        } else { // (row indices outside the matrix: the comparison sort orders whatever is there)
            std::iota(perm.begin(), perm.end(), size_t(0));
            std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
                if (ri[a] != ri[b]) return ri[a] < ri[b];
                return and_column && ci[a] < ci[b];
            });
        }
        array1d<IndexType, host_memory> ri2(ri.size()), ci2(ri.size());
        array1d<ValueType, host_memory> va2(ri.size());
        for (size_t k = 0; k < perm.size(); k++) { ri2[k] = ri[perm[k]]; ci2[k] = ci[perm[k]]; va2[k] = va[perm[k]]; }
        row_indices = ri2; column_indices = ci2; values = va2;
    }
    static int device_sorted(int64_t r, int64_t n, const int *ai, const int *aj, int ac, int *out) { return cmi_coo_is_sorted(r, n, ai, aj, ac, out, nullptr); }
    template <typename A> static int device_sorted(int64_t, int64_t, const A *, const A *, int, int *) { return -1; }
    bool sorted_impl(bool and_column) const
    {
        if (on_device) {
            int sorted = 0;
            detail::check(device_sorted((int64_t)this->num_rows, (int64_t)row_indices.size(), row_indices.data(), column_indices.data(), and_column ? 1 : 0, &sorted));
            return sorted != 0;
        }
        array1d<IndexType, host_memory> ri(row_indices), ci(column_indices);
        for (size_t k = 1; k < ri.size(); k++) {
            if (ri[k - 1] > ri[k]) return false;
            if (and_column && ri[k - 1] == ri[k] && ci[k - 1] > ci[k]) return false;
        }
        return true;
    }
};

template <typename I, typename V, typename M>
class coo_matrix_view : public detail::matrix_base<typename std::remove_const<I>::type, typename std::remove_const<V>::type, M, coo_format> {
    typedef detail::matrix_base<typename std::remove_const<I>::type, typename std::remove_const<V>::type, M, coo_format> Parent;
public:
    typedef coo_matrix<typename Parent::index_type, typename Parent::value_type, M> container;
    typedef coo_matrix_view view;
    array1d_view<I, M> row_indices, column_indices;
    array1d_view<V, M> values;

    coo_matrix_view() {}
    coo_matrix_view(size_t rows, size_t cols, size_t entries, array1d_view<I, M> ri, array1d_view<I, M> ci, array1d_view<V, M> v)
        : Parent(rows, cols, entries), row_indices(ri), column_indices(ci), values(v) {}
    coo_matrix_view(container &m) : Parent(m), row_indices(m.row_indices), column_indices(m.column_indices), values(m.values) {}
    coo_matrix_view(const container &m) : Parent(m), row_indices(m.row_indices), column_indices(m.column_indices), values(m.values) {}
};

template <typename I, typename V, typename M>
coo_matrix_view<I, V, M> make_coo_matrix_view(size_t rows, size_t cols, size_t entries, array1d_view<I, M> ri, array1d_view<I, M> ci, array1d_view<V, M> v)
{
    return coo_matrix_view<I, V, M>(rows, cols, entries, ri, ci, v);
}
template <typename I, typename V, typename M> coo_matrix_view<I, V, M> make_coo_matrix_view(coo_matrix<I, V, M> &m) { return coo_matrix_view<I, V, M>(m); }
template <typename I, typename V, typename M> coo_matrix_view<const I, const V, M> make_coo_matrix_view(const coo_matrix<I, V, M> &m) { return coo_matrix_view<const I, const V, M>(m); }
template <typename I, typename V, typename M> coo_matrix_view<I, V, M> make_coo_matrix_view(const coo_matrix_view<I, V, M> &v) { return v; }

// ---------------------------------------------------------------------------------------------
// ELL
// ---------------------------------------------------------------------------------------------
template <typename I, typename V, typename M> class ell_matrix_view;

template <typename IndexType, typename ValueType, typename MemorySpace>
class ell_matrix : public detail::matrix_base<IndexType, ValueType, MemorySpace, ell_format> {
    typedef detail::matrix_base<IndexType, ValueType, MemorySpace, ell_format> Parent;
public:
    typedef array2d<IndexType, MemorySpace, column_major> column_indices_array_type;
    typedef array2d<ValueType, MemorySpace, column_major> values_array_type;
    typedef ell_matrix container;
    typedef ell_matrix_view<IndexType, ValueType, MemorySpace> view;
    typedef ell_matrix_view<IndexType, ValueType, MemorySpace> const_view;
    template <typename Space> struct rebind { typedef ell_matrix<IndexType, ValueType, Space> type; };

    static const IndexType invalid_index = static_cast<IndexType>(-1); // reference cusp/ell_matrix.h:129

    column_indices_array_type column_indices;
    values_array_type values;

    ell_matrix() {}
    // reference cusp/detail/ell_matrix.inl:30-37: pitch = round_up(num_rows, alignment), default 32
    ell_matrix(size_t rows, size_t cols, size_t entries, size_t entries_per_row, size_t alignment = 32) { resize(rows, cols, entries, entries_per_row, alignment); }
    ell_matrix(const ell_matrix &) = default;
    ell_matrix(ell_matrix &&) = default;
    ell_matrix &operator=(const ell_matrix &) = default;
    ell_matrix &operator=(ell_matrix &&) = default;
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    ell_matrix(const Matrix &m) { cusp::convert(m, *this); }
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    ell_matrix &operator=(const Matrix &m) { cusp::convert(m, *this); return *this; }

    void resize(size_t rows, size_t cols, size_t entries, size_t entries_per_row, size_t alignment = 32)
    {
        Parent::resize(rows, cols, entries);
        const size_t pitch = detail::round_up(rows, alignment);
        column_indices.resize(rows, entries_per_row, pitch);
        values.resize(rows, entries_per_row, pitch);
    }
    void swap(ell_matrix &o) { Parent::swap(o); column_indices.swap(o.column_indices); values.swap(o.values); }
};
template <typename I, typename V, typename M> const I ell_matrix<I, V, M>::invalid_index;

template <typename I, typename V, typename M>
class ell_matrix_view : public detail::matrix_base<I, V, M, ell_format> {
    typedef detail::matrix_base<I, V, M, ell_format> Parent;
public:
    typedef ell_matrix<I, V, M> container;
    typedef ell_matrix_view view;
    static const I invalid_index = static_cast<I>(-1);
    // 2-D views: pointer + shape + pitch of the container's column-major arrays
    struct view2d_i { const I *ptr; size_t num_rows, num_cols, pitch; const I *data() const { return ptr; } } column_indices;
    struct view2d_v { const V *ptr; size_t num_rows, num_cols, pitch; const V *data() const { return ptr; } } values;
    ell_matrix_view(const container &m)
        : Parent(m), column_indices{m.column_indices.values.data(), m.column_indices.num_rows, m.column_indices.num_cols, m.column_indices.pitch},
          values{m.values.values.data(), m.values.num_rows, m.values.num_cols, m.values.pitch} {}
};
template <typename I, typename V, typename M> const I ell_matrix_view<I, V, M>::invalid_index;

// ---------------------------------------------------------------------------------------------
// DIA
// ---------------------------------------------------------------------------------------------
template <typename I, typename V, typename M> class dia_matrix_view;

template <typename IndexType, typename ValueType, typename MemorySpace>
class dia_matrix : public detail::matrix_base<IndexType, ValueType, MemorySpace, dia_format> {
    typedef detail::matrix_base<IndexType, ValueType, MemorySpace, dia_format> Parent;
public:
    typedef array1d<IndexType, MemorySpace> diagonal_offsets_array_type;
    typedef array2d<ValueType, MemorySpace, column_major> values_array_type;
    typedef dia_matrix container;
    typedef dia_matrix_view<IndexType, ValueType, MemorySpace> view;
    typedef dia_matrix_view<IndexType, ValueType, MemorySpace> const_view;
    template <typename Space> struct rebind { typedef dia_matrix<IndexType, ValueType, Space> type; };

    diagonal_offsets_array_type diagonal_offsets;
    values_array_type values;

    dia_matrix() {}
    dia_matrix(size_t rows, size_t cols, size_t entries, size_t diagonals, size_t alignment = 32) { resize(rows, cols, entries, diagonals, alignment); }
    dia_matrix(const dia_matrix &) = default;
    dia_matrix(dia_matrix &&) = default;
    dia_matrix &operator=(const dia_matrix &) = default;
    dia_matrix &operator=(dia_matrix &&) = default;
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    dia_matrix(const Matrix &m) { cusp::convert(m, *this); }
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    dia_matrix &operator=(const Matrix &m) { cusp::convert(m, *this); return *this; }

    // reference cusp/detail/dia_matrix.inl:26-35,65-69
    void resize(size_t rows, size_t cols, size_t entries, size_t diagonals, size_t alignment = 32)
    {
        Parent::resize(rows, cols, entries);
        diagonal_offsets.resize(diagonals);
        values.resize(rows, diagonals, detail::round_up(rows, alignment));
    }
    void swap(dia_matrix &o) { Parent::swap(o); diagonal_offsets.swap(o.diagonal_offsets); values.swap(o.values); }
};

template <typename I, typename V, typename M>
class dia_matrix_view : public detail::matrix_base<I, V, M, dia_format> {
    typedef detail::matrix_base<I, V, M, dia_format> Parent;
public:
    typedef dia_matrix<I, V, M> container;
    typedef dia_matrix_view view;
    array1d_view<const I, M> diagonal_offsets;
    struct view2d_v { const V *ptr; size_t num_rows, num_cols, pitch; const V *data() const { return ptr; } } values;
    dia_matrix_view(const container &m)
        : Parent(m), diagonal_offsets(m.diagonal_offsets),
          values{m.values.values.data(), m.values.num_rows, m.values.num_cols, m.values.pitch} {}
};

// ---------------------------------------------------------------------------------------------
// HYB
// ---------------------------------------------------------------------------------------------
template <typename I, typename V, typename M> class hyb_matrix_view;

template <typename IndexType, typename ValueType, typename MemorySpace>
class hyb_matrix : public detail::matrix_base<IndexType, ValueType, MemorySpace, hyb_format> {
    typedef detail::matrix_base<IndexType, ValueType, MemorySpace, hyb_format> Parent;
public:
    typedef ell_matrix<IndexType, ValueType, MemorySpace> ell_matrix_type;
    typedef coo_matrix<IndexType, ValueType, MemorySpace> coo_matrix_type;
    typedef hyb_matrix container;
    typedef hyb_matrix_view<IndexType, ValueType, MemorySpace> view;
    typedef hyb_matrix_view<IndexType, ValueType, MemorySpace> const_view;
    template <typename Space> struct rebind { typedef hyb_matrix<IndexType, ValueType, Space> type; };

    ell_matrix_type ell;
    coo_matrix_type coo;

    hyb_matrix() {}
    hyb_matrix(size_t rows, size_t cols, size_t ell_entries, size_t coo_entries, size_t entries_per_row, size_t alignment = 32)
    {
        resize(rows, cols, ell_entries, coo_entries, entries_per_row, alignment);
    }
    hyb_matrix(const hyb_matrix &) = default;
    hyb_matrix(hyb_matrix &&) = default;
    hyb_matrix &operator=(const hyb_matrix &) = default;
    hyb_matrix &operator=(hyb_matrix &&) = default;
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    hyb_matrix(const Matrix &m) { cusp::convert(m, *this); }
    template <typename Matrix, typename = typename std::enable_if<detail::has_format<Matrix>::value>::type>
    hyb_matrix &operator=(const Matrix &m) { cusp::convert(m, *this); return *this; }

    void resize(size_t rows, size_t cols, size_t ell_entries, size_t coo_entries, size_t entries_per_row, size_t alignment = 32)
    {
        Parent::resize(rows, cols, ell_entries + coo_entries);
        ell.resize(rows, cols, ell_entries, entries_per_row, alignment);
        coo.resize(rows, cols, coo_entries);
        plan_.reset();
    }
    void swap(hyb_matrix &o) { Parent::swap(o); ell.swap(o.ell); coo.swap(o.coo); std::swap(plan_, o.plan_); }

    // device_memory: the engine's plan (detail::hyb_plan_slot) -- COO part sorted by row: cusp::multiply is ONE launch
    const cmi_plan *plan(void *stream = nullptr) const
    {
        return plan_.get(detail::dtype_code<ValueType>::value, this->num_rows, this->num_cols, ell.column_indices.num_cols,
                         coo.num_entries, coo.row_indices.data(), stream);
    }
    void invalidate_plan() const { plan_.reset(); }
private:
    mutable detail::hyb_plan_slot plan_;
};

template <typename I, typename V, typename M>
class hyb_matrix_view : public detail::matrix_base<I, V, M, hyb_format> {
    typedef detail::matrix_base<I, V, M, hyb_format> Parent;
public:
    typedef hyb_matrix<I, V, M> container;
    typedef hyb_matrix_view view;
    ell_matrix_view<I, V, M> ell;
    coo_matrix_view<const I, const V, M> coo;
    hyb_matrix_view(const container &m) : Parent(m), ell(m.ell), coo(m.coo) {}
};

// ---------------------------------------------------------------------------------------------
// ELLR (the fork's cusp::ktt::ellr_matrix, cusp/ktt/ellr_matrix.h:17-90): ELL + per-row lengths
// ---------------------------------------------------------------------------------------------
namespace ktt {
template <typename IndexType, typename ValueType, typename MemorySpace>
class ellr_matrix : public cusp::ell_matrix<IndexType, ValueType, MemorySpace> {
    typedef cusp::ell_matrix<IndexType, ValueType, MemorySpace> Parent;
public:
    typedef ellr_matrix container;
    cusp::array1d<IndexType, MemorySpace> row_lengths;

    ellr_matrix() {}
    ellr_matrix(size_t rows, size_t cols, size_t entries, size_t entries_per_row, size_t alignment = 32)
        : Parent(rows, cols, entries, entries_per_row, alignment), row_lengths(rows) {}
    template <typename Matrix, typename = typename std::enable_if<cusp::detail::has_format<Matrix>::value>::type>
    ellr_matrix(const Matrix &m) : Parent(m) { compute_row_lengths(); }
    template <typename Matrix, typename = typename std::enable_if<cusp::detail::has_format<Matrix>::value>::type>
    ellr_matrix &operator=(const Matrix &m) { Parent::operator=(m); compute_row_lengths(); return *this; }

    // reference cusp/ktt/detail/ellr_matrix.inl:16-53: length of the leading run of valid columns
    void compute_row_lengths();
};
} // namespace ktt

} // namespace cusp
