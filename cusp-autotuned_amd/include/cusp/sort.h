// cusp/sort.h -- sort_by_row / sort_by_row_and_column on the three arrays of a COO matrix (reference cusp/sort.h:231,302: what
// coo_matrix::sort_by_row[_and_column] calls).  device_memory arrays of int indices and float / double values: the C-ABI's device sort
// (cmi_coo_sort_by_row_*, csrc/sort.hip: stable radix sort of the keys carrying the entry's position + gathers, in place); everything else:
// a stable host sort.  The optional index bounds of the reference's signatures (min_row, max_row, ...) are accepted and not needed: they
// only size the reference's counting sort.  counting_sort / counting_sort_by_key (cusp/sort.h:93,165) are not on the multiply path and are
// not provided.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "array1d.h"
#include "coo_matrix.h"
#include "execution_policy.h"

namespace cusp {
namespace detail {
template <typename A1, typename A2, typename A3> void sort_coo_arrays(A1 &rows, A2 &cols, A3 &vals, bool and_column)
{
    typedef typename A1::value_type I;
    typedef typename A3::value_type V;
    typedef typename A1::memory_space M;
    if (rows.size() != cols.size() || rows.size() != vals.size()) throw cusp::invalid_input_exception("sort_by_row: the three arrays differ in length");
    if (rows.size() < 2) return;
    // the container's sort on views of the caller's arrays (it decides host / device and sorts in place); the row count only bounds
    // the device sort's key bits: the largest row index + 1
    I top = 0;
    {
        array1d<I, host_memory> h(rows);
        for (size_t k = 0; k < h.size(); k++) top = std::max(top, h[k]);
    }
    coo_matrix<I, V, M> tmp(static_cast<size_t>(top) + 1, static_cast<size_t>(top) + 1, 0);
    tmp.row_indices.swap(rows);
    tmp.column_indices.swap(cols);
    tmp.values.swap(vals);
    tmp.num_entries = tmp.row_indices.size();
    try {
        if (and_column) tmp.sort_by_row_and_column(); else tmp.sort_by_row();
    } catch (...) { // hand the arrays back whatever happened
        tmp.row_indices.swap(rows); tmp.column_indices.swap(cols); tmp.values.swap(vals);
        throw;
    }
    tmp.row_indices.swap(rows);
    tmp.column_indices.swap(cols);
    tmp.values.swap(vals);
}
} // namespace detail

template <typename ArrayType1, typename ArrayType2, typename ArrayType3>
void sort_by_row(ArrayType1 &row_indices, ArrayType2 &column_indices, ArrayType3 &values, typename ArrayType1::value_type = 0,
                 typename ArrayType1::value_type = 0)
{
    detail::sort_coo_arrays(row_indices, column_indices, values, false);
}

template <typename ArrayType1, typename ArrayType2, typename ArrayType3>
void sort_by_row_and_column(ArrayType1 &row_indices, ArrayType2 &column_indices, ArrayType3 &values, typename ArrayType1::value_type = 0,
                            typename ArrayType1::value_type = 0, typename ArrayType2::value_type = 0, typename ArrayType2::value_type = 0)
{
    detail::sort_coo_arrays(row_indices, column_indices, values, true);
}

// (the reference's policy-taking overloads: the policy selects nothing here -- the arrays' memory space does)
template <typename Derived, typename ArrayType1, typename ArrayType2, typename ArrayType3>
void sort_by_row(const cusp::execution_policy<Derived> &, ArrayType1 &row_indices, ArrayType2 &column_indices, ArrayType3 &values) { sort_by_row(row_indices, column_indices, values); }
template <typename Derived, typename ArrayType1, typename ArrayType2, typename ArrayType3>
void sort_by_row_and_column(const cusp::execution_policy<Derived> &, ArrayType1 &row_indices, ArrayType2 &column_indices, ArrayType3 &values) { sort_by_row_and_column(row_indices, column_indices, values); }
} // namespace cusp
