// cusp/verify.h -- is_valid_matrix(A[, stream]), assert_is_valid_matrix(A), assert_same_dimensions(...)
// (reference cusp/verify.h, cusp/detail/verify.inl:97-430): the structural contract a container must meet before cusp::multiply may
// touch it.  Set-up work: the index arrays are inspected on host copies (one D2H copy each for device_memory containers); nothing here is
// on the multiply path.  What is checked, per format:
//   coo  the three arrays hold num_entries elements; rows in [0, num_rows) and non-decreasing; columns in [0, num_cols)
//   csr  row_offsets holds num_rows + 1 values from 0 to num_entries, non-decreasing; column / value arrays of num_entries; columns in range
//   dia  the value array has at least num_rows rows
//   ell  index and value arrays of one shape with at least num_rows rows; as many valid columns as num_entries, all of them in range
//        (a slot is padding when its column is invalid_index)
//   hyb  the parts agree with the whole in shape; num_entries = the parts' sum; both parts valid
//   array2d  num_entries = num_rows * num_cols = the size of the value array
#pragma once
#include <ostream>
#include <sstream>

#include "array1d.h"
#include "array2d.h"
#include "detail/matrices.h"
#include "exception.h"

namespace cusp {
namespace detail {

template <typename Array, typename Out> bool indices_in_range(const Array &a, size_t bound, const char *what, Out &out, bool non_decreasing)
{
    array1d<typename Array::value_type, host_memory> h(a);
    for (size_t k = 0; k < h.size(); k++) {
        if (h[k] < 0) { out << what << " " << k << " is negative (" << h[k] << ")"; return false; }
        if (static_cast<size_t>(h[k]) >= bound) { out << what << " " << k << " (" << h[k] << ") is not below " << bound; return false; }
        if (non_decreasing && k > 0 && h[k] < h[k - 1]) { out << what << " " << k << " (" << h[k] << ") is smaller than its predecessor (" << h[k - 1] << ")"; return false; }
    }
    return true;
}

template <typename M, typename Out> bool valid(const M &A, Out &out, coo_format)
{
    if (A.row_indices.size() != A.num_entries || A.column_indices.size() != A.num_entries || A.values.size() != A.num_entries) {
        out << "array lengths (" << A.row_indices.size() << ", " << A.column_indices.size() << ", " << A.values.size() << ") differ from num_entries (" << A.num_entries << ")";
        return false;
    }
    return indices_in_range(A.row_indices, A.num_rows, "row index", out, true) && indices_in_range(A.column_indices, A.num_cols, "column index", out, false);
}

template <typename M, typename Out> bool valid(const M &A, Out &out, csr_format)
{
    if (A.row_offsets.size() != A.num_rows + 1) { out << "row_offsets holds " << A.row_offsets.size() << " values for " << A.num_rows << " rows"; return false; }
    if (A.column_indices.size() != A.num_entries || A.values.size() != A.num_entries) {
        out << "array lengths (" << A.column_indices.size() << ", " << A.values.size() << ") differ from num_entries (" << A.num_entries << ")";
        return false;
    }
    array1d<typename M::index_type, host_memory> off(A.row_offsets);
    if (off[0] != 0) { out << "row_offsets starts at " << off[0] << ", not 0"; return false; }
    if (static_cast<size_t>(off[A.num_rows]) != A.num_entries) { out << "row_offsets ends at " << off[A.num_rows] << ", not num_entries (" << A.num_entries << ")"; return false; }
    for (size_t i = 0; i < A.num_rows; i++)
        if (off[i + 1] < off[i]) { out << "row_offsets decreases at row " << i << " (" << off[i] << " -> " << off[i + 1] << ")"; return false; }
    return indices_in_range(A.column_indices, A.num_cols, "column index", out, false);
}

template <typename M, typename Out> bool valid(const M &A, Out &out, dia_format)
{
    if (A.num_rows > A.values.num_rows) { out << "the value array has " << A.values.num_rows << " rows for a matrix of " << A.num_rows; return false; }
    return true;
}

template <typename M, typename Out> bool valid(const M &A, Out &out, ell_format)
{
    if (A.column_indices.num_rows != A.values.num_rows || A.column_indices.num_cols != A.values.num_cols) {
        out << "column_indices is " << A.column_indices.num_rows << " x " << A.column_indices.num_cols << " but values is " << A.values.num_rows << " x " << A.values.num_cols;
        return false;
    }
    if (A.num_rows > A.values.num_rows) { out << "the value array has " << A.values.num_rows << " rows for a matrix of " << A.num_rows; return false; }
    typedef typename M::index_type I;
    array1d<I, host_memory> cj(A.column_indices.values);
    size_t present = 0, in_range = 0;
    for (size_t k = 0; k < cj.size(); k++) {
        if (cj[k] == I(M::invalid_index)) continue;
        present++;
        in_range += cj[k] >= 0 && static_cast<size_t>(cj[k]) < A.num_cols;
    }
    if (present != A.num_entries) { out << present << " slots carry a column index, num_entries says " << A.num_entries; return false; }
    if (in_range != present) { out << (present - in_range) << " column indices lie outside [0, " << A.num_cols << ")"; return false; }
    return true;
}

template <typename M, typename Out> bool valid(const M &A, Out &out, hyb_format)
{
    if (A.num_rows != A.ell.num_rows || A.num_rows != A.coo.num_rows || A.num_cols != A.ell.num_cols || A.num_cols != A.coo.num_cols) {
        out << "the parts' shapes (" << A.ell.num_rows << " x " << A.ell.num_cols << ", " << A.coo.num_rows << " x " << A.coo.num_cols << ") differ from the matrix's ("
            << A.num_rows << " x " << A.num_cols << ")";
        return false;
    }
    if (A.num_entries != A.ell.num_entries + A.coo.num_entries) {
        out << "num_entries (" << A.num_entries << ") is not the parts' sum (" << A.ell.num_entries << " + " << A.coo.num_entries << ")";
        return false;
    }
    return valid(A.ell, out, ell_format()) && valid(A.coo, out, coo_format());
}

template <typename M, typename Out> bool valid(const M &A, Out &out, array2d_format)
{
    if (A.num_rows * A.num_cols != A.num_entries || A.num_entries > A.values.size()) {
        out << A.num_rows << " x " << A.num_cols << " entries, num_entries " << A.num_entries << ", " << A.values.size() << " values stored";
        return false;
    }
    return true;
}

} // namespace detail

template <typename MatrixType, typename OutputStream> bool is_valid_matrix(const MatrixType &A, OutputStream &ostream)
{
    return detail::valid(A, ostream, typename MatrixType::format());
}
template <typename MatrixType> bool is_valid_matrix(const MatrixType &A)
{
    std::ostringstream unused;
    return cusp::is_valid_matrix(A, unused);
}
template <typename MatrixType> void assert_is_valid_matrix(const MatrixType &A)
{
    std::ostringstream why;
    if (!cusp::is_valid_matrix(A, why)) throw cusp::format_exception(why.str());
}

template <typename Array1, typename Array2> void assert_same_dimensions(const Array1 &a, const Array2 &b)
{
    if (a.size() != b.size()) throw cusp::invalid_input_exception("array dimensions do not match");
}
template <typename Array1, typename Array2, typename Array3> void assert_same_dimensions(const Array1 &a, const Array2 &b, const Array3 &c)
{
    assert_same_dimensions(a, b);
    assert_same_dimensions(b, c);
}
template <typename Array1, typename Array2, typename Array3, typename Array4> void assert_same_dimensions(const Array1 &a, const Array2 &b, const Array3 &c, const Array4 &d)
{
    assert_same_dimensions(a, b, c);
    assert_same_dimensions(c, d);
}

} // namespace cusp
