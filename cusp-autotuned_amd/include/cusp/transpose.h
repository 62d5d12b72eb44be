// cusp/transpose.h -- cusp::transpose(A, At): At <- A^T for any of the formats (reference cusp/transpose.h:84, generic/transpose.inl: per format, all of
// them through sorted coordinates in the end).  Here: A -> COO, rows and columns exchanged, sort_by_row_and_column -- on device_memory the C-ABI's
// stable radix sort (cmi_coo_sort_by_row_*: everything stays in HBM) -- then COO -> the target format.  What bi-conjugate gradients needs beside A
// (cusp/krylov/bicg.h); not on the multiply path itself.
#pragma once
#include "convert.h"
#include "coo_matrix.h"

namespace cusp {

template <typename MatrixType1, typename MatrixType2> void transpose(const MatrixType1 &A, MatrixType2 &At)
{
    typedef typename MatrixType2::index_type I;
    typedef typename MatrixType2::value_type V;
    typedef typename MatrixType2::memory_space M;
    coo_matrix<I, V, M> C(A);            // (a conversion in A's space when the spaces agree, the containers' cross-space copy otherwise)
    C.row_indices.swap(C.column_indices);
    std::swap(C.num_rows, C.num_cols);
    C.invalidate_plan();
    C.sort_by_row_and_column();
    At = C;
}

} // namespace cusp
