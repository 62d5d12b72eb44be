// cusp/format_utils.h -- the reference's include path for offsets_to_indices, indices_to_offsets, compute_max_entries_per_row and
// compute_optimal_entries_per_row (cusp/format_utils.h:83,133,276,320).  The general (host) versions live in cusp/convert.h, which the
// conversions use; here are the device_memory overloads that stay in HBM: CSR row offsets <-> COO row indices through the C-ABI builders
// (cmi_csr_row_indices; cmi_coo_row_offsets for row-sorted indices, the container's contract -- indices in any order take the host version).
// extract_diagonal (host set-up of the Jacobi preconditioner) is below; count_diagonals serves the DIA conversion inside cusp/convert.h.
#pragma once
#include "convert.h"

namespace cusp {

inline void offsets_to_indices(const array1d<int, device_memory> &offsets, array1d<int, device_memory> &indices)
{
    const size_t rows = offsets.size() ? offsets.size() - 1 : 0;
    int last = 0;
    if (rows) detail::check(cmi_memcpy_d2h(&last, offsets.data() + rows, sizeof(int), nullptr));
    indices.resize(static_cast<size_t>(last));
    if (rows && last > 0) {
        detail::check(cmi_csr_row_indices(static_cast<int64_t>(rows), offsets.data(), indices.data(), nullptr));
        detail::check(cmi_stream_synchronize(nullptr));
    }
}

inline void indices_to_offsets(const array1d<int, device_memory> &indices, array1d<int, device_memory> &offsets)
{
    // offsets.size() - 1 rows, as in the reference (the caller sizes `offsets`)
    if (offsets.size() == 0) return;
    int sorted = 0;
    detail::check(cmi_coo_row_offsets(static_cast<int64_t>(offsets.size() - 1), static_cast<int64_t>(indices.size()), indices.data(), offsets.data(), &sorted, nullptr));
    if (sorted) return;
    // not row-sorted (or out of range): the general version on host copies
    array1d<int, host_memory> hi(indices), ho(offsets.size());
    indices_to_offsets(hi, ho);
    offsets = ho;
}

// extract_diagonal(A, output): output[i] = A(i, i) -- 0 where the diagonal entry is not stored, the SUM where it is stored more than once
// (reference cusp/format_utils.h:184, generic/format_utils.inl extract_diagonal per format).  Set-up work (the Jacobi preconditioner's
// constructor): on a host copy of the matrix in CSR form, whatever A's format and memory space; `output` may live in either space.
template <typename MatrixType, typename ArrayType> void extract_diagonal(const MatrixType &A, ArrayType &output)
{
    typedef typename MatrixType::index_type I;
    typedef typename MatrixType::value_type V;
    detail::host_csr<I, V> H;
    detail::to_host_csr(A, H, typename MatrixType::format());
    const size_t n = std::min(A.num_rows, A.num_cols);
    array1d<typename ArrayType::value_type, host_memory> d(A.num_rows, typename ArrayType::value_type(0));
    for (size_t i = 0; i < n; i++)
        for (I jj = H.row_offsets[i]; jj < H.row_offsets[i + 1]; jj++)
            if (static_cast<size_t>(H.column_indices[jj]) == i) d[i] += H.values[jj];
    output = d;
}

} // namespace cusp
