// cusp/format_utils.h -- the reference's include path for offsets_to_indices, indices_to_offsets, compute_max_entries_per_row and
// compute_optimal_entries_per_row (cusp/format_utils.h:83,133,276,320).  The general (host) versions live in cusp/convert.h, which the
// conversions use; here are the device_memory overloads that stay in HBM: CSR row offsets <-> COO row indices through the C-ABI builders
// (cmi_csr_row_indices; cmi_coo_row_offsets for row-sorted indices, the container's contract -- indices in any order take the host version).
// extract_diagonal / count_diagonals serve the preconditioners and are not on the multiply path (out of scope).
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

} // namespace cusp
