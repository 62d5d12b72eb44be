// cusp/distributed/multiply.h -- cusp::multiply(A, x, y) for a row-block sharded operator and sharded vectors: x's slices are
// exchanged into the operator's full-length buffer (one collective step, cusp/distributed/csr_matrix.h), then every rank runs the
// single-GPU hot path on its rows.  The reference's entry point (cusp/multiply.h:40,101 -> generic/multiply.inl:98-111) knows one
// device; this overload is what configs[4] adds behind the same name.
#pragma once
#include "csr_matrix.h"

namespace cusp {

template <typename I, typename V, typename L, typename X, typename Y>
void multiply(const distributed::csr_matrix<I, V, L> &A, const X &x, Y &y)
{
    static_assert(std::is_same<typename X::memory_space, cusp::distributed_memory<L>>::value && std::is_same<typename Y::memory_space, cusp::distributed_memory<L>>::value,
                  "cusp::multiply: a sharded operator multiplies sharded vectors of the same local memory space");
    if (x.size() != A.local_rows() || y.size() != A.local_rows() || x.global_size() != A.num_cols)
        throw cusp::invalid_input_exception("cusp::multiply: vector slices do not match the operator's row partition");
    // x's slice goes to its place in the exchange buffer -- unless it already lives there (A.exchange_slice(): CG's p)
    distributed::vector<V, L> slot = A.exchange_slice();
    if (static_cast<const void *>(x.data()) != static_cast<const void *>(slot.data())) {
        auto dst = slot.local();
        cusp::blas::copy(x.local(), dst);
    }
    auto yl = y.local();
    A.exchange_and_multiply(yl); // (two-sided halo mode on device_memory: interior rows on a side stream while the halo is in flight)
}

} // namespace cusp
