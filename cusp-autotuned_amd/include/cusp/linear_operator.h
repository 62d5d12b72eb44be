// cusp/linear_operator.h -- identity_operator, CG's default preconditioner
// (reference cusp/linear_operator.h:183-223: operator()(x, y) copies x to y).
#pragma once
#include "blas/blas.h"

namespace cusp {

struct linear_operator_format {};

template <typename ValueType, typename MemorySpace, typename IndexType = int> class linear_operator {
public:
    typedef IndexType index_type;
    typedef ValueType value_type;
    typedef MemorySpace memory_space;
    size_t num_rows, num_cols, num_entries;
    linear_operator() : num_rows(0), num_cols(0), num_entries(0) {}
    linear_operator(size_t r, size_t c, size_t n = 0) : num_rows(r), num_cols(c), num_entries(n) {}
};

template <typename ValueType, typename MemorySpace, typename IndexType = int>
class identity_operator : public linear_operator<ValueType, MemorySpace, IndexType> {
    typedef linear_operator<ValueType, MemorySpace, IndexType> Parent;
public:
    identity_operator() {}
    identity_operator(size_t rows, size_t cols) : Parent(rows, cols) {}
    template <typename X, typename Y> void operator()(const X &x, Y &y) const { cusp::blas::copy(x, y); }
};

} // namespace cusp
