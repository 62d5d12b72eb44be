// cusp/precond/diagonal.h -- cusp::precond::diagonal<ValueType, MemorySpace>: the Jacobi preconditioner, y <- D^-1 x
// (reference cusp/precond/diagonal.h:85-108, detail/diagonal.inl: extract_diagonal, reciprocals, operator() = blas::xmy).
// Built once from any of the five formats (set-up: the diagonal is read from a host copy of the matrix and the reciprocals are placed in
// MemorySpace); applying it is ONE elementwise kernel (cmi_blas_xmy_*), which is what makes cusp::krylov::cg(A, x, b, monitor, M) with this M
// a device-resident solve: the multiply through A's plan + the library's BLAS-1 per iteration.
#pragma once
#include "../blas/blas.h"
#include "../format_utils.h"
#include "../linear_operator.h"

namespace cusp {
namespace precond {

template <typename ValueType, typename MemorySpace> class diagonal : public cusp::linear_operator<ValueType, MemorySpace> {
    typedef cusp::linear_operator<ValueType, MemorySpace> Parent;
    cusp::array1d<ValueType, MemorySpace> diagonal_reciprocals;

public:
    template <typename MatrixType> diagonal(const MatrixType &A) : Parent(A.num_rows, A.num_cols, A.num_rows), diagonal_reciprocals(A.num_rows)
    {
        cusp::array1d<ValueType, cusp::host_memory> d(A.num_rows);
        cusp::extract_diagonal(A, d);
        for (size_t i = 0; i < d.size(); i++) d[i] = ValueType(1) / d[i]; // (a zero on the diagonal gives inf, as in the reference)
        diagonal_reciprocals = d;
    }
    // 1 / a_ii (what cusp::krylov::cg's fused Jacobi path on device_memory reads instead of storing z = D^-1 r)
    const cusp::array1d<ValueType, MemorySpace> &reciprocals() const { return diagonal_reciprocals; }
    template <typename VectorType1, typename VectorType2> void operator()(const VectorType1 &x, VectorType2 &y) const
    {
        cusp::blas::xmy(diagonal_reciprocals, x, y);
    }
};

} // namespace precond
} // namespace cusp
