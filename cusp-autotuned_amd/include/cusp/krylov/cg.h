// cusp/krylov/cg.h -- (preconditioned) conjugate gradients, the caller of the SpMV hot path
// (reference cusp/krylov/cg.h, cusp/krylov/detail/cg.inl:41-107: same operations in the same order,
// so the residual history matches the reference's, e.g. docs/quickstart.md:72-87).
// One cusp::multiply per iteration -> cmi_spmv_* on device_memory; vector updates -> cusp::blas.
#pragma once
#include <cassert>

#include "../array1d.h"
#include "../blas/blas.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"

namespace cusp {
namespace krylov {

namespace detail {
// z <- M r for a matrix-like preconditioner or a linear operator with operator()
template <typename M, typename X, typename Y> auto apply(const M &m, const X &x, Y &y, int) -> decltype(m(x, y), void()) { m(x, y); }
template <typename M, typename X, typename Y> void apply(const M &m, const X &x, Y &y, long) { cusp::multiply(m, x, y); }
} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("cg: matrix must be square");
    const size_t N = A.num_rows;

    // workspace (reference: four temporary_array's, cg.inl:55-58)
    cusp::array1d<ValueType, MemorySpace> y(N), z(N), r(N), p(N);

    cusp::multiply(A, x, y);                                   // y <- A x
    cusp::blas::axpby(b, y, r, ValueType(1), ValueType(-1));   // r <- b - A x
    detail::apply(M, r, z, 0);                                 // z <- M r
    cusp::blas::copy(z, p);                                    // p <- z
    ValueType rz = cusp::blas::dotc(r, z);                     // rz = <r, z>

    while (!monitor.finished(r)) {
        cusp::multiply(A, p, y);                               // y <- A p          (the hot path)
        const ValueType alpha = rz / cusp::blas::dotc(y, p);   // alpha <- <r,z>/<y,p>
        cusp::blas::axpy(p, x, alpha);                         // x <- x + alpha p
        cusp::blas::axpy(y, r, -alpha);                        // r <- r - alpha y
        detail::apply(M, r, z, 0);                             // z <- M r
        const ValueType rz_old = rz;
        rz = cusp::blas::dotc(r, z);
        const ValueType beta = rz / rz_old;
        cusp::blas::axpby(z, p, p, ValueType(1), beta);        // p <- z + beta p
        ++monitor;
    }
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    cusp::identity_operator<ValueType, MemorySpace> M(A.num_rows, A.num_cols);
    cusp::krylov::cg(A, x, b, monitor, M);
}

template <typename LinearOperator, typename VectorType1, typename VectorType2>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    typedef typename LinearOperator::value_type ValueType;
    cusp::monitor<ValueType> monitor(b);
    cusp::krylov::cg(A, x, b, monitor);
}

} // namespace krylov
} // namespace cusp
