// cusp/krylov/cr.h -- cusp::krylov::cr(A, x, b[, monitor[, M]]): the conjugate residual method for symmetric (possibly indefinite) systems
// (reference cusp/krylov/cr.h, detail/cr.inl:38-125 -- the same operation order: alpha = <r, A z> / <A p, A p>, the residual recomputed from
// b - A x every 8th iteration, y = A p updated by recurrence).  A caller of the hot path: one or two cusp::multiply(A, ., .) per iteration.
#pragma once
#include "../array1d.h"
#include "../blas/blas.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"
#include "cg.h"

namespace cusp {
namespace krylov {

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = detail::not_policy<LinearOperator>>
void cr(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("cr: the operator must be square");
    const size_t N = A.num_rows, recompute_r = 8; // (cr.inl:50: how often the residual is rebuilt from b - A x)
    const bool plain = detail::is_identity<Preconditioner>::value; // M = identity_operator: z IS r (no copy per iteration)
    cusp::array1d<ValueType, MemorySpace> y(N), z_own(plain ? 0 : N), r(N), p(N), Az(N), Ax(N);
    cusp::array1d<ValueType, MemorySpace> &z = plain ? r : z_own;

    cusp::multiply(A, x, Ax);
    cusp::blas::axpby(b, Ax, r, ValueType(1), ValueType(-1)); // r <- b - A x
    if (!plain) detail::apply(M, r, z, 0);                    // z <- M r
    cusp::blas::copy(z, p);
    cusp::multiply(A, p, y);                                  // y <- A p
    cusp::multiply(A, z, Az);
    ValueType rz = cusp::blas::dotc(r, Az);                   // <r, A z>

    while (!monitor.finished(r)) {
        const ValueType alpha = rz / cusp::blas::dotc(y, y);
        cusp::blas::axpy(p, x, alpha);                        // x <- x + alpha p
        const size_t iter = monitor.iteration_count();
        if ((iter % recompute_r) && iter > 0) {
            cusp::blas::axpy(y, r, -alpha);                   // r <- r - alpha A p
        } else {
            cusp::multiply(A, x, Ax);
            cusp::blas::axpby(b, Ax, r, ValueType(1), ValueType(-1));
        }
        if (!plain) detail::apply(M, r, z, 0);
        cusp::multiply(A, z, Az);
        const ValueType rz_old = rz;
        rz = cusp::blas::dotc(r, Az);
        const ValueType beta = rz / rz_old;
        cusp::blas::axpby(z, p, p, ValueType(1), beta);       // p <- z + beta p
        cusp::blas::axpby(Az, y, y, ValueType(1), beta);      // y = A p <- A z + beta y
        ++monitor;
    }
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename = detail::not_policy<LinearOperator>>
void cr(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    cusp::identity_operator<typename LinearOperator::value_type, typename LinearOperator::memory_space> M(A.num_rows, A.num_cols);
    cusp::krylov::cr(A, x, b, monitor, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename = detail::not_policy<LinearOperator>>
void cr(const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    cusp::monitor<typename LinearOperator::value_type> monitor(b);
    cusp::krylov::cr(A, x, b, monitor);
}
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cr(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{ cusp::krylov::cr(A, x, b, monitor, M); }
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cr(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{ cusp::krylov::cr(A, x, b, monitor); }

} // namespace krylov
} // namespace cusp
