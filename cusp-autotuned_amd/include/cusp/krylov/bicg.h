// cusp/krylov/bicg.h -- cusp::krylov::bicg(A, At, x, b[, monitor[, M, Mt]]): bi-conjugate gradients for non-symmetric systems, with the transposed
// operator supplied by the caller (cusp::transpose) (reference cusp/krylov/bicg.h, detail/bicg.inl:41-141 -- the same operation order: the early
// return on a converged start, alpha = rho / <p*, A p>, the three updates, the convergence test on r BEFORE the preconditioner is applied again, the
// breakdown exit on rho == 0).  A caller of the hot path twice over: one cusp::multiply with A and one with At per iteration, each through its plan.
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
void bicg(const LinearOperator &A, const LinearOperator &At, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M, Preconditioner &Mt)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols || At.num_rows != A.num_cols || At.num_cols != A.num_rows) throw cusp::invalid_input_exception("bicg: A must be square and At its transpose");
    const size_t N = A.num_rows;
    const bool plain = detail::is_identity<Preconditioner>::value; // identity preconditioners: z IS r, z* IS r* (no copies)
    cusp::array1d<ValueType, MemorySpace> y(N), p(N), p_star(N), q(N), q_star(N), r(N), r_star(N), z_own(plain ? 0 : N), zs_own(plain ? 0 : N);
    cusp::array1d<ValueType, MemorySpace> &z = plain ? r : z_own, &z_star = plain ? r_star : zs_own;

    cusp::multiply(A, x, y);
    cusp::blas::axpby(b, y, r, ValueType(1), ValueType(-1)); // r <- b - A x
    if (monitor.finished(r)) return;
    cusp::blas::copy(r, r_star);
    if (!plain) { detail::apply(M, r, z, 0); detail::apply(Mt, r_star, z_star, 0); }
    ValueType rho = cusp::blas::dotc(z, r_star);
    cusp::blas::copy(z, p);
    cusp::blas::copy(z_star, p_star);
    for (;;) {
        cusp::multiply(A, p, q);                              // q <- A p
        cusp::multiply(At, p_star, q_star);                   // q* <- A^T p*
        const ValueType alpha = rho / cusp::blas::dotc(p_star, q);
        cusp::blas::axpby(x, p, x, ValueType(1), alpha);
        cusp::blas::axpby(r, q, r, ValueType(1), -alpha);
        cusp::blas::axpby(r_star, q_star, r_star, ValueType(1), -alpha);
        if (monitor.finished(r)) break;
        if (!plain) { detail::apply(M, r, z, 0); detail::apply(Mt, r_star, z_star, 0); }
        const ValueType prev_rho = rho;
        rho = cusp::blas::dotc(z, r_star);
        if (rho == ValueType(0)) break;                       // breakdown (bicg.inl:127-131)
        const ValueType beta = rho / prev_rho;
        cusp::blas::axpby(p, z, p, beta, ValueType(1));       // p <- z + beta p
        cusp::blas::axpby(p_star, z_star, p_star, beta, ValueType(1));
        ++monitor;
    }
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename = detail::not_policy<LinearOperator>>
void bicg(const LinearOperator &A, const LinearOperator &At, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    cusp::identity_operator<typename LinearOperator::value_type, typename LinearOperator::memory_space> M(A.num_rows, A.num_cols);
    cusp::krylov::bicg(A, At, x, b, monitor, M, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename = detail::not_policy<LinearOperator>>
void bicg(const LinearOperator &A, const LinearOperator &At, VectorType1 &x, const VectorType2 &b)
{
    cusp::monitor<typename LinearOperator::value_type> monitor(b);
    cusp::krylov::bicg(A, At, x, b, monitor);
}
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void bicg(const cusp::execution_policy<Derived> &, const LinearOperator &A, const LinearOperator &At, VectorType1 &x, const VectorType2 &b, Monitor &monitor,
          Preconditioner &M, Preconditioner &Mt)
{ cusp::krylov::bicg(A, At, x, b, monitor, M, Mt); }

} // namespace krylov
} // namespace cusp
