// cusp/krylov/bicgstab.h -- cusp::krylov::bicgstab(A, x, b[, monitor[, M]]): the stabilised bi-conjugate gradient method for
// non-symmetric systems, right-preconditioned (reference cusp/krylov/bicgstab.h, detail/bicgstab.inl:48-128 -- the same operation order, so the
// same iteration counts: alpha from <r*, A M p>, the early exit on s, omega from <A M s, s> / <A M s, A M s>, x and p updated with axpbypcz).
// Another CALLER of the hot path: two cusp::multiply(A, ., .) per iteration through A's plan, everything else cusp::blas (device_memory: the
// library's kernels; the scalars make four 8-byte reads per iteration, as the reference's Thrust reductions do).  Any memory space, any format,
// any M with operator()(x, y) or a matrix.
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
void bicgstab(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("bicgstab: the operator must be square");
    const size_t N = A.num_rows;
    // M = identity_operator (the default): M p IS p and M s IS s -- the two copies per iteration the reference makes are skipped (aliases below)
    const bool plain = detail::is_identity<Preconditioner>::value;
    cusp::array1d<ValueType, MemorySpace> p(N), r(N), r_star(N), s(N), Mp_own(plain ? 0 : N), AMp(N), Ms_own(plain ? 0 : N), AMs(N);
    cusp::array1d<ValueType, MemorySpace> &Mp = plain ? p : Mp_own, &Ms = plain ? s : Ms_own;

    cusp::multiply(A, x, r);
    cusp::blas::axpby(b, r, r, ValueType(1), ValueType(-1)); // r <- b - A x
    cusp::blas::copy(r, p);
    cusp::blas::copy(r, r_star);                             // the shadow residual stays r_0
    ValueType rho = cusp::blas::dotc(r_star, r);

    while (!monitor.finished(r)) {
        if (!plain) detail::apply(M, p, Mp, 0);
        cusp::multiply(A, Mp, AMp);
        const ValueType alpha = rho / cusp::blas::dotc(r_star, AMp);
        cusp::blas::axpby(r, AMp, s, ValueType(1), -alpha);  // s <- r - alpha A M p
        if (monitor.finished(s)) {                           // half a step is enough
            cusp::blas::axpby(x, Mp, x, ValueType(1), alpha);
            break;
        }
        if (!plain) detail::apply(M, s, Ms, 0);
        cusp::multiply(A, Ms, AMs);
        const ValueType omega = cusp::blas::dotc(AMs, s) / cusp::blas::dotc(AMs, AMs);
        cusp::blas::axpbypcz(x, Mp, Ms, x, ValueType(1), alpha, omega);     // x <- x + alpha M p + omega M s
        cusp::blas::axpby(s, AMs, r, ValueType(1), -omega);                  // r <- s - omega A M s
        const ValueType rho_new = cusp::blas::dotc(r_star, r);
        const ValueType beta = (rho_new / rho) * (alpha / omega);
        rho = rho_new;
        cusp::blas::axpbypcz(r, p, AMp, p, ValueType(1), beta, -beta * omega); // p <- r + beta (p - omega A M p)
        ++monitor;
    }
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename = detail::not_policy<LinearOperator>>
void bicgstab(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    cusp::identity_operator<typename LinearOperator::value_type, typename LinearOperator::memory_space> M(A.num_rows, A.num_cols);
    cusp::krylov::bicgstab(A, x, b, monitor, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename = detail::not_policy<LinearOperator>>
void bicgstab(const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    cusp::monitor<typename LinearOperator::value_type> monitor(b);
    cusp::krylov::bicgstab(A, x, b, monitor);
}
// (the reference's policy-taking forms: the policy selects nothing here -- the containers' memory space does)
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void bicgstab(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{ cusp::krylov::bicgstab(A, x, b, monitor, M); }
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void bicgstab(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{ cusp::krylov::bicgstab(A, x, b, monitor); }

} // namespace krylov
} // namespace cusp
