// cusp/distributed/bicgstab.h -- cusp::krylov::bicgstab on a row-block sharded operator: non-symmetric systems across the GPUs of a node.
// The reference's operation order (cusp/krylov/detail/bicgstab.inl:78-125, identity preconditioner) on sharded vectors: every rank holds its slices,
// the two multiplies of an iteration are exchange + local hot path (p and s are staged into the operator's exchange buffer by cusp::multiply),
// the four inner products and the two norms are 8-byte all-reduces done by cusp::blas -- so every rank takes the same decisions.
// Any local memory space; operation by operation (the fused single-GPU passes of cusp/krylov/bicgstab.h need their scalars all-reduced between
// launches: not built).
#pragma once
#include "../krylov/bicgstab.h"
#include "../monitor.h"
#include "multiply.h"

namespace cusp {
namespace krylov {

template <typename I, typename V, typename L, typename X, typename B, typename Monitor>
void bicgstab(const distributed::csr_matrix<I, V, L> &A, X &x, const B &b, Monitor &monitor)
{
    static_assert(std::is_same<typename X::memory_space, cusp::distributed_memory<L>>::value, "bicgstab: x must be a cusp::distributed::vector in the operator's local space");
    if (x.size() != A.local_rows() || b.size() != A.local_rows()) throw cusp::invalid_input_exception("bicgstab: x and b must be this rank's slices of the operator's row partition");
    typedef distributed::vector<V, L> vec;
    vec p = A.make_vector(), r = A.make_vector(), r_star = A.make_vector(), s = A.make_vector(), AMp = A.make_vector(), AMs = A.make_vector();
    cusp::multiply(A, x, r);
    cusp::blas::axpby(b, r, r, V(1), V(-1));
    cusp::blas::copy(r, p);
    cusp::blas::copy(r, r_star);
    V rho = cusp::blas::dotc(r_star, r);
    while (!monitor.finished(r)) {
        cusp::multiply(A, p, AMp);
        const V alpha = rho / cusp::blas::dotc(r_star, AMp);
        cusp::blas::axpby(r, AMp, s, V(1), -alpha);
        if (monitor.finished(s)) {
            cusp::blas::axpby(x, p, x, V(1), alpha);
            break;
        }
        cusp::multiply(A, s, AMs);
        const V omega = cusp::blas::dotc(AMs, s) / cusp::blas::dotc(AMs, AMs);
        cusp::blas::axpbypcz(x, p, s, x, V(1), alpha, omega);
        cusp::blas::axpby(s, AMs, r, V(1), -omega);
        const V rho_new = cusp::blas::dotc(r_star, r);
        const V beta = (rho_new / rho) * (alpha / omega);
        rho = rho_new;
        cusp::blas::axpbypcz(r, p, AMp, p, V(1), beta, -beta * omega);
        ++monitor;
    }
}
template <typename I, typename V, typename L, typename X, typename B>
void bicgstab(const distributed::csr_matrix<I, V, L> &A, X &x, const B &b)
{
    cusp::monitor<V> monitor(b);
    cusp::krylov::bicgstab(A, x, b, monitor);
}

} // namespace krylov
} // namespace cusp
