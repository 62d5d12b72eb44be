// cusp/krylov/gmres.h -- cusp::krylov::gmres(A, x, b, restart[, monitor[, M]]): restarted GMRES, left-preconditioned, Givens rotations on the
// host (reference cusp/krylov/gmres.h, detail/gmres.inl:36-211 -- the same operation order: w = M (A x - b) negated and normalised, modified
// Gram-Schmidt against the Krylov vectors one dot + one axpy at a time, the least-squares residual |s[i+1]| handed to the monitor after every inner
// step, the small triangular solve and the update of x at a restart).  A caller of the hot path: ONE cusp::multiply(A, ., .) per inner
// iteration through A's plan; the basis lives in the operator's memory space (restart + 1 vectors), the (restart + 1) x restart Hessenberg matrix
// on the host.  Real value types.
#pragma once
#include <cmath>
#include <vector>

#include "../array1d.h"
#include "../blas/blas.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"
#include "cg.h"

namespace cusp {
namespace krylov {
namespace detail {

template <typename V> void apply_rotation(V &dx, V &dy, V cs, V sn)
{
    const V t = cs * dx + sn * dy;
    dy = -sn * dx + cs * dy;
    dx = t;
}
template <typename V> void make_rotation(V dx, V dy, V &cs, V &sn) // the rotation that zeroes dy against dx
{
    if (dx == V(0)) { cs = V(0); sn = V(1); return; }
    const V scale = std::abs(dx) + std::abs(dy);
    const V norm = scale * std::sqrt((dx / scale) * (dx / scale) + (dy / scale) * (dy / scale));
    cs = std::abs(dx) / norm;
    sn = (dx / std::abs(dx)) * dy / norm;
}

} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = detail::not_policy<LinearOperator>>
void gmres(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, const size_t restart, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("gmres: the operator must be square");
    if (restart == 0) throw cusp::invalid_input_exception("gmres: restart must be positive");
    const size_t N = A.num_rows;
    const int R = static_cast<int>(restart);
    const bool plain = detail::is_identity<Preconditioner>::value;
    cusp::array1d<ValueType, MemorySpace> w(N), t(N);
    std::vector<cusp::array1d<ValueType, MemorySpace>> V(R + 1, cusp::array1d<ValueType, MemorySpace>(N, ValueType(0))); // the Krylov basis
    std::vector<ValueType> H((size_t)(R + 1) * R, ValueType(0)), s(R + 1), cs(R), sn(R);                               // H(k, i) = H[k + i (R + 1)]
    auto h = [&](int k, int i) -> ValueType & { return H[(size_t)k + (size_t)i * (R + 1)]; };
    cusp::array1d<ValueType, cusp::host_memory> resid(1);
    int i = -1;

    do {
        cusp::multiply(A, x, w);
        cusp::blas::axpy(b, w, ValueType(-1));            // w <- A x - b
        if (plain) t.swap(w); else detail::apply(M, w, t, 0); // t <- M (A x - b)   (operators here take distinct arguments)
        const ValueType beta = cusp::blas::nrm2(t);
        cusp::blas::scal(t, ValueType(-1.0 / beta));      // the first basis vector: M (b - A x) / beta
        cusp::blas::copy(t, V[0]);
        std::fill(s.begin(), s.end(), ValueType(0));
        s[0] = beta;
        i = -1;
        resid[0] = std::abs(s[0]);
        if (monitor.finished(resid)) break;
        do {
            ++i;
            ++monitor;
            if (plain) cusp::multiply(A, V[i], w);        // (the hot path) M = identity_operator: w <- A v_i directly
            else { cusp::multiply(A, V[i], t); detail::apply(M, t, w, 0); } // w <- M A v_i
            for (int k = 0; k <= i; k++) {                // modified Gram-Schmidt
                h(k, i) = cusp::blas::dotc(V[k], w);
                cusp::blas::axpy(V[k], w, -h(k, i));
            }
            h(i + 1, i) = cusp::blas::nrm2(w);
            cusp::blas::scal(w, ValueType(1.0) / h(i + 1, i));
            cusp::blas::copy(w, V[i + 1]);
            for (int k = 0; k < i; k++) detail::apply_rotation(h(k, i), h(k + 1, i), cs[k], sn[k]);
            detail::make_rotation(h(i, i), h(i + 1, i), cs[i], sn[i]);
            detail::apply_rotation(h(i, i), h(i + 1, i), cs[i], sn[i]);
            detail::apply_rotation(s[i], s[i + 1], cs[i], sn[i]);
            resid[0] = std::abs(s[i + 1]);
            if (monitor.finished(resid)) break;
        } while (i + 1 < R && monitor.iteration_count() + 1 <= monitor.iteration_limit());
        for (int j = i; j >= 0; j--) {                    // back substitution: H(0:i, 0:i) y = s
            s[j] /= h(j, j);
            for (int k = j - 1; k >= 0; k--) s[k] -= h(k, j) * s[j];
        }
        for (int j = 0; j <= i; j++) cusp::blas::axpy(V[j], x, s[j]); // x <- x + V y
    } while (!monitor.finished(resid));
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename = detail::not_policy<LinearOperator>>
void gmres(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, const size_t restart, Monitor &monitor)
{
    cusp::identity_operator<typename LinearOperator::value_type, typename LinearOperator::memory_space> M(A.num_rows, A.num_cols);
    cusp::krylov::gmres(A, x, b, restart, monitor, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename = detail::not_policy<LinearOperator>>
void gmres(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, const size_t restart)
{
    cusp::monitor<typename LinearOperator::value_type> monitor(b);
    cusp::krylov::gmres(A, x, b, restart, monitor);
}
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void gmres(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, const size_t restart, Monitor &monitor, Preconditioner &M)
{ cusp::krylov::gmres(A, x, b, restart, monitor, M); }
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void gmres(const cusp::execution_policy<Derived> &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, const size_t restart, Monitor &monitor)
{ cusp::krylov::gmres(A, x, b, restart, monitor); }

} // namespace krylov
} // namespace cusp
