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

// Modified Gram-Schmidt of w against V[0..i]: hcol[k] <- <V[k], w> with w <- w - hcol[k] V[k] applied before the next dot; hcol[i + 1] <- ||w||.
// Any memory space: one dotc + one axpy per vector (gmres.inl:145-152: i + 2 host reads).  device_memory, float / double: a chain of fused steps
// (cmi_blas_axpy_dot_*: the axpy of vector k and the dot with vector k + 1 in one pass, coefficients in device memory) and ONE read of the column.
template <typename Vec, typename ValueType, typename Space> void orthogonalize(const std::vector<Vec> &V, int i, Vec &w, ValueType *hcol, Space)
{
    for (int k = 0; k <= i; k++) {
        hcol[k] = cusp::blas::dotc(V[k], w);
        cusp::blas::axpy(V[k], w, -hcol[k]);
    }
    hcol[i + 1] = cusp::blas::nrm2(w);
}
inline int axpy_dot_(size_t n, const double *h, const double *v, double *w, const double *u, double *out, void *ws) { return cmi_blas_axpy_dot_f64(n, h, v, w, u, out, ws, nullptr); }
inline int axpy_dot_(size_t n, const double *h, const float *v, float *w, const float *u, double *out, void *ws) { return cmi_blas_axpy_dot_f32(n, h, v, w, u, out, ws, nullptr); }
template <typename ValueType> struct mgs_on_device : std::integral_constant<bool, std::is_same<ValueType, double>::value || std::is_same<ValueType, float>::value> {};
template <typename Vec, typename ValueType>
typename std::enable_if<mgs_on_device<ValueType>::value>::type orthogonalize(const std::vector<Vec> &V, int i, Vec &w, ValueType *hcol, cusp::device_memory)
{
    static thread_local cusp::array1d<double, cusp::device_memory> coeff;
    if (coeff.size() < static_cast<size_t>(i + 2)) coeff.resize(static_cast<size_t>(i + 2) + 32);
    cusp::blas::detail::device_workspace &ws = cusp::blas::detail::workspace();
    const size_t n = w.size();
    double *c = coeff.data();
    cusp::detail::check(axpy_dot_(n, nullptr, w.data(), w.data(), V[0].data(), c, ws.ws));                                   // <V[0], w>
    for (int k = 0; k < i; k++) cusp::detail::check(axpy_dot_(n, c + k, V[k].data(), w.data(), V[k + 1].data(), c + k + 1, ws.ws)); // w -= h_k V[k]; <V[k+1], w>
    cusp::detail::check(axpy_dot_(n, c + i, V[i].data(), w.data(), w.data(), c + i + 1, ws.ws));                             // w -= h_i V[i]; <w, w>
    std::vector<double> host(static_cast<size_t>(i + 2));
    cusp::detail::check(cmi_memcpy_d2h(host.data(), c, host.size() * sizeof(double), nullptr));
    for (int k = 0; k <= i; k++) hcol[k] = static_cast<ValueType>(host[k]);
    hcol[i + 1] = static_cast<ValueType>(std::sqrt(host[i + 1]));
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
            detail::orthogonalize(V, i, w, &h(0, i), MemorySpace()); // modified Gram-Schmidt: h(0..i, i) and h(i + 1, i) = ||w||
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
