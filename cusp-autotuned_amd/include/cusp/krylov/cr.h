// cusp/krylov/cr.h -- cusp::krylov::cr(A, x, b[, monitor[, M]]): the conjugate residual method for symmetric (possibly indefinite) systems
// (reference cusp/krylov/cr.h, detail/cr.inl:38-125 -- the same operation order: alpha = <r, A z> / <A p, A p>, the residual recomputed from
// b - A x every 8th iteration, y = A p updated by recurrence).  A caller of the hot path: one or two cusp::multiply(A, ., .) per iteration.
#pragma once
#include <cmath>
#include <cstdlib>

#include "../array1d.h"
#include "../blas/blas.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"
#include "cg.h"

namespace cusp {
namespace krylov {
namespace detail {

// ---- device_memory, identity preconditioner: the fused iteration (csrc/blas1_extra.hip cmi_cr_*) ----------------------------------------
inline int cr_xr_(size_t n, const double *rz, const double *yy, const double *p, const double *y, double *x, double *r, int upd, double *rr, double *m, void *ws)
{ return cmi_cr_xr_f64(n, rz, yy, p, y, x, r, upd, rr, m, ws, nullptr); }
inline int cr_xr_(size_t n, const double *rz, const double *yy, const float *p, const float *y, float *x, float *r, int upd, double *rr, double *m, void *ws)
{ return cmi_cr_xr_f32(n, rz, yy, p, y, x, r, upd, rr, m, ws, nullptr); }
inline int cr_py_(size_t n, const double *rz_new, const double *rz, const double *r, const double *Ar, double *p, double *y, double *yy, void *ws)
{ return cmi_cr_py_f64(n, rz_new, rz, r, Ar, p, y, yy, ws, nullptr); }
inline int cr_py_(size_t n, const double *rz_new, const double *rz, const float *r, const float *Ar, float *p, float *y, double *yy, void *ws)
{ return cmi_cr_py_f32(n, rz_new, rz, r, Ar, p, y, yy, ws, nullptr); }

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cr_fused_device(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type T;
    typedef cusp::array1d<T, cusp::device_memory> Vec;
    const size_t N = A.num_rows, recompute_r = 8;
    Vec y(N), r(N), p(N), Az(N), Ax(N);
    cusp::array1d<double, cusp::device_memory> scalars(4); // <r, A r>[0], [1], <A p, A p>, <r, r>
    double *rz[2] = {scalars.data(), scalars.data() + 1};
    double *yy = scalars.data() + 2, *rr = scalars.data() + 3;
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    pinned_scalar rr_host;
    cusp::multiply(A, x, Ax);
    cusp::blas::axpby(b, Ax, r, T(1), T(-1));
    cusp::blas::copy(r, p);
    multiply_dot_any(A, r, Az, rz[0], w.ws, std::is_same<T, double>()); // A r and <A r, r>
    cusp::blas::copy(Az, y);                                            // y = A p (p == r at the start)
    cusp::detail::check(dotd_(N, y.data(), y.data(), yy, w.ws));
    cusp::detail::check(dotd_(N, r.data(), r.data(), rr, w.ws));
    rr_host.fetch(rr);
    int cur = 0;
    while (!monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) { // the one host read; the previous iteration's multiply
        const size_t iter = monitor.iteration_count();                                                   // and py-pass are still running behind it
        const bool update_r = (iter % recompute_r) && iter > 0;
        cusp::detail::check(cr_xr_(N, rz[cur], yy, p.data(), y.data(), x.data(), r.data(), update_r ? 1 : 0, rr, rr_host.host, w.ws));
        if (update_r) rr_host.record();
        else { // every 8th iteration the residual is rebuilt from b - A x (cr.inl:96-107)
            cusp::multiply(A, x, Ax);
            cusp::blas::axpby(b, Ax, r, T(1), T(-1));
            cusp::detail::check(dotd_(N, r.data(), r.data(), rr, w.ws));
            rr_host.fetch(rr);
        }
        multiply_dot_any(A, r, Az, rz[cur ^ 1], w.ws, std::is_same<T, double>()); // the hot path: A r and <A r, r>
        cusp::detail::check(cr_py_(N, rz[cur ^ 1], rz[cur], r.data(), Az.data(), p.data(), y.data(), yy, w.ws));
        cur ^= 1;
        ++monitor;
    }
    cusp::detail::check(cmi_device_synchronize());
}

template <typename A> auto cr_has_format(const A *) -> decltype(typename A::format(), std::true_type());
inline std::false_type cr_has_format(...);
template <typename A, typename X, typename M, typename Mon> struct use_fused_cr {
    typedef typename A::value_type T;
    static const bool value = std::is_same<typename A::memory_space, cusp::device_memory>::value &&
                              (std::is_same<T, double>::value || std::is_same<T, float>::value) && std::is_same<typename X::value_type, T>::value &&
                              is_identity<M>::value && decltype(has_finished_norm(static_cast<Mon *>(nullptr)))::value &&
                              decltype(cr_has_format(static_cast<const A *>(nullptr)))::value;
};
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
bool cr_try_fused(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, std::true_type)
{
    const char *e = std::getenv("CMI_CR_FUSED");
    if (e && e[0] == '0') return false; // (measurements: the operation-by-operation path)
    cr_fused_device(A, x, b, monitor);
    return true;
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
bool cr_try_fused(const LinearOperator &, VectorType1 &, const VectorType2 &, Monitor &, std::false_type) { return false; }

} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = detail::not_policy<LinearOperator>>
void cr(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("cr: the operator must be square");
    if (detail::cr_try_fused(A, x, b, monitor, std::integral_constant<bool, detail::use_fused_cr<LinearOperator, VectorType1, Preconditioner, Monitor>::value>())) return;
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
