// cusp/krylov/bicgstab.h -- cusp::krylov::bicgstab(A, x, b[, monitor[, M]]): the stabilised bi-conjugate gradient method for
// non-symmetric systems, right-preconditioned (reference cusp/krylov/bicgstab.h, detail/bicgstab.inl:48-128 -- the same operation order, so the
// same iteration counts: alpha from <r*, A M p>, the early exit on s, omega from <A M s, s> / <A M s, A M s>, x and p updated with axpbypcz).
// Another CALLER of the hot path: two cusp::multiply(A, ., .) per iteration through A's plan, everything else cusp::blas (device_memory: the
// library's kernels; the scalars make four 8-byte reads per iteration, as the reference's Thrust reductions do).  Any memory space, any format,
// any M with operator()(x, y) or a matrix.
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

// ---- device_memory, identity preconditioner: the fused iteration (csrc/blas1_extra.hip cmi_bicgstab_*) --------------------------------
// y <- A v and *out <- <y, w> for any w: CSR in ONE launch through the container's plan (cmi_spmv_csr_dot_plan_*), the other formats multiply + dot
inline int csr_dot_w_(const cmi_plan *pl, const int *Ap, const int *Aj, const double *Ax, const double *v, double *y, const double *w, double *out, void *ws)
{ return cmi_spmv_csr_dot_plan_f64(pl, Ap, Aj, Ax, v, y, w, out, ws, nullptr); }
inline int csr_dot_w_(const cmi_plan *pl, const int *Ap, const int *Aj, const float *Ax, const float *v, float *y, const float *w, double *out, void *ws)
{ return cmi_spmv_csr_dot_plan_f32(pl, Ap, Aj, Ax, v, y, w, out, ws, nullptr); }
template <typename A, typename V> void multiply_dot_w(const A &a, const V &v, V &y, const V &w, double *out, void *ws, cusp::csr_format)
{
    cusp::detail::require_int_index<A>();
    if (const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0)) {
        cusp::detail::check(csr_dot_w_(pl, a.row_offsets.data(), a.column_indices.data(), a.values.data(), v.data(), y.data(), w.data(), out, ws));
        return;
    }
    cusp::multiply(a, v, y);
    cusp::detail::check(dotd_(y.size(), y.data(), w.data(), out, ws));
}
template <typename A, typename V, typename Format> void multiply_dot_w(const A &a, const V &v, V &y, const V &w, double *out, void *ws, Format)
{
    cusp::multiply(a, v, y);
    cusp::detail::check(dotd_(y.size(), y.data(), w.data(), out, ws));
}
inline int bicg_s_(size_t n, const double *rho, const double *d1, const double *r, const double *AMp, double *s, double *ss, double *m, void *ws)
{ return cmi_bicgstab_s_f64(n, rho, d1, r, AMp, s, ss, m, ws, nullptr); }
inline int bicg_s_(size_t n, const double *rho, const double *d1, const float *r, const float *AMp, float *s, double *ss, double *m, void *ws)
{ return cmi_bicgstab_s_f32(n, rho, d1, r, AMp, s, ss, m, ws, nullptr); }
inline int bicg_xr_(size_t n, const double *rho, const double *d1, const double *d2, const double *d3, const double *p, const double *s, const double *AMs,
                    const double *rs, double *x, double *r, double *rho_new, double *rr, double *m, void *ws)
{ return cmi_bicgstab_xr_f64(n, rho, d1, d2, d3, p, s, AMs, rs, x, r, rho_new, rr, m, ws, nullptr); }
inline int bicg_xr_(size_t n, const double *rho, const double *d1, const double *d2, const double *d3, const float *p, const float *s, const float *AMs,
                    const float *rs, float *x, float *r, double *rho_new, double *rr, double *m, void *ws)
{ return cmi_bicgstab_xr_f32(n, rho, d1, d2, d3, p, s, AMs, rs, x, r, rho_new, rr, m, ws, nullptr); }
inline int bicg_p_(size_t n, const double *rho_new, const double *rho, const double *d1, const double *d2, const double *d3, const double *r, const double *AMp, double *p)
{ return cmi_bicgstab_p_f64(n, rho_new, rho, d1, d2, d3, r, AMp, p, nullptr); }
inline int bicg_p_(size_t n, const double *rho_new, const double *rho, const double *d1, const double *d2, const double *d3, const float *r, const float *AMp, float *p)
{ return cmi_bicgstab_p_f32(n, rho_new, rho, d1, d2, d3, r, AMp, p, nullptr); }
inline int axpy_ratio_(size_t n, const double *num, const double *den, const double *x, double *y) { return cmi_blas_axpy_ratio_f64(n, num, den, x, y, nullptr); }
inline int axpy_ratio_(size_t n, const double *num, const double *den, const float *x, float *y) { return cmi_blas_axpy_ratio_f32(n, num, den, x, y, nullptr); }

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void bicgstab_fused_device(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type T;
    typedef cusp::array1d<T, cusp::device_memory> Vec;
    const size_t N = A.num_rows;
    Vec p(N), r(N), r_star(N), s(N), AMp(N), AMs(N);
    cusp::array1d<double, cusp::device_memory> scalars(7); // rho[0], rho[1], d1 = <r*, A p>, d2 = <A s, s>, d3 = <A s, A s>, <s, s>, <r, r>
    double *rho[2] = {scalars.data(), scalars.data() + 1};
    double *d1 = scalars.data() + 2, *d2 = scalars.data() + 3, *d3 = scalars.data() + 4, *ss = scalars.data() + 5, *rr = scalars.data() + 6;
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    pinned_scalar ss_host, rr_host;
    cusp::multiply(A, x, r);
    cusp::blas::axpby(b, r, r, T(1), T(-1));
    cusp::blas::copy(r, p);
    cusp::blas::copy(r, r_star);
    cusp::detail::check(dotd_(N, r_star.data(), r.data(), rho[0], w.ws));
    cusp::detail::check(dotd_(N, r.data(), r.data(), rr, w.ws));
    rr_host.fetch(rr);
    int cur = 0;
    for (;;) {
        multiply_dot_w(A, p, AMp, r_star, d1, w.ws, typename LinearOperator::format());                    // speculative: A p and <r*, A p>
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) break;  // ||r||
        cusp::detail::check(bicg_s_(N, rho[cur], d1, r.data(), AMp.data(), s.data(), ss, ss_host.host, w.ws));
        ss_host.record();
        multiply_dot_w(A, s, AMs, s, d2, w.ws, typename LinearOperator::format());                         // speculative: A s and <A s, s>
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(ss_host.wait())))) {       // ||s||: half a step is enough
            cusp::detail::check(axpy_ratio_(N, rho[cur], d1, p.data(), x.data()));
            break;
        }
        cusp::detail::check(dotd_(N, AMs.data(), AMs.data(), d3, w.ws));
        cusp::detail::check(bicg_xr_(N, rho[cur], d1, d2, d3, p.data(), s.data(), AMs.data(), r_star.data(), x.data(), r.data(), rho[cur ^ 1], rr, rr_host.host, w.ws));
        rr_host.record();
        cusp::detail::check(bicg_p_(N, rho[cur ^ 1], rho[cur], d1, d2, d3, r.data(), AMp.data(), p.data()));
        cur ^= 1;
        ++monitor;
    }
    cusp::detail::check(cmi_device_synchronize()); // the discarded multiply must not outlive its vectors
}

template <typename A> auto has_format_tag(const A *) -> decltype(typename A::format(), std::true_type());
inline std::false_type has_format_tag(...);

template <typename A, typename X, typename M, typename Mon> struct use_fused_bicgstab {
    typedef typename A::value_type T;
    static const bool value = std::is_same<typename A::memory_space, cusp::device_memory>::value &&
                              (std::is_same<T, double>::value || std::is_same<T, float>::value) && std::is_same<typename X::value_type, T>::value &&
                              is_identity<M>::value && decltype(has_finished_norm(static_cast<Mon *>(nullptr)))::value &&
                              decltype(has_format_tag(static_cast<const A *>(nullptr)))::value;
};

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
bool bicgstab_try_fused(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &, std::true_type)
{
    const char *e = std::getenv("CMI_BICGSTAB_FUSED");
    if (e && e[0] == '0') return false; // (measurements: the operation-by-operation path)
    bicgstab_fused_device(A, x, b, monitor);
    return true;
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
bool bicgstab_try_fused(const LinearOperator &, VectorType1 &, const VectorType2 &, Monitor &, Preconditioner &, std::false_type) { return false; }

} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = detail::not_policy<LinearOperator>>
void bicgstab(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("bicgstab: the operator must be square");
    if (detail::bicgstab_try_fused(A, x, b, monitor, M, std::integral_constant<bool, detail::use_fused_bicgstab<LinearOperator, VectorType1, Preconditioner, Monitor>::value>())) return;
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
