// cusp/blas/blas.h -- the BLAS-1 routines cusp::krylov::cg uses, with the reference's signatures
// (cusp/blas/blas.h; generic implementations cusp/system/detail/generic/blas.h:175-220,283-340):
//   axpy(x, y, alpha)            y <- alpha*x + y
//   axpby(x, y, z, alpha, beta)  z <- alpha*x + beta*y
//   copy(x, y)   fill(x, v)   dot(x, y)   dotc(x, y)   nrm2(x)
// and the rest of the reference's BLAS-1 set for the multiply's other callers (preconditioned cg, bicgstab, cr):
//   scal(x, alpha)   xmy(x, y, z)   axpbypcz(x, y, z, out, alpha, beta, gamma)   nrm1(x)   nrmmax(x)   amax(x)
// host_memory: plain loops.  device_memory: cmi_blas_*_{f64,f32} kernels; reductions are deterministic two-stage trees whose scalar is copied back (one 8-byte D2H
// per call -- the reference's Thrust reductions synchronise the same way).
#pragma once
#include <cmath>

#include "../array1d.h"
#include "../execution_policy.h"

namespace cusp {
namespace blas {

namespace detail {

struct device_workspace { // lazily allocated per thread, freed at exit: no allocation per call
    void *ws;
    void *result; // 16 bytes: a double or a float scalar, and behind it the position amax reports
    device_workspace() : ws(nullptr), result(nullptr) {}
    ~device_workspace() { if (ws) cmi_free(ws); if (result) cmi_free(result); }
    void ensure()
    {
        if (!ws) cusp::detail::check(cmi_malloc(&ws, cmi_blas_workspace_bytes()));
        if (!result) cusp::detail::check(cmi_malloc(&result, 2 * sizeof(double)));
    }
};
inline device_workspace &workspace()
{
    static thread_local device_workspace w;
    w.ensure();
    return w;
}

template <typename X, typename Y> void same_size(const X &x, const Y &y)
{
    if (x.size() != y.size()) throw cusp::invalid_input_exception("cusp::blas: array sizes differ");
}

// C-ABI overload sets (double / float)
inline int c_axpy(int64_t n, double a, const double *x, double *y) { return cmi_blas_axpy_f64(n, a, x, y, nullptr); }
inline int c_axpy(int64_t n, float a, const float *x, float *y) { return cmi_blas_axpy_f32(n, a, x, y, nullptr); }
inline int c_axpby(int64_t n, double a, const double *x, double b, const double *y, double *z) { return cmi_blas_axpby_f64(n, a, x, b, y, z, nullptr); }
inline int c_axpby(int64_t n, float a, const float *x, float b, const float *y, float *z) { return cmi_blas_axpby_f32(n, a, x, b, y, z, nullptr); }
inline int c_copy(int64_t n, const double *x, double *y) { return cmi_blas_copy_f64(n, x, y, nullptr); }
inline int c_copy(int64_t n, const float *x, float *y) { return cmi_blas_copy_f32(n, x, y, nullptr); }
inline int c_fill(int64_t n, double v, double *y) { return cmi_blas_fill_f64(n, v, y, nullptr); }
inline int c_fill(int64_t n, float v, float *y) { return cmi_blas_fill_f32(n, v, y, nullptr); }
inline int c_dot(int64_t n, const double *x, const double *y, double *r, void *ws) { return cmi_blas_dot_f64(n, x, y, r, ws, nullptr); }
inline int c_dot(int64_t n, const float *x, const float *y, float *r, void *ws) { return cmi_blas_dot_f32(n, x, y, r, ws, nullptr); }
inline int c_nrm2(int64_t n, const double *x, double *r, void *ws) { return cmi_blas_nrm2_f64(n, x, r, ws, nullptr); }
inline int c_nrm2(int64_t n, const float *x, float *r, void *ws) { return cmi_blas_nrm2_f32(n, x, r, ws, nullptr); }

inline int c_scal(int64_t n, double a, double *x) { return cmi_blas_scal_f64(n, a, x, nullptr); }
inline int c_scal(int64_t n, float a, float *x) { return cmi_blas_scal_f32(n, a, x, nullptr); }
inline int c_xmy(int64_t n, const double *x, const double *y, double *z) { return cmi_blas_xmy_f64(n, x, y, z, nullptr); }
inline int c_xmy(int64_t n, const float *x, const float *y, float *z) { return cmi_blas_xmy_f32(n, x, y, z, nullptr); }
inline int c_axpbypcz(int64_t n, double a, const double *x, double b, const double *y, double c, const double *z, double *o) { return cmi_blas_axpbypcz_f64(n, a, x, b, y, c, z, o, nullptr); }
inline int c_axpbypcz(int64_t n, float a, const float *x, float b, const float *y, float c, const float *z, float *o) { return cmi_blas_axpbypcz_f32(n, a, x, b, y, c, z, o, nullptr); }
inline int c_asum(int64_t n, const double *x, double *r, void *ws) { return cmi_blas_asum_f64(n, x, r, ws, nullptr); }
inline int c_asum(int64_t n, const float *x, float *r, void *ws) { return cmi_blas_asum_f32(n, x, r, ws, nullptr); }
inline int c_amax(int64_t n, const double *x, double *v, int64_t *i, void *ws) { return cmi_blas_amax_f64(n, x, v, i, ws, nullptr); }
inline int c_amax(int64_t n, const float *x, float *v, int64_t *i, void *ws) { return cmi_blas_amax_f32(n, x, v, i, ws, nullptr); }

template <typename V> struct require_real {
    static_assert(std::is_same<V, double>::value || std::is_same<V, float>::value, "device_memory cusp::blas routines are implemented for float and double");
};

// ---- host ----
template <typename X, typename Y, typename S> void axpy(const X &x, Y &y, S a, host_memory)
{ for (size_t i = 0; i < x.size(); i++) y[i] = a * x[i] + y[i]; }
template <typename X, typename Y, typename Z, typename S> void axpby(const X &x, const Y &y, Z &z, S a, S b, host_memory)
{ for (size_t i = 0; i < x.size(); i++) z[i] = a * x[i] + b * y[i]; }
template <typename X, typename Y> void copy(const X &x, Y &y, host_memory)
{ for (size_t i = 0; i < x.size(); i++) y[i] = x[i]; }
template <typename X, typename S> void fill(X &x, S v, host_memory)
{ for (size_t i = 0; i < x.size(); i++) x[i] = v; }
template <typename X, typename Y> typename X::value_type dot(const X &x, const Y &y, host_memory)
{
    typename X::value_type s = 0;
    for (size_t i = 0; i < x.size(); i++) s += x[i] * y[i];
    return s;
}
// ---- device ----
template <typename X, typename Y, typename S> void axpy(const X &x, Y &y, S a, device_memory)
{ require_real<typename Y::value_type>(); cusp::detail::check(c_axpy(x.size(), a, x.data(), y.data())); }
template <typename X, typename Y, typename Z, typename S> void axpby(const X &x, const Y &y, Z &z, S a, S b, device_memory)
{ require_real<typename Z::value_type>(); cusp::detail::check(c_axpby(x.size(), a, x.data(), b, y.data(), z.data())); }
template <typename X, typename Y> void copy(const X &x, Y &y, device_memory)
{ require_real<typename Y::value_type>(); cusp::detail::check(c_copy(x.size(), x.data(), y.data())); }
template <typename X, typename S> void fill(X &x, S v, device_memory)
{ require_real<typename X::value_type>(); cusp::detail::check(c_fill(x.size(), v, x.data())); }
template <typename X, typename Y> typename X::value_type dot(const X &x, const Y &y, device_memory)
{
    typedef typename X::value_type V;
    require_real<V>();
    device_workspace &w = workspace();
    cusp::detail::check(c_dot(x.size(), x.data(), y.data(), static_cast<V *>(w.result), w.ws));
    V r;
    cusp::detail::check(cmi_memcpy_d2h(&r, w.result, sizeof(V), nullptr));
    return r;
}
template <typename X> typename X::value_type nrm2(const X &x, device_memory)
{
    typedef typename X::value_type V;
    require_real<V>();
    device_workspace &w = workspace();
    cusp::detail::check(c_nrm2(x.size(), x.data(), static_cast<V *>(w.result), w.ws));
    V r;
    cusp::detail::check(cmi_memcpy_d2h(&r, w.result, sizeof(V), nullptr));
    return r;
}
template <typename X> typename X::value_type nrm2(const X &x, host_memory)
{
    typename X::value_type s = 0;
    for (size_t i = 0; i < x.size(); i++) s += x[i] * x[i];
    return std::sqrt(s);
}

// ---- the rest of the set: host loops and device kernels ----
template <typename X, typename S> void scal(X &x, S a, host_memory) { for (size_t i = 0; i < x.size(); i++) x[i] = a * x[i]; }
template <typename X, typename S> void scal(X &x, S a, device_memory) { require_real<typename X::value_type>(); cusp::detail::check(c_scal(x.size(), a, x.data())); }
template <typename X, typename Y, typename Z> void xmy(const X &x, const Y &y, Z &z, host_memory) { for (size_t i = 0; i < x.size(); i++) z[i] = x[i] * y[i]; }
template <typename X, typename Y, typename Z> void xmy(const X &x, const Y &y, Z &z, device_memory)
{ require_real<typename Z::value_type>(); cusp::detail::check(c_xmy(x.size(), x.data(), y.data(), z.data())); }
template <typename X, typename Y, typename Z, typename O, typename S> void axpbypcz(const X &x, const Y &y, const Z &z, O &out, S a, S b, S c, host_memory)
{ for (size_t i = 0; i < x.size(); i++) out[i] = a * x[i] + b * y[i] + c * z[i]; }
template <typename X, typename Y, typename Z, typename O, typename S> void axpbypcz(const X &x, const Y &y, const Z &z, O &out, S a, S b, S c, device_memory)
{ require_real<typename O::value_type>(); cusp::detail::check(c_axpbypcz(x.size(), a, x.data(), b, y.data(), c, z.data(), out.data())); }
template <typename X> typename X::value_type nrm1(const X &x, host_memory)
{
    typename X::value_type s = 0;
    for (size_t i = 0; i < x.size(); i++) s += std::abs(x[i]);
    return s;
}
template <typename X> typename X::value_type nrm1(const X &x, device_memory)
{
    typedef typename X::value_type V;
    require_real<V>();
    device_workspace &w = workspace();
    cusp::detail::check(c_asum(x.size(), x.data(), static_cast<V *>(w.result), w.ws));
    V r;
    cusp::detail::check(cmi_memcpy_d2h(&r, w.result, sizeof(V), nullptr));
    return r;
}
template <typename X> void max_abs(const X &x, typename X::value_type &value, size_t &index, host_memory)
{
    value = 0; index = 0;
    bool any = false;
    for (size_t i = 0; i < x.size(); i++) { const typename X::value_type a = std::abs(x[i]); if (!any || a > value) { value = a; index = i; any = true; } }
}
template <typename X> void max_abs(const X &x, typename X::value_type &value, size_t &index, device_memory)
{
    typedef typename X::value_type V;
    require_real<V>();
    device_workspace &w = workspace();
    int64_t *pos = reinterpret_cast<int64_t *>(static_cast<char *>(w.result) + sizeof(double));
    cusp::detail::check(c_amax(x.size(), x.data(), static_cast<V *>(w.result), pos, w.ws));
    int64_t p = 0;
    cusp::detail::check(cmi_memcpy_d2h(&value, w.result, sizeof(V), nullptr));
    cusp::detail::check(cmi_memcpy_d2h(&p, pos, sizeof(int64_t), nullptr));
    index = static_cast<size_t>(p);
}

// ---- sharded vectors (cusp/distributed/vector.h): element-wise work on this rank's slice, reductions all-reduced --------------
// The slice is an array1d_view in the local space, so the loops / kernels above do the work.  Reductions: the local partial as a
// DOUBLE (device: left in device memory by the library's deterministic two-stage reduction, all-reduced there by RCCL, then the one
// 8-byte read the single-GPU routines also make; host: summed in rank order by the star) -- every rank returns the same bits.
template <typename X, typename Y, typename S, typename L> void axpy(const X &x, Y &y, S a, distributed_memory<L>)
{ auto xl = x.local(); auto yl = y.local(); axpy(xl, yl, a, L()); }
template <typename X, typename Y, typename Z, typename S, typename L> void axpby(const X &x, const Y &y, Z &z, S a, S b, distributed_memory<L>)
{ auto xl = x.local(); auto yl = y.local(); auto zl = z.local(); axpby(xl, yl, zl, a, b, L()); }
template <typename X, typename Y, typename L> void copy(const X &x, Y &y, distributed_memory<L>)
{ auto xl = x.local(); auto yl = y.local(); copy(xl, yl, L()); }
template <typename X, typename S, typename L> void fill(X &x, S v, distributed_memory<L>)
{ auto xl = x.local(); fill(xl, v, L()); }
template <typename X, typename S, typename L> void scal(X &x, S a, distributed_memory<L>)
{ auto xl = x.local(); scal(xl, a, L()); }
template <typename X, typename Y, typename Z, typename L> void xmy(const X &x, const Y &y, Z &z, distributed_memory<L>)
{ auto xl = x.local(); auto yl = y.local(); auto zl = z.local(); xmy(xl, yl, zl, L()); }
template <typename X, typename Y, typename Z, typename O, typename S, typename L> void axpbypcz(const X &x, const Y &y, const Z &z, O &out, S a, S b, S c, distributed_memory<L>)
{ auto xl = x.local(); auto yl = y.local(); auto zl = z.local(); auto ol = out.local(); axpbypcz(xl, yl, zl, ol, a, b, c, L()); }
inline int c_dotd(int64_t n, const double *x, const double *y, double *r, void *ws) { return cmi_blas_dot_f64(n, x, y, r, ws, nullptr); }
inline int c_dotd(int64_t n, const float *x, const float *y, double *r, void *ws) { return cmi_blas_dotd_f32(n, x, y, r, ws, nullptr); }
template <typename X, typename Y> double dot_all(const X &x, const Y &y, host_memory)
{
    double s = 0;
    for (size_t i = 0; i < x.size(); i++) s += (double)x.data()[i] * (double)y.data()[i];
    x.comm().allreduce_sum(&s, 1, host_memory());
    return s;
}
template <typename X, typename Y> double dot_all(const X &x, const Y &y, device_memory)
{
    require_real<typename X::value_type>();
    device_workspace &w = workspace();
    double *res = static_cast<double *>(w.result);
    cusp::detail::check(c_dotd(x.size(), x.data(), y.data(), res, w.ws));
    x.comm().allreduce_sum(res, 1, device_memory());
    double r;
    cusp::detail::check(cmi_memcpy_d2h(&r, res, sizeof(double), nullptr));
    return r;
}
template <typename X, typename Y, typename L> typename X::value_type dot(const X &x, const Y &y, distributed_memory<L>)
{ return static_cast<typename X::value_type>(dot_all(x, y, L())); }
template <typename X, typename L> typename X::value_type nrm2(const X &x, distributed_memory<L>)
{ return static_cast<typename X::value_type>(std::sqrt(dot_all(x, x, L()))); }

} // namespace detail

template <typename X, typename Y, typename S> void axpy(const X &x, Y &y, S alpha)
{ detail::same_size(x, y); detail::axpy(x, y, static_cast<typename Y::value_type>(alpha), typename Y::memory_space()); }
template <typename X, typename Y, typename Z, typename S1, typename S2> void axpby(const X &x, const Y &y, Z &z, S1 alpha, S2 beta)
{
    typedef typename Z::value_type V;
    detail::same_size(x, y); detail::same_size(x, z);
    detail::axpby(x, y, z, static_cast<V>(alpha), static_cast<V>(beta), typename Z::memory_space());
}
template <typename X, typename Y> void copy(const X &x, Y &y) { detail::same_size(x, y); detail::copy(x, y, typename Y::memory_space()); }
template <typename X, typename S> void fill(X &x, S v) { detail::fill(x, static_cast<typename X::value_type>(v), typename X::memory_space()); }
template <typename X, typename Y> typename X::value_type dot(const X &x, const Y &y) { detail::same_size(x, y); return detail::dot(x, y, typename X::memory_space()); }
template <typename X, typename Y> typename X::value_type dotc(const X &x, const Y &y) { return dot(x, y); } // real types: conj is the identity
template <typename X> typename X::value_type nrm2(const X &x) { return detail::nrm2(x, typename X::memory_space()); }
template <typename X, typename S> void scal(X &x, S alpha) { detail::scal(x, static_cast<typename X::value_type>(alpha), typename X::memory_space()); }
template <typename T, typename M, typename S> void scal(const array1d_view<T, M> &x, S alpha) { array1d_view<T, M> v(x); detail::scal(v, static_cast<typename array1d_view<T, M>::value_type>(alpha), M()); } // (a view passed as a temporary refers to the same elements)
template <typename X, typename Y, typename Z> void xmy(const X &x, const Y &y, Z &z) { detail::same_size(x, y); detail::same_size(x, z); detail::xmy(x, y, z, typename Z::memory_space()); }
template <typename X, typename Y, typename Z, typename O, typename S1, typename S2, typename S3>
void axpbypcz(const X &x, const Y &y, const Z &z, O &out, S1 alpha, S2 beta, S3 gamma)
{
    typedef typename O::value_type V;
    detail::same_size(x, y); detail::same_size(x, z); detail::same_size(x, out);
    detail::axpbypcz(x, y, z, out, static_cast<V>(alpha), static_cast<V>(beta), static_cast<V>(gamma), typename O::memory_space());
}
template <typename X> typename X::value_type nrm1(const X &x) { return detail::nrm1(x, typename X::memory_space()); }
template <typename X> typename X::value_type nrmmax(const X &x)
{
    typename X::value_type v; size_t i;
    detail::max_abs(x, v, i, typename X::memory_space());
    return v;
}
template <typename X> int amax(const X &x) // position of the first entry of largest magnitude (reference: an int)
{
    typename X::value_type v; size_t i;
    detail::max_abs(x, v, i, typename X::memory_space());
    return static_cast<int>(i);
}

// ---- execution-policy overloads (reference cusp/blas/blas.h: every routine also takes a policy first) ----
// A policy derived from cusp::execution_policy<Derived> reaches a user overload `axpy(my_policy&, ...)` by
// ADL (testing/blas.cu:752-1208); a policy without one gets the memory-space dispatch above.
namespace detail {
namespace policy_default {
template <typename D> using if_policy = typename std::enable_if<std::is_base_of<cusp::execution_policy<D>, D>::value>::type;
template <typename D, typename X, typename Y, typename S, typename = if_policy<D>> void axpy(D &, const X &x, Y &y, S a) { cusp::blas::axpy(x, y, a); }
template <typename D, typename X, typename Y, typename Z, typename S1, typename S2, typename = if_policy<D>>
void axpby(D &, const X &x, const Y &y, Z &z, S1 a, S2 b) { cusp::blas::axpby(x, y, z, a, b); }
template <typename D, typename X, typename Y, typename = if_policy<D>> void copy(D &, const X &x, Y &y) { cusp::blas::copy(x, y); }
template <typename D, typename X, typename S, typename = if_policy<D>> void fill(D &, X &x, S v) { cusp::blas::fill(x, v); }
template <typename D, typename X, typename Y, typename = if_policy<D>> typename X::value_type dot(D &, const X &x, const Y &y) { return cusp::blas::dot(x, y); }
template <typename D, typename X, typename Y, typename = if_policy<D>> typename X::value_type dotc(D &, const X &x, const Y &y) { return cusp::blas::dotc(x, y); }
template <typename D, typename X, typename = if_policy<D>> typename X::value_type nrm2(D &, const X &x) { return cusp::blas::nrm2(x); }
template <typename D, typename X, typename S, typename = if_policy<D>> void scal(D &, X &x, S a) { cusp::blas::scal(x, a); }
template <typename D, typename X, typename Y, typename Z, typename = if_policy<D>> void xmy(D &, const X &x, const Y &y, Z &z) { cusp::blas::xmy(x, y, z); }
template <typename D, typename X, typename Y, typename Z, typename O, typename S1, typename S2, typename S3, typename = if_policy<D>>
void axpbypcz(D &, const X &x, const Y &y, const Z &z, O &o, S1 a, S2 b, S3 c) { cusp::blas::axpbypcz(x, y, z, o, a, b, c); }
template <typename D, typename X, typename = if_policy<D>> typename X::value_type nrm1(D &, const X &x) { return cusp::blas::nrm1(x); }
template <typename D, typename X, typename = if_policy<D>> typename X::value_type nrmmax(D &, const X &x) { return cusp::blas::nrmmax(x); }
template <typename D, typename X, typename = if_policy<D>> int amax(D &, const X &x) { return cusp::blas::amax(x); }
} // namespace policy_default
} // namespace detail

template <typename D, typename X, typename Y, typename S> void axpy(const cusp::execution_policy<D> &exec, const X &x, Y &y, S alpha)
{ using detail::policy_default::axpy; axpy(const_cast<D &>(exec.derived()), x, y, alpha); }
template <typename D, typename X, typename Y, typename Z, typename S1, typename S2>
void axpby(const cusp::execution_policy<D> &exec, const X &x, const Y &y, Z &z, S1 alpha, S2 beta)
{ using detail::policy_default::axpby; axpby(const_cast<D &>(exec.derived()), x, y, z, alpha, beta); }
template <typename D, typename X, typename Y> void copy(const cusp::execution_policy<D> &exec, const X &x, Y &y)
{ using detail::policy_default::copy; copy(const_cast<D &>(exec.derived()), x, y); }
template <typename D, typename X, typename S> void fill(const cusp::execution_policy<D> &exec, X &x, S v)
{ using detail::policy_default::fill; fill(const_cast<D &>(exec.derived()), x, v); }
template <typename D, typename X, typename Y> typename X::value_type dot(const cusp::execution_policy<D> &exec, const X &x, const Y &y)
{ using detail::policy_default::dot; return dot(const_cast<D &>(exec.derived()), x, y); }
template <typename D, typename X, typename Y> typename X::value_type dotc(const cusp::execution_policy<D> &exec, const X &x, const Y &y)
{ using detail::policy_default::dotc; return dotc(const_cast<D &>(exec.derived()), x, y); }
template <typename D, typename X> typename X::value_type nrm2(const cusp::execution_policy<D> &exec, const X &x)
{ using detail::policy_default::nrm2; return nrm2(const_cast<D &>(exec.derived()), x); }
template <typename D, typename X, typename S> void scal(const cusp::execution_policy<D> &exec, X &x, S alpha)
{ using detail::policy_default::scal; scal(const_cast<D &>(exec.derived()), x, alpha); }
template <typename D, typename X, typename Y, typename Z> void xmy(const cusp::execution_policy<D> &exec, const X &x, const Y &y, Z &z)
{ using detail::policy_default::xmy; xmy(const_cast<D &>(exec.derived()), x, y, z); }
template <typename D, typename X, typename Y, typename Z, typename O, typename S1, typename S2, typename S3>
void axpbypcz(const cusp::execution_policy<D> &exec, const X &x, const Y &y, const Z &z, O &out, S1 alpha, S2 beta, S3 gamma)
{ using detail::policy_default::axpbypcz; axpbypcz(const_cast<D &>(exec.derived()), x, y, z, out, alpha, beta, gamma); }
template <typename D, typename X> typename X::value_type nrm1(const cusp::execution_policy<D> &exec, const X &x)
{ using detail::policy_default::nrm1; return nrm1(const_cast<D &>(exec.derived()), x); }
template <typename D, typename X> typename X::value_type nrmmax(const cusp::execution_policy<D> &exec, const X &x)
{ using detail::policy_default::nrmmax; return nrmmax(const_cast<D &>(exec.derived()), x); }
template <typename D, typename X> int amax(const cusp::execution_policy<D> &exec, const X &x)
{ using detail::policy_default::amax; return amax(const_cast<D &>(exec.derived()), x); }

} // namespace blas
} // namespace cusp
