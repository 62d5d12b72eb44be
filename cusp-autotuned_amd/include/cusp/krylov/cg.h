// cusp/krylov/cg.h -- (preconditioned) conjugate gradients, the caller of the SpMV hot path
// (reference cusp/krylov/cg.h, cusp/krylov/detail/cg.inl:41-107: same operations in the same order,
// so the residual history matches the reference's, e.g. docs/quickstart.md:72-87).
// One cusp::multiply per iteration -> cmi_spmv_* on device_memory; vector updates -> cusp::blas.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "../array1d.h"
#include "../blas/blas.h"
#include "../execution_policy.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"

namespace cusp {
namespace precond { template <typename ValueType, typename MemorySpace> class diagonal; }
namespace krylov {

namespace detail {
// the overloads without a policy must not swallow cg(policy, A, x, b[, monitor]) calls
template <typename T> struct is_policy : std::is_base_of<cusp::execution_policy<T>, T> {};
// (the solvers skip the copy an identity preconditioner would make)
template <typename M> struct is_identity : std::false_type {};
template <typename V, typename S, typename I> struct is_identity<cusp::identity_operator<V, S, I>> : std::true_type {};
template <typename V, typename S, typename I> struct is_identity<const cusp::identity_operator<V, S, I>> : std::true_type {};
template <typename T> using not_policy = typename std::enable_if<!is_policy<T>::value>::type;

// z <- M r for a matrix-like preconditioner or a linear operator with operator()
template <typename M, typename X, typename Y> auto apply(const M &m, const X &x, Y &y, int) -> decltype(m(x, y), void()) { m(x, y); }
template <typename M, typename X, typename Y> void apply(const M &m, const X &x, Y &y, long) { cusp::multiply(m, x, y); }
} // namespace detail

namespace detail {

// Fused unpreconditioned CG on the device (f64 and f32): z == r is folded away, alpha and beta stay in device
// memory, and an iteration is
//     cmi_spmv_csr_dot_f64 (y <- A p and <y,p> in one pass; other formats: SpMV + cmi_blas_dot)
//     cmi_cg_update_f64      (r, <r,r> in one pass)
//     cmi_cg_direction_x_f64 (x <- x + alpha p, p <- r + beta p in one pass)
// i.e. 3 launches, 8 vector passes and ONE host read (the convergence check) instead of cg.inl's 7 passes and 3 host syncs.
// The host read needs no copy: cmi_cg_update's reduction also writes <r,r> into page-locked host
// memory, and the host waits for the event behind it only AFTER it has queued the next iteration's SpMV (which reads p and writes the scratch y -- no solver state -- so
// running it speculatively is harmless if the monitor then stops), so the device never idles on the
// round trip.  Per-element arithmetic unchanged; the residual history agrees with the plain path to
// rounding.
template <typename Monitor> auto has_finished_norm(Monitor *m) -> decltype(m->finished_norm(typename Monitor::Real()), std::true_type());
std::false_type has_finished_norm(...);

struct pinned_scalar { // 8 page-locked bytes + the event that says they have landed
    double *host;
    void *event;
    pinned_scalar() : host(nullptr), event(nullptr)
    {
        void *h = nullptr;
        cusp::detail::check(cmi_malloc_host(&h, sizeof(double)));
        host = static_cast<double *>(h);
        const int st = cmi_event_create(&event);
        if (st != CMI_SUCCESS) { cmi_free_host(host); cusp::detail::check(st); }
    }
    ~pinned_scalar() { if (event) cmi_event_destroy(event); if (host) cmi_free_host(host); }
    pinned_scalar(const pinned_scalar &) = delete;
    pinned_scalar &operator=(const pinned_scalar &) = delete;
    void fetch(const double *dev) // queue the copy; returns at once
    {
        cusp::detail::check(cmi_memcpy_d2h_async(host, dev, sizeof(double), nullptr));
        record();
    }
    void record() { cusp::detail::check(cmi_event_record(event, nullptr)); } // behind a kernel that wrote *host itself
    double wait() { cusp::detail::check(cmi_event_synchronize(event)); return *host; }
};

// y <- A p and *yp <- <y, p>: one fused launch for CSR, SpMV + dot for the other formats
inline int csr_dot_(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, double *yp, void *ws)
{ return cmi_spmv_csr_dot_f64(r, c, n, Ap, Aj, Ax, x, y, x, yp, ws, cusp::detail::forced_config(), nullptr); }
inline int csr_dot_(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, double *yp, void *ws)
{ return cmi_spmv_csr_dot_f32(r, c, n, Ap, Aj, Ax, x, y, x, yp, ws, cusp::detail::forced_config(), nullptr); }
inline int csr_dot_plan_(const cmi_plan *pl, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, double *yp, void *ws)
{ return cmi_spmv_csr_dot_plan_f64(pl, Ap, Aj, Ax, x, y, x, yp, ws, nullptr); }
inline int csr_dot_plan_(const cmi_plan *pl, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, double *yp, void *ws)
{ return cmi_spmv_csr_dot_plan_f32(pl, Ap, Aj, Ax, x, y, x, yp, ws, nullptr); }
// CSR, double or float: the fused launch, steered by the container's plan where there is one
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, cusp::csr_format)
{
    cusp::detail::require_int_index<A>();
    if (const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0)) {
        cusp::detail::check(csr_dot_plan_(pl, a.row_offsets.data(), a.column_indices.data(), a.values.data(), p.data(), y.data(), yp, ws));
        return;
    }
    cusp::detail::check(csr_dot_(a.num_rows, a.num_cols, a.num_entries, a.row_offsets.data(), a.column_indices.data(), a.values.data(), p.data(),
                                 y.data(), yp, ws));
}
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, cusp::ell_format)
{
    cusp::detail::require_int_index<A>();
    cusp::detail::check(cmi_spmv_ell_dot_f64(a.num_rows, a.num_cols, a.column_indices.num_cols, a.column_indices.pitch, cusp::detail::data_of(a.column_indices),
                                             cusp::detail::data_of(a.values), cusp::detail::row_lengths_of(a, 0), p.data(), y.data(), p.data(), yp, ws,
                                             cusp::detail::forced_config(), nullptr));
}
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, cusp::dia_format)
{
    cusp::detail::require_int_index<A>();
    cusp::detail::check(cmi_spmv_dia_dot_f64(a.num_rows, a.num_cols, a.values.num_cols, a.values.pitch, a.diagonal_offsets.data(),
                                             cusp::detail::data_of(a.values), p.data(), y.data(), p.data(), yp, ws, cusp::detail::forced_config(), nullptr));
}
// COO through its plan: sorted entries run the CSR kernel's fused dot on the plan's row offsets
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, cusp::coo_format)
{
    cusp::detail::require_int_index<A>();
    if (const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0)) {
        cusp::detail::check(cmi_spmv_coo_dot_plan_f64(pl, a.row_indices.data(), a.column_indices.data(), a.values.data(), p.data(), y.data(), p.data(), yp, ws, nullptr));
        return;
    }
    cusp::multiply(a, p, y);
    cusp::detail::check(cmi_blas_dot_f64(a.num_rows, y.data(), p.data(), yp, ws, nullptr));
}
// HYB whose COO part is empty (the tuned width rule keeps regular matrices entirely in the ELL part): the ELL kernel's fused dot
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, cusp::hyb_format)
{
    if (a.coo.num_entries == 0 && a.ell.column_indices.pitch == a.ell.values.pitch) {
        multiply_dot(a.ell, p, y, yp, ws, cusp::ell_format());
        return;
    }
    if (const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0)) { // one launch through the plan: its fused dot
        cusp::detail::check(cmi_spmv_hyb_dot_plan_f64(pl, a.ell.column_indices.pitch, cusp::detail::data_of(a.ell.column_indices), cusp::detail::data_of(a.ell.values),
                                                      a.coo.row_indices.data(), a.coo.column_indices.data(), a.coo.values.data(), p.data(), y.data(), p.data(), yp, ws, nullptr));
        return;
    }
    cusp::multiply(a, p, y);
    cusp::detail::check(cmi_blas_dot_f64(a.num_rows, y.data(), p.data(), yp, ws, nullptr));
}
template <typename A, typename V, typename Format> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, Format)
{
    cusp::multiply(a, p, y);
    cusp::detail::check(cmi_blas_dot_f64(a.num_rows, y.data(), p.data(), yp, ws, nullptr));
}
template <typename A, typename V> auto multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, int) -> decltype(typename A::format(), void())
{
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    multiply_dot(a, p, y, yp, ws, typename A::format());
}
template <typename A, typename V> void multiply_dot(const A &a, const V &p, V &y, double *yp, void *ws, long) // a linear operator without a format
{
    cusp::multiply(a, p, y);
    cusp::detail::check(cmi_blas_dot_f64(y.size(), y.data(), p.data(), yp, ws, nullptr));
}

// Fold-ahead form (CSR through its plan): the SpMV leaves the per-tile partials of <y, p> in the workspace and the update
// kernel folds them itself; likewise <r, r> is folded at the front of the direction kernel -- no fold launches in between
// (cmi_cg_update_fold_*, cusp_mi355x.h); opt-in, see fold_ahead_enabled().  Returns the partial count, 0 when this operator / plan
// cannot (then *yp is not set).
inline int csr_dot_partials_(const cmi_plan *pl, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, void *ws, int *np)
{ return cmi_spmv_csr_dot_plan_partials_f64(pl, Ap, Aj, Ax, x, y, x, ws, np, nullptr); }
inline int csr_dot_partials_(const cmi_plan *pl, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, void *ws, int *np)
{ return cmi_spmv_csr_dot_plan_partials_f32(pl, Ap, Aj, Ax, x, y, x, ws, np, nullptr); }
inline bool fold_ahead_enabled()
{
    // opt-in ($CMI_CG_FOLD_AHEAD=1): measured 268.8 vs 272.7 us per iteration on the headline matrix (-1.4 %), 252.5 vs 254.0
    // with the 16-bit column copy -- inside the box-to-box spread, so the default stays the five-launch iteration, which has
    // no workgroup waiting on another
    static const bool on = [] { const char *e = std::getenv("CMI_CG_FOLD_AHEAD"); return e && e[0] == '1'; }();
    return on;
}
template <typename A, typename V> int multiply_dot_partials(const A &a, const V &p, V &y, void *ws, cusp::csr_format)
{
    cusp::detail::require_int_index<A>();
    if (!fold_ahead_enabled() || p.size() != a.num_cols || y.size() != a.num_rows) return 0;
    const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0);
    if (!pl) return 0;
    int np = 0;
    cusp::detail::check(csr_dot_partials_(pl, a.row_offsets.data(), a.column_indices.data(), a.values.data(), p.data(), y.data(), ws, &np));
    if (np > 0) return np;
    return -1; // y is computed, but this plan's kernel left no partials: the caller adds the separate dot
}
template <typename A, typename V, typename Format> int multiply_dot_partials(const A &, const V &, V &, void *, Format) { return 0; }
template <typename A, typename V> auto multiply_dot_partials_any(const A &a, const V &p, V &y, void *ws, int) -> decltype(typename A::format(), int())
{
    return multiply_dot_partials(a, p, y, ws, typename A::format());
}
template <typename A, typename V> int multiply_dot_partials_any(const A &, const V &, V &, void *, long) { return 0; }
inline int cg_update_fold_(size_t n, const double *rz, double *yp, int np, const double *y, double *r, void *ws, int *np_rr)
{ return cmi_cg_update_fold_f64(n, rz, yp, np, y, r, ws, np_rr, nullptr); }
inline int cg_update_fold_(size_t n, const double *rz, double *yp, int np, const float *y, float *r, void *ws, int *np_rr)
{ return cmi_cg_update_fold_f32(n, rz, yp, np, y, r, ws, np_rr, nullptr); }
inline int cg_direction_x_fold_(size_t n, double *rn, double *mirror, int np, const double *ro, const double *yp, const double *r, double *p, double *x, void *ws)
{ return cmi_cg_direction_x_fold_f64(n, rn, mirror, np, ro, yp, r, p, x, ws, nullptr); }
inline int cg_direction_x_fold_(size_t n, double *rn, double *mirror, int np, const double *ro, const double *yp, const float *r, float *p, float *x, void *ws)
{ return cmi_cg_direction_x_fold_f32(n, rn, mirror, np, ro, yp, r, p, x, ws, nullptr); }

// the fused steps by value type (scalars are doubles in device memory for both)
inline int cg_update_(size_t n, const double *rz, const double *yp, const double *y, double *r, double *rr, double *mirror, void *ws)
{ return cmi_cg_update_f64(n, rz, yp, nullptr, y, nullptr, r, rr, mirror, ws, nullptr); }
inline int cg_update_(size_t n, const double *rz, const double *yp, const float *y, float *r, double *rr, double *mirror, void *ws)
{ return cmi_cg_update_f32(n, rz, yp, nullptr, y, nullptr, r, rr, mirror, ws, nullptr); }
inline int cg_direction_x_(size_t n, const double *rn, const double *ro, const double *yp, const double *r, double *p, double *x)
{ return cmi_cg_direction_x_f64(n, rn, ro, yp, r, p, x, nullptr); }
inline int cg_direction_x_(size_t n, const double *rn, const double *ro, const double *yp, const float *r, float *p, float *x)
{ return cmi_cg_direction_x_f32(n, rn, ro, yp, r, p, x, nullptr); }
inline int dotd_(size_t n, const double *x, const double *y, double *res, void *ws) { return cmi_blas_dot_f64(n, x, y, res, ws, nullptr); }
inline int dotd_(size_t n, const float *x, const float *y, double *res, void *ws) { return cmi_blas_dotd_f32(n, x, y, res, ws, nullptr); }

// y <- A p and *yp <- <y, p> for either value type: f64 matrices fuse the dot into the SpMV where the format can,
// f32 runs the SpMV and a dot that accumulates in double
template <typename A, typename V> void multiply_dot_any(const A &a, const V &p, V &y, double *yp, void *ws, std::true_type /* double */)
{
    multiply_dot(a, p, y, yp, ws, 0);
}
template <typename A, typename V> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, cusp::csr_format)
{
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    multiply_dot(a, p, y, yp, ws, cusp::csr_format()); // CSR fuses the dot for float too (the scalar stays a double)
}
template <typename A, typename V> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, cusp::ell_format)
{
    cusp::detail::require_int_index<A>();
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    cusp::detail::check(cmi_spmv_ell_dot_f32(a.num_rows, a.num_cols, a.column_indices.num_cols, a.column_indices.pitch, cusp::detail::data_of(a.column_indices),
                                             cusp::detail::data_of(a.values), cusp::detail::row_lengths_of(a, 0), p.data(), y.data(), p.data(), yp, ws,
                                             cusp::detail::forced_config(), nullptr));
}
template <typename A, typename V> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, cusp::dia_format)
{
    cusp::detail::require_int_index<A>();
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    cusp::detail::check(cmi_spmv_dia_dot_f32(a.num_rows, a.num_cols, a.values.num_cols, a.values.pitch, a.diagonal_offsets.data(),
                                             cusp::detail::data_of(a.values), p.data(), y.data(), p.data(), yp, ws, cusp::detail::forced_config(), nullptr));
}
template <typename A, typename V> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, cusp::coo_format)
{
    cusp::detail::require_int_index<A>();
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    if (const cmi_plan *pl = cusp::detail::plan_of(a, nullptr, 0)) {
        cusp::detail::check(cmi_spmv_coo_dot_plan_f32(pl, a.row_indices.data(), a.column_indices.data(), a.values.data(), p.data(), y.data(), p.data(), yp, ws, nullptr));
        return;
    }
    cusp::multiply(a, p, y);
    cusp::detail::check(dotd_(y.size(), y.data(), p.data(), yp, ws));
}
template <typename A, typename V> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, cusp::hyb_format)
{
    cusp::detail::require_int_index<A>();
    if (p.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cg: vector sizes do not match the matrix");
    if (const cmi_plan *pl = (a.ell.column_indices.pitch == a.ell.values.pitch) ? cusp::detail::plan_of(a, nullptr, 0) : nullptr) {
        cusp::detail::check(cmi_spmv_hyb_dot_plan_f32(pl, a.ell.column_indices.pitch, cusp::detail::data_of(a.ell.column_indices), cusp::detail::data_of(a.ell.values),
                                                      a.coo.row_indices.data(), a.coo.column_indices.data(), a.coo.values.data(), p.data(), y.data(), p.data(), yp, ws, nullptr));
        return;
    }
    cusp::multiply(a, p, y);
    cusp::detail::check(dotd_(y.size(), y.data(), p.data(), yp, ws));
}
template <typename A, typename V, typename Format> void multiply_dot_f32(const A &a, const V &p, V &y, double *yp, void *ws, Format)
{
    cusp::multiply(a, p, y);
    cusp::detail::check(dotd_(y.size(), y.data(), p.data(), yp, ws));
}
template <typename A, typename V> auto multiply_dot_f32_any(const A &a, const V &p, V &y, double *yp, void *ws, int) -> decltype(typename A::format(), void())
{
    multiply_dot_f32(a, p, y, yp, ws, typename A::format());
}
template <typename A, typename V> void multiply_dot_f32_any(const A &a, const V &p, V &y, double *yp, void *ws, long)
{
    cusp::multiply(a, p, y);
    cusp::detail::check(dotd_(y.size(), y.data(), p.data(), yp, ws));
}
template <typename A, typename V> void multiply_dot_any(const A &a, const V &p, V &y, double *yp, void *ws, std::false_type /* float */)
{
    multiply_dot_f32_any(a, p, y, yp, ws, 0);
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cg_fused_device(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type T;
    const size_t N = A.num_rows;
    cusp::array1d<T, cusp::device_memory> y(N), r(N), p(N);
    cusp::array1d<double, cusp::device_memory> scalars(3); // rr[0], rr[1], <y,p>
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    double *rr[2] = {scalars.data(), scalars.data() + 1};
    double *yp = scalars.data() + 2;
    pinned_scalar rr_host;
    cusp::multiply(A, x, y);
    cusp::blas::axpby(b, y, r, T(1), T(-1));
    cusp::blas::copy(r, p);
    cusp::detail::check(dotd_(N, r.data(), r.data(), rr[0], w.ws));
    rr_host.fetch(rr[0]);
    int cur = 0;
    for (;;) {
        // the hot path (speculative, see above).  CSR through its plan: the partials of <y, p> stay in the workspace (np > 0)
        int np = multiply_dot_partials_any(A, p, y, w.ws, 0);
        if (np == 0) multiply_dot_any(A, p, y, yp, w.ws, std::is_same<T, double>());
        else if (np < 0) { cusp::detail::check(dotd_(N, y.data(), p.data(), yp, w.ws)); np = 0; }
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) break; // the one host read
        if (np > 0) { // fold-ahead: three launches per iteration, the two folds ride at the front of their consumers
            int np_rr = 0;
            cusp::detail::check(cg_update_fold_(N, rr[cur], yp, np, y.data(), r.data(), w.ws, &np_rr));
            cusp::detail::check(cg_direction_x_fold_(N, rr[cur ^ 1], rr_host.host, np_rr, rr[cur], yp, r.data(), p.data(), x.data(), w.ws));
            rr_host.record();
        } else {
            cusp::detail::check(cg_update_(N, rr[cur], yp, y.data(), r.data(), rr[cur ^ 1], rr_host.host, w.ws));
            rr_host.record();
            // x <- x + alpha p rides with the direction pass (it reads p anyway): 8 vector passes per iteration, not 9
            cusp::detail::check(cg_direction_x_(N, rr[cur ^ 1], rr[cur], yp, r.data(), p.data(), x.data()));
        }
        cur ^= 1;
        ++monitor;
    }
    cusp::detail::check(cmi_device_synchronize()); // the discarded SpMV must not outlive y
}

// M = cusp::precond::diagonal on device_memory: the same schedule with z = D^-1 r never stored (cmi_pcg_update_jacobi_* / cmi_pcg_direction_x_jacobi_*):
// SpMV + <y,p>, the update with BOTH <r,z> (for alpha / beta) and <r,r> (for the monitor, as the reference's monitor.finished(r)), the direction
// pass -- one host read per iteration behind the next, speculative multiply.
inline int pcg_update_(size_t n, const double *rz, const double *yp, const double *y, double *r, const double *d, double *rz_new, double *rr, double *mirror, void *ws)
{ return cmi_pcg_update_jacobi_f64(n, rz, yp, y, r, d, rz_new, rr, mirror, ws, nullptr); }
inline int pcg_update_(size_t n, const double *rz, const double *yp, const float *y, float *r, const float *d, double *rz_new, double *rr, double *mirror, void *ws)
{ return cmi_pcg_update_jacobi_f32(n, rz, yp, y, r, d, rz_new, rr, mirror, ws, nullptr); }
inline int pcg_direction_(size_t n, const double *rz_new, const double *rz_old, const double *yp, const double *r, const double *d, double *p, double *x)
{ return cmi_pcg_direction_x_jacobi_f64(n, rz_new, rz_old, yp, r, d, p, x, nullptr); }
inline int pcg_direction_(size_t n, const double *rz_new, const double *rz_old, const double *yp, const float *r, const float *d, float *p, float *x)
{ return cmi_pcg_direction_x_jacobi_f32(n, rz_new, rz_old, yp, r, d, p, x, nullptr); }

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Diagonal>
void cg_fused_jacobi_device(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, const Diagonal &M)
{
    typedef typename LinearOperator::value_type T;
    const size_t N = A.num_rows;
    const T *dinv = M.reciprocals().data();
    cusp::array1d<T, cusp::device_memory> y(N), r(N), p(N);
    cusp::array1d<double, cusp::device_memory> scalars(4); // <r,z>[0], <r,z>[1], <y,p>, <r,r>
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    double *rz[2] = {scalars.data(), scalars.data() + 1};
    double *yp = scalars.data() + 2, *rr = scalars.data() + 3;
    pinned_scalar rr_host;
    cusp::multiply(A, x, y);
    cusp::blas::axpby(b, y, r, T(1), T(-1));
    cusp::blas::xmy(M.reciprocals(), r, p);                           // p <- z = D^-1 r
    cusp::detail::check(dotd_(N, r.data(), p.data(), rz[0], w.ws));   // <r, z>
    cusp::detail::check(dotd_(N, r.data(), r.data(), rr, w.ws));
    rr_host.fetch(rr);
    int cur = 0;
    for (;;) {
        multiply_dot_any(A, p, y, yp, w.ws, std::is_same<T, double>()); // the hot path, speculative: y <- A p, *yp <- <y, p>
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) break; // the one host read
        cusp::detail::check(pcg_update_(N, rz[cur], yp, y.data(), r.data(), dinv, rz[cur ^ 1], rr, rr_host.host, w.ws));
        rr_host.record();
        cusp::detail::check(pcg_direction_(N, rz[cur ^ 1], rz[cur], yp, r.data(), dinv, p.data(), x.data()));
        cur ^= 1;
        ++monitor;
    }
    cusp::detail::check(cmi_device_synchronize()); // the discarded SpMV must not outlive y
}

template <typename A, typename X, typename M, typename Mon> struct use_fused_jacobi {
    typedef typename A::value_type T;
    static const bool value = std::is_same<typename A::memory_space, cusp::device_memory>::value &&
                              (std::is_same<T, double>::value || std::is_same<T, float>::value) && std::is_same<typename X::value_type, T>::value &&
                              std::is_same<typename std::remove_const<M>::type, cusp::precond::diagonal<T, cusp::device_memory>>::value &&
                              decltype(has_finished_norm(static_cast<Mon *>(nullptr)))::value;
};

template <typename A, typename X, typename M, typename Mon> struct use_fused {
    typedef typename A::value_type T;
    static const bool value = std::is_same<typename A::memory_space, cusp::device_memory>::value &&
                              (std::is_same<T, double>::value || std::is_same<T, float>::value) && std::is_same<typename X::value_type, T>::value &&
                              std::is_same<M, cusp::identity_operator<T, cusp::device_memory>>::value &&
                              decltype(has_finished_norm(static_cast<Mon *>(nullptr)))::value;
};

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg_plain(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M);

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg_select(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &, std::true_type)
{
    cg_fused_device(A, x, b, monitor);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg_select_plain(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M, std::true_type) // Jacobi, device
{
    const char *e = std::getenv("CMI_CG_FUSED_JACOBI");
    if (e && e[0] == '0') { cg_plain(A, x, b, monitor, M); return; } // (measurements: the operation-by-operation path)
    cg_fused_jacobi_device(A, x, b, monitor, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg_select_plain(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M, std::false_type)
{
    cg_plain(A, x, b, monitor, M);
}
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg_select(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M, std::false_type)
{
    cg_select_plain(A, x, b, monitor, M, std::integral_constant<bool, use_fused_jacobi<LinearOperator, VectorType1, Preconditioner, Monitor>::value>());
}

} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = detail::not_policy<LinearOperator>>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("cg: matrix must be square");
    detail::cg_select(A, x, b, monitor, M,
                      std::integral_constant<bool, detail::use_fused<LinearOperator, VectorType1, Preconditioner, Monitor>::value>());
}

// reference cusp/krylov/detail/cg.inl:41-107, operation by operation (any memory space, any M)
template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void detail::cg_plain(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
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

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename = detail::not_policy<LinearOperator>>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    cusp::identity_operator<ValueType, MemorySpace> M(A.num_rows, A.num_cols);
    cusp::krylov::cg(A, x, b, monitor, M);
}

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename = detail::not_policy<LinearOperator>>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    typedef typename LinearOperator::value_type ValueType;
    cusp::monitor<ValueType> monitor(b);
    cusp::krylov::cg(A, x, b, monitor);
}

} // namespace krylov
} // namespace cusp

// ---- execution-policy overloads (reference cusp/krylov/cg.h: cg(exec, A, x, b[, monitor[, M]])) ---------
// A policy derived from cusp::execution_policy<Derived> reaches a user `cg(my_policy&, ...)` overload by ADL
// (testing/cg.cu:11-44); a policy without one gets the memory-space dispatch above.
namespace cusp {
namespace krylov {
namespace detail {
namespace policy_default {
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner,
          typename = typename std::enable_if<std::is_base_of<cusp::execution_policy<Derived>, Derived>::value>::type>
void cg(Derived &, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    cusp::krylov::cg(A, x, b, monitor, M);
}
} // namespace policy_default
} // namespace detail

template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
void cg(const cusp::execution_policy<Derived> &exec, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M)
{
    using cusp::krylov::detail::policy_default::cg;
    cg(const_cast<Derived &>(exec.derived()), A, x, b, monitor, M);
}
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cg(const cusp::execution_policy<Derived> &exec, const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    cusp::identity_operator<ValueType, MemorySpace> M(A.num_rows, A.num_cols);
    cusp::krylov::cg(exec, A, x, b, monitor, M);
}
template <typename Derived, typename LinearOperator, typename VectorType1, typename VectorType2>
void cg(const cusp::execution_policy<Derived> &exec, const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    typedef typename LinearOperator::value_type ValueType;
    cusp::monitor<ValueType> monitor(b);
    cusp::krylov::cg(exec, A, x, b, monitor);
}
} // namespace krylov
} // namespace cusp
