// cusp/distributed/cg.h -- cusp::krylov::cg on a row-block sharded operator (BASELINE.json configs[4]: "row-block sharded across
// 8 x MI355X, RCCL allgather(x) over xGMI, inside cusp::krylov::cg").  Same operations in the same order as the reference
// (cusp/krylov/detail/cg.inl:41-107); what changes is where the pieces live: every rank holds its slice of x, b and of the work
// vectors, the search direction p lives INSIDE the operator's exchange buffer (y <- A p is exchange + local SpMV, no staging copy),
// and the two inner products and the residual norm of an iteration are 8-byte all-reduces.
//   plain  (any local memory space): cg.inl operation by operation on sharded vectors -- cusp::blas does the all-reducing.
//   fused  (device_memory, float / double, a monitor with finished_norm): the single-GPU fused iteration of cusp/krylov/cg.h with
//          the all-reduces between its launches --  exchange p | SpMV + local <y,p> | all-reduce | r, local <r,r> | all-reduce |
//          x, p  -- scalars never leave device memory, ONE host read per iteration (the convergence check, behind the next
//          iteration's exchange + SpMV, which are queued before the host waits).
//   fused, one-sided exchange (exchange_mode::peer): p is exchanged ONCE.  Afterwards every rank keeps the p entries of its halo
//          itself -- p_halo <- r_halo + beta p_halo, the owner's arithmetic with the same all-reduced beta, hence the owner's bits --
//          and r_halo is PULLED from the owner right behind the all-reduce of <r,r>, which completes only when every rank's update
//          kernel (the one that wrote that r) has; the owner overwrites r again only behind the next all-reduce of <y,p>, which
//          needs this rank's contribution, queued behind this pull.  The algorithm's own all-reduces are all the ordering the pulls
//          need: no collective is added and nothing but 2 x halo values moves per iteration.
#pragma once
#include <cmath>

#include "../krylov/cg.h"
#include "../monitor.h"
#include "multiply.h"

namespace cusp {
namespace distributed {

// reference cusp/krylov/detail/cg.inl:41-107, identity preconditioner, on sharded vectors
template <typename I, typename V, typename L, typename X, typename B, typename Monitor>
void cg_plain(const csr_matrix<I, V, L> &A, X &x, const B &b, Monitor &monitor)
{
    typedef vector<V, L> vec;
    vec y = A.make_vector(), z = A.make_vector(), r = A.make_vector();
    vec p = A.exchange_slice();                              // p lives in the exchange buffer
    cusp::multiply(A, x, y);                                 // y <- A x   (x's slice is staged through that buffer once)
    cusp::blas::axpby(b, y, r, V(1), V(-1));                 // r <- b - A x
    cusp::blas::copy(r, z);                                  // z <- M r, M = I
    cusp::blas::copy(z, p);                                  // p <- z
    V rz = cusp::blas::dotc(r, z);                           // (all-reduced)
    while (!monitor.finished(r)) {                           // ||r|| all-reduced: every rank takes the same decision
        cusp::multiply(A, p, y);                             // y <- A p: exchange + the single-GPU hot path
        const V alpha = rz / cusp::blas::dotc(y, p);
        cusp::blas::axpy(p, x, alpha);
        cusp::blas::axpy(y, r, -alpha);
        cusp::blas::copy(r, z);
        const V rz_old = rz;
        rz = cusp::blas::dotc(r, z);
        const V beta = rz / rz_old;
        cusp::blas::axpby(z, p, p, V(1), beta);
        ++monitor;
    }
}

namespace detail {
inline int local_dot_plan(const cmi_plan *pl, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, const double *w, double *yp, void *ws)
{ return cmi_spmv_csr_dot_plan_f64(pl, Ap, Aj, Ax, x, y, w, yp, ws, nullptr); }
inline int local_dot_plan(const cmi_plan *pl, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, const float *w, double *yp, void *ws)
{ return cmi_spmv_csr_dot_plan_f32(pl, Ap, Aj, Ax, x, y, w, yp, ws, nullptr); }
inline int local_dot(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, const double *w, double *yp, void *ws)
{ return cmi_spmv_csr_dot_f64(r, c, n, Ap, Aj, Ax, x, y, w, yp, ws, nullptr, nullptr); }
inline int local_dot(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, const float *w, double *yp, void *ws)
{ return cmi_spmv_csr_dot_f32(r, c, n, Ap, Aj, Ax, x, y, w, yp, ws, nullptr, nullptr); }
} // namespace detail

// y_local <- A_local * (exchange buffer) and *yp <- <y_local, w_local> in one launch (w: this rank's slice of p)
template <typename I, typename V>
void multiply_dot_local(const csr_matrix<I, V, cusp::device_memory> &A, V *y_local, const V *w_local, double *yp, void *ws)
{
    const auto &a = A.local;
    if (a.num_rows == 0) { cusp::detail::check(cmi_memset(yp, 0, sizeof(double), nullptr)); return; }
    if (const cmi_plan *pl = a.num_entries ? a.plan() : nullptr)
        cusp::detail::check(detail::local_dot_plan(pl, a.row_offsets.data(), a.column_indices.data(), a.values.data(), A.x_full(), y_local, w_local, yp, ws));
    else
        cusp::detail::check(detail::local_dot((int64_t)a.num_rows, (int64_t)a.num_cols, (int64_t)a.num_entries, a.row_offsets.data(), a.column_indices.data(), a.values.data(),
                                              A.x_full(), y_local, w_local, yp, ws));
}

namespace detail {
inline int cg_direction_(size_t n, const double *rn, const double *ro, const double *r, double *p) { return cmi_cg_direction_f64((int64_t)n, rn, ro, r, p, nullptr); }
inline int cg_direction_(size_t n, const double *rn, const double *ro, const float *r, float *p) { return cmi_cg_direction_f32((int64_t)n, rn, ro, r, p, nullptr); }
} // namespace detail

template <typename I, typename V, typename X, typename B, typename Monitor>
void cg_fused_peer(const csr_matrix<I, V, cusp::device_memory> &A, X &x, const B &b, Monitor &monitor)
{
    namespace kd = cusp::krylov::detail;
    typedef vector<V, cusp::device_memory> vec;
    communicator &comm = A.comm();
    const size_t N = A.local_rows();
    exchange_buffer<V, cusp::device_memory> r_ex;
    A.make_exchange_buffer(r_ex);                                        // (collective) peers map it: they pull r's boundary values
    vec y = A.make_vector();
    vec r(comm, r_ex.data.data() + A.row_begin(), N, A.num_rows, A.row_begin());
    vec p = A.exchange_slice();
    V *p_full = const_cast<V *>(A.x_full()), *r_full = r_ex.data.data();
    const auto halo = A.halo_ranges();
    cusp::array1d<double, cusp::device_memory> scalars(3);
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    double *rr[2] = {scalars.data(), scalars.data() + 1};
    double *yp = scalars.data() + 2;
    kd::pinned_scalar rr_host;
    cusp::multiply(A, x, y);                                             // (fenced exchange: set-up, once)
    cusp::blas::axpby(b, y, r, V(1), V(-1));
    A.fence();                                                           // the peers have pulled x out of the buffer p reuses
    cusp::blas::copy(r, p);
    cusp::detail::check(kd::dotd_(N, r.data(), r.data(), rr[0], w.ws));
    comm.allreduce_sum(rr[0], 1, cusp::device_memory());
    A.pull();                                                            // p's halo, ordered behind every rank's copy by the all-reduce
    rr_host.fetch(rr[0]);
    int cur = 0;
    for (;;) {
        multiply_dot_local(A, y.data(), p.data(), yp, w.ws);             // halos of p are already in place: no exchange
        comm.allreduce_sum(yp, 1, cusp::device_memory());
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) break;
        cusp::detail::check(kd::cg_update_(N, rr[cur], yp, y.data(), r.data(), rr[cur ^ 1], nullptr, w.ws));
        comm.allreduce_sum(rr[cur ^ 1], 1, cusp::device_memory());
        rr_host.fetch(rr[cur ^ 1]);
        r_ex.pull();                                                     // the peers' boundary r
        for (const auto &h : halo)                                       // halo: p <- r + beta p (the owner's arithmetic)
            cusp::detail::check(detail::cg_direction_((size_t)h.second, rr[cur ^ 1], rr[cur], r_full + h.first, p_full + h.first));
        cusp::detail::check(kd::cg_direction_x_(N, rr[cur ^ 1], rr[cur], yp, r.data(), p.data(), x.data()));
        cur ^= 1;
        ++monitor;
    }
    A.fence();                                                           // nobody is still pulling from r_ex; the discarded SpMV has ended
    r_ex.close();
}

template <typename I, typename V, typename X, typename B, typename Monitor>
void cg_fused(const csr_matrix<I, V, cusp::device_memory> &A, X &x, const B &b, Monitor &monitor)
{
    if (A.mode() == exchange_mode::peer) { cg_fused_peer(A, x, b, monitor); return; }
    namespace kd = cusp::krylov::detail;
    typedef vector<V, cusp::device_memory> vec;
    communicator &comm = A.comm();
    const size_t N = A.local_rows();
    vec y = A.make_vector(), r = A.make_vector();
    vec p = A.exchange_slice();
    cusp::array1d<double, cusp::device_memory> scalars(3); // rr[0], rr[1], <y,p>
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    double *rr[2] = {scalars.data(), scalars.data() + 1};
    double *yp = scalars.data() + 2;
    kd::pinned_scalar rr_host;
    cusp::multiply(A, x, y);
    cusp::blas::axpby(b, y, r, V(1), V(-1));
    cusp::blas::copy(r, p);
    cusp::detail::check(kd::dotd_(N, r.data(), r.data(), rr[0], w.ws));
    comm.allreduce_sum(rr[0], 1, cusp::device_memory());
    rr_host.fetch(rr[0]);
    int cur = 0;
    for (;;) {
        A.exchange();                                                   // p's slices -> every rank's buffer      (speculative,
        multiply_dot_local(A, y.data(), p.data(), yp, w.ws);            // y <- A p, local <y, p>                  like the SpMV
        comm.allreduce_sum(yp, 1, cusp::device_memory());               //                                          of cg.h)
        if (monitor.finished_norm(static_cast<typename Monitor::Real>(std::sqrt(rr_host.wait())))) break; // the one host read
        cusp::detail::check(kd::cg_update_(N, rr[cur], yp, y.data(), r.data(), rr[cur ^ 1], nullptr, w.ws)); // r, local <r,r>
        comm.allreduce_sum(rr[cur ^ 1], 1, cusp::device_memory());
        rr_host.fetch(rr[cur ^ 1]);
        cusp::detail::check(kd::cg_direction_x_(N, rr[cur ^ 1], rr[cur], yp, r.data(), p.data(), x.data())); // x, p
        cur ^= 1;
        ++monitor;
    }
    cusp::detail::check(cmi_device_synchronize()); // the discarded exchange + SpMV must not outlive y
}

} // namespace distributed

namespace krylov {
namespace detail {
template <typename I, typename V, typename X, typename B, typename Monitor>
void cg_sharded(const distributed::csr_matrix<I, V, cusp::device_memory> &A, X &x, const B &b, Monitor &monitor, std::true_type) { distributed::cg_fused(A, x, b, monitor); }
template <typename I, typename V, typename L, typename X, typename B, typename Monitor>
void cg_sharded(const distributed::csr_matrix<I, V, L> &A, X &x, const B &b, Monitor &monitor, std::false_type) { distributed::cg_plain(A, x, b, monitor); }
} // namespace detail

// cusp::krylov::cg(A, x, b, monitor) / (A, x, b) with A sharded: x, b = this rank's slices (cusp::distributed::vector)
template <typename I, typename V, typename L, typename X, typename B, typename Monitor>
void cg(const distributed::csr_matrix<I, V, L> &A, X &x, const B &b, Monitor &monitor)
{
    static_assert(std::is_same<typename X::memory_space, cusp::distributed_memory<L>>::value, "cg: x must be a cusp::distributed::vector in the operator's local space");
    if (x.size() != A.local_rows() || b.size() != A.local_rows()) throw cusp::invalid_input_exception("cg: x and b must be this rank's slices of the operator's row partition");
    constexpr bool fused = std::is_same<L, cusp::device_memory>::value && (std::is_same<V, double>::value || std::is_same<V, float>::value) &&
                           std::is_same<typename X::value_type, V>::value && decltype(detail::has_finished_norm(static_cast<Monitor *>(nullptr)))::value;
    detail::cg_sharded(A, x, b, monitor, std::integral_constant<bool, fused>());
}
template <typename I, typename V, typename L, typename X, typename B>
void cg(const distributed::csr_matrix<I, V, L> &A, X &x, const B &b)
{
    cusp::monitor<V> monitor(b);
    cusp::krylov::cg(A, x, b, monitor);
}
} // namespace krylov
} // namespace cusp
