// cusp/krylov/cg.h -- (preconditioned) conjugate gradients, the caller of the SpMV hot path
// (reference cusp/krylov/cg.h, cusp/krylov/detail/cg.inl:41-107: same operations in the same order,
// so the residual history matches the reference's, e.g. docs/quickstart.md:72-87).
// One cusp::multiply per iteration -> cmi_spmv_* on device_memory; vector updates -> cusp::blas.
#pragma once
#include <cassert>
#include <cmath>
#include <type_traits>

#include "../array1d.h"
#include "../blas/blas.h"
#include "../linear_operator.h"
#include "../monitor.h"
#include "../multiply.h"

namespace cusp {
namespace krylov {

namespace detail {
// z <- M r for a matrix-like preconditioner or a linear operator with operator()
template <typename M, typename X, typename Y> auto apply(const M &m, const X &x, Y &y, int) -> decltype(m(x, y), void()) { m(x, y); }
template <typename M, typename X, typename Y> void apply(const M &m, const X &x, Y &y, long) { cusp::multiply(m, x, y); }
} // namespace detail

namespace detail {

// Fused unpreconditioned CG on the device (f64): z == r is folded away, alpha and beta stay in device
// memory, and the vector work of an iteration is cmi_blas_dot + cmi_cg_update + cmi_cg_direction:
// 4 vector passes and ONE host read (the convergence check) instead of cg.inl's 7 passes and 3 host
// syncs.  Per-element arithmetic unchanged; the residual history agrees with the plain path to rounding.
template <typename Monitor> auto has_finished_norm(Monitor *m) -> decltype(m->finished_norm(typename Monitor::Real()), std::true_type());
std::false_type has_finished_norm(...);

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cg_fused_device(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    const size_t N = A.num_rows;
    cusp::array1d<double, cusp::device_memory> y(N), r(N), p(N), scalars(3); // scalars: rr[0], rr[1], <y,p>
    cusp::blas::detail::device_workspace &w = cusp::blas::detail::workspace();
    double *rr[2] = {scalars.data(), scalars.data() + 1};
    double *yp = scalars.data() + 2;
    cusp::multiply(A, x, y);
    cusp::blas::axpby(b, y, r, 1.0, -1.0);
    cusp::blas::copy(r, p);
    cusp::detail::check(cmi_blas_dot_f64(N, r.data(), r.data(), rr[0], w.ws, nullptr));
    int cur = 0;
    for (;;) {
        double rr_host;
        cusp::detail::check(cmi_memcpy_d2h(&rr_host, rr[cur], sizeof(double), nullptr)); // the one host read
        if (monitor.finished_norm(std::sqrt(rr_host))) break;
        cusp::multiply(A, p, y);                                                          // the hot path
        cusp::detail::check(cmi_blas_dot_f64(N, y.data(), p.data(), yp, w.ws, nullptr));
        cusp::detail::check(cmi_cg_update_f64(N, rr[cur], yp, p.data(), y.data(), x.data(), r.data(), rr[cur ^ 1], w.ws, nullptr));
        cusp::detail::check(cmi_cg_direction_f64(N, rr[cur ^ 1], rr[cur], r.data(), p.data(), nullptr));
        cur ^= 1;
        ++monitor;
    }
}

template <typename A, typename X, typename M, typename Mon> struct use_fused {
    static const bool value = std::is_same<typename A::memory_space, cusp::device_memory>::value &&
                              std::is_same<typename A::value_type, double>::value && std::is_same<typename X::value_type, double>::value &&
                              std::is_same<M, cusp::identity_operator<double, cusp::device_memory>>::value &&
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
void cg_select(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor, Preconditioner &M, std::false_type)
{
    cg_plain(A, x, b, monitor, M);
}

} // namespace detail

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor, typename Preconditioner>
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

template <typename LinearOperator, typename VectorType1, typename VectorType2, typename Monitor>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b, Monitor &monitor)
{
    typedef typename LinearOperator::value_type ValueType;
    typedef typename LinearOperator::memory_space MemorySpace;
    cusp::identity_operator<ValueType, MemorySpace> M(A.num_rows, A.num_cols);
    cusp::krylov::cg(A, x, b, monitor, M);
}

template <typename LinearOperator, typename VectorType1, typename VectorType2>
void cg(const LinearOperator &A, VectorType1 &x, const VectorType2 &b)
{
    typedef typename LinearOperator::value_type ValueType;
    cusp::monitor<ValueType> monitor(b);
    cusp::krylov::cg(A, x, b, monitor);
}

} // namespace krylov
} // namespace cusp
