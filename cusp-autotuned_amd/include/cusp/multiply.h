// cusp/multiply.h -- cusp::multiply(A, x, y): the drop-in boundary of the SpMV hot path.
//
// Reference: cusp/multiply.h:40,101,113,190 (the four overloads), cusp/detail/multiply.inl:27-105
// (memory-space dispatch), cusp/system/detail/generic/multiply.inl:98-111,173-191 (format dispatch).
//
//   host_memory   : the reference's sequential loops, restated (sequential/multiply/{csr,coo,ell,dia,
//                   hyb}_spmv.h) -- same summation order, fully generic in (initialize, combine, reduce).
//   device_memory : forwards to the MI355X C-ABI (cmi_spmv_{csr,coo,ell,dia,hyb}_{f64,f32}): hand-written
//                   gfx950 kernels, launch shape from the persisted tuning table.  NEVER falls back to the
//                   host: a functor combination the kernels do not implement throws
//                   cusp::not_implemented_exception.  Supported on the device: combine = multiplies,
//                   reduce = plus, initialize = constant_functor(0) (y = A x) or identity_function
//                   (y = y + A x) -- exactly the combinations the reference's own multiply / hyb path use.
//
// x and y may be any contiguous vector type of the matrix's memory space with data()/size():
// cusp::array1d, cusp::array1d_view, or a solver's temporary (the reference's CG passes
// temporary_array, which the fork's KTT hook fails to match -- SURVEY.md 3.3; here every vector type
// reaches the tuned kernels).
//
// Execution policies: multiply(exec, A, x, y) with cusp::hip::par.on(stream) selects the stream.  A
// user-derived policy reaches a user-provided multiply overload by ADL without being copied
// (testing/multiply.cu:792-858): see cusp/execution_policy.h.
#pragma once
#include <utility>
#include "array1d.h"
#include "array2d.h"
#include "convert.h"
#include "execution_policy.h"
#include "functional.h"

namespace cusp {

namespace detail {

// ---- host loops (reference sequential backend) ---------------------------------------------------
template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, csr_format)
{
    typedef typename A::index_type I;
    typedef typename Y::value_type V;
    for (size_t i = 0; i < a.num_rows; i++) { // sequential/multiply/csr_spmv.h:56-72
        V acc = initialize(y[i]);
        for (I jj = a.row_offsets[i]; jj < a.row_offsets[i + 1]; jj++) acc = reduce(acc, combine(a.values[jj], x[a.column_indices[jj]]));
        y[i] = acc;
    }
}

template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, coo_format)
{
    for (size_t i = 0; i < a.num_rows; i++) y[i] = initialize(y[i]); // coo_spmv.h:56-67
    for (size_t n = 0; n < a.num_entries; n++) {
        const auto i = a.row_indices[n];
        y[i] = reduce(y[i], combine(a.values[n], x[a.column_indices[n]]));
    }
}

template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, ell_format)
{
    typedef typename A::index_type I;
    const size_t width = a.column_indices.num_cols, pitch = a.column_indices.pitch;
    const I *cj = a.column_indices.data_ptr();
    const auto *va = a.values.data_ptr();
    for (size_t i = 0; i < a.num_rows; i++) y[i] = initialize(y[i]); // ell_spmv.h:59-74
    for (size_t n = 0; n < width; n++)
        for (size_t i = 0; i < a.num_rows; i++) {
            const I j = cj[n * pitch + i];
            if (j != I(-1)) y[i] = reduce(y[i], combine(va[n * pitch + i], x[j]));
        }
}

template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, dia_format)
{
    typedef typename A::index_type I;
    const size_t nd = a.values.num_cols, pitch = a.values.pitch;
    const auto *va = a.values.data_ptr();
    for (size_t i = 0; i < a.num_rows; i++) y[i] = initialize(y[i]); // dia_spmv.h:61-80
    for (size_t d = 0; d < nd; d++) {
        const I k = a.diagonal_offsets[d];
        const size_t i_start = std::max<I>(0, -k), j_start = std::max<I>(0, k);
        if (i_start >= a.num_rows || j_start >= a.num_cols) continue;
        const size_t N = std::min(a.num_rows - i_start, a.num_cols - j_start);
        for (size_t n = 0; n < N; n++) y[i_start + n] = reduce(y[i_start + n], combine(va[d * pitch + i_start + n], x[j_start + n]));
    }
}

template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, hyb_format)
{
    typedef typename Y::value_type V;
    host_multiply(a.ell, x, y, initialize, combine, reduce, ell_format()); // hyb_spmv.h:55-56
    host_multiply(a.coo, x, y, identity_function<V>(), combine, reduce, coo_format());
}

// dense matrix-vector product on the host (the reference tests' ground truth, array2d_mv.h)
template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void host_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, array2d_format)
{
    typedef typename Y::value_type V;
    for (size_t i = 0; i < a.num_rows; i++) {
        V acc = initialize(y[i]);
        for (size_t j = 0; j < a.num_cols; j++) acc = reduce(acc, combine(a(i, j), x[j]));
        y[i] = acc;
    }
}

// uniform access to the raw 2-D storage of containers and views
template <typename T, typename M, typename O> const T *data_of(const array2d<T, M, O> &a) { return a.values.data(); }
template <typename V2> auto data_of(const V2 &v) -> decltype(v.data()) { return v.data(); }

// ---- device dispatch: which accumulate flag does (initialize, combine, reduce) mean? -------------
template <typename V, typename Init, typename Comb, typename Red> struct device_functors {
    static int accumulate(const Init &)
    {
        throw cusp::not_implemented_exception("cusp::multiply on device_memory implements combine = multiplies, reduce = plus, "
                                              "initialize = constant_functor(0) or identity_function; other functors are host-only");
    }
};
template <typename V> struct device_functors<V, constant_functor<V>, multiplies<V>, plus<V>> {
    static int accumulate(const constant_functor<V> &f)
    {
        if (!(f.value == V(0))) throw cusp::not_implemented_exception("cusp::multiply on device_memory: initialize must be constant 0 or identity");
        return 0;
    }
};
template <typename V> struct device_functors<V, identity_function<V>, multiplies<V>, plus<V>> {
    static int accumulate(const identity_function<V> &) { return 1; }
};

// explicit launch configuration for cusp::ktt::tune (thread-local; nullptr = tuning table)
inline const cmi_config *&forced_config() { static thread_local const cmi_config *c = nullptr; return c; }

inline int spmv_csr(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_csr_f64(r, c, n, Ap, Aj, Ax, x, y, acc, forced_config(), s); }
inline int spmv_csr(int64_t r, int64_t c, int64_t n, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_csr_f32(r, c, n, Ap, Aj, Ax, x, y, acc, forced_config(), s); }
inline int spmv_coo(int64_t r, int64_t c, int64_t n, const int *Ai, const int *Aj, const double *Ax, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_coo_f64(r, c, n, Ai, Aj, Ax, x, y, acc, forced_config(), s); }
inline int spmv_coo(int64_t r, int64_t c, int64_t n, const int *Ai, const int *Aj, const float *Ax, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_coo_f32(r, c, n, Ai, Aj, Ax, x, y, acc, forced_config(), s); }
inline int spmv_ell(int64_t r, int64_t c, int64_t w, int64_t p, const int *Aj, const double *Ax, const int *rl, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_ell_f64(r, c, w, p, Aj, Ax, rl, x, y, acc, forced_config(), s); }
inline int spmv_ell(int64_t r, int64_t c, int64_t w, int64_t p, const int *Aj, const float *Ax, const int *rl, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_ell_f32(r, c, w, p, Aj, Ax, rl, x, y, acc, forced_config(), s); }
inline int spmv_dia(int64_t r, int64_t c, int64_t d, int64_t p, const int *off, const double *v, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_dia_f64(r, c, d, p, off, v, x, y, acc, forced_config(), s); }
inline int spmv_dia(int64_t r, int64_t c, int64_t d, int64_t p, const int *off, const float *v, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_dia_f32(r, c, d, p, off, v, x, y, acc, forced_config(), s); }
inline int spmv_hyb(int64_t r, int64_t c, int64_t w, int64_t p, const int *eAj, const double *eAx, int64_t n, const int *cAi, const int *cAj,
                    const double *cAx, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_hyb_f64(r, c, w, p, eAj, eAx, n, cAi, cAj, cAx, x, y, acc, nullptr, nullptr, s); }
inline int spmv_hyb(int64_t r, int64_t c, int64_t w, int64_t p, const int *eAj, const float *eAx, int64_t n, const int *cAi, const int *cAj,
                    const float *cAx, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_hyb_f32(r, c, w, p, eAj, eAx, n, cAi, cAj, cAx, x, y, acc, nullptr, nullptr, s); }

inline int spmv_csr_plan(const cmi_plan *p, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_csr_plan_f64(p, Ap, Aj, Ax, x, y, acc, s); }
inline int spmv_csr_plan(const cmi_plan *p, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_csr_plan_f32(p, Ap, Aj, Ax, x, y, acc, s); }
inline int spmv_coo_plan(const cmi_plan *p, const int *Ai, const int *Aj, const double *Ax, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_coo_plan_f64(p, Ai, Aj, Ax, x, y, acc, s); }
inline int spmv_coo_plan(const cmi_plan *p, const int *Ai, const int *Aj, const float *Ax, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_coo_plan_f32(p, Ai, Aj, Ax, x, y, acc, s); }

inline int spmv_hyb_plan(const cmi_plan *p, size_t pitch, const int *eAj, const double *eAx, const int *Ai, const int *Aj, const double *Ax, const double *x, double *y, int acc, void *s)
{ return cmi_spmv_hyb_plan_f64(p, (int64_t)pitch, eAj, eAx, Ai, Aj, Ax, x, y, acc, s); }
inline int spmv_hyb_plan(const cmi_plan *p, size_t pitch, const int *eAj, const float *eAx, const int *Ai, const int *Aj, const float *Ax, const float *x, float *y, int acc, void *s)
{ return cmi_spmv_hyb_plan_f32(p, (int64_t)pitch, eAj, eAx, Ai, Aj, Ax, x, y, acc, s); }

// the container's plan (views, and anything while cusp::ktt::tune forces a configuration: none)
template <typename A> auto plan_of(const A &a, void *stream, int) -> decltype(a.plan(stream))
{
    return (forced_config() || a.num_entries == 0) ? nullptr : a.plan(stream);
}
template <typename A> const cmi_plan *plan_of(const A &, void *, long) { return nullptr; }

template <typename A> void require_int_index()
{
    static_assert(std::is_same<typename A::index_type, int>::value, "device_memory matrices use 32-bit int indices (the reference's default IndexType)");
}

// optional ELLR row lengths
template <typename A> auto row_lengths_of(const A &a, int) -> decltype(a.row_lengths.data()) { return a.row_lengths.size() ? a.row_lengths.data() : nullptr; }
template <typename A> const int *row_lengths_of(const A &, long) { return nullptr; }

template <typename A, typename X, typename Y> void device_multiply(const A &a, const X &x, Y &y, int acc, void *stream, csr_format)
{
    require_int_index<A>();
    if (const cmi_plan *p = plan_of(a, stream, 0)) {
        check(spmv_csr_plan(p, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(), acc, stream));
        return;
    }
    check(spmv_csr(a.num_rows, a.num_cols, a.num_entries, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(), acc, stream));
}
template <typename A, typename X, typename Y> void device_multiply(const A &a, const X &x, Y &y, int acc, void *stream, coo_format)
{
    require_int_index<A>();
    if (const cmi_plan *p = plan_of(a, stream, 0)) {
        check(spmv_coo_plan(p, a.row_indices.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(), acc, stream));
        return;
    }
    check(spmv_coo(a.num_rows, a.num_cols, a.num_entries, a.row_indices.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(), acc, stream));
}
template <typename A, typename X, typename Y> void device_multiply(const A &a, const X &x, Y &y, int acc, void *stream, ell_format)
{
    require_int_index<A>();
    // reference ell_spmv.h:138 asserts equal pitches; here it is an error the caller sees
    if (a.column_indices.pitch != a.values.pitch) throw cusp::invalid_input_exception("ell_matrix: column_indices.pitch != values.pitch");
    check(spmv_ell(a.num_rows, a.num_cols, a.column_indices.num_cols, a.column_indices.pitch, data_of(a.column_indices), data_of(a.values),
                   row_lengths_of(a, 0), x.data(), y.data(), acc, stream));
}
template <typename A, typename X, typename Y> void device_multiply(const A &a, const X &x, Y &y, int acc, void *stream, dia_format)
{
    require_int_index<A>();
    check(spmv_dia(a.num_rows, a.num_cols, a.values.num_cols, a.values.pitch, a.diagonal_offsets.data(), data_of(a.values), x.data(), y.data(), acc, stream));
}
template <typename A, typename X, typename Y> void device_multiply(const A &a, const X &x, Y &y, int acc, void *stream, hyb_format)
{
    require_int_index<A>();
    if (a.ell.column_indices.pitch != a.ell.values.pitch) throw cusp::invalid_input_exception("hyb_matrix: ell.column_indices.pitch != ell.values.pitch");
    if (const cmi_plan *p = plan_of(a, stream, 0)) { // COO part sorted by row (every conversion's output): one launch, y written once
        check(spmv_hyb_plan(p, a.ell.column_indices.pitch, data_of(a.ell.column_indices), data_of(a.ell.values), a.coo.row_indices.data(),
                            a.coo.column_indices.data(), a.coo.values.data(), x.data(), y.data(), acc, stream));
        return;
    }
    check(spmv_hyb(a.num_rows, a.num_cols, a.ell.column_indices.num_cols, a.ell.column_indices.pitch, data_of(a.ell.column_indices), data_of(a.ell.values),
                   a.coo.num_entries, a.coo.row_indices.data(), a.coo.column_indices.data(), a.coo.values.data(), x.data(), y.data(), acc, stream));
}

// host adaptor: containers expose operator[] / data; give ELL/DIA a data_ptr() view for the loops
template <typename M> struct host_ell_adaptor {
    const M &m;
    size_t num_rows, num_cols, num_entries;
    struct arr_i { const typename M::index_type *p; size_t num_cols, pitch; const typename M::index_type *data_ptr() const { return p; } } column_indices;
    struct arr_v { const typename M::value_type *p; size_t num_cols, pitch; const typename M::value_type *data_ptr() const { return p; } } values;
    typedef typename M::index_type index_type;
    explicit host_ell_adaptor(const M &mm)
        : m(mm), num_rows(mm.num_rows), num_cols(mm.num_cols), num_entries(mm.num_entries),
          column_indices{data_of(mm.column_indices), mm.column_indices.num_cols, mm.column_indices.pitch},
          values{data_of(mm.values), mm.values.num_cols, mm.values.pitch} {}
};
template <typename M> struct host_dia_adaptor {
    size_t num_rows, num_cols, num_entries;
    decltype(std::declval<const M &>().diagonal_offsets) const &diagonal_offsets;
    struct arr_v { const typename M::value_type *p; size_t num_cols, pitch; const typename M::value_type *data_ptr() const { return p; } } values;
    typedef typename M::index_type index_type;
    explicit host_dia_adaptor(const M &mm)
        : num_rows(mm.num_rows), num_cols(mm.num_cols), num_entries(mm.num_entries), diagonal_offsets(mm.diagonal_offsets),
          values{data_of(mm.values), mm.values.num_cols, mm.values.pitch} {}
};
template <typename M> struct host_hyb_adaptor {
    size_t num_rows, num_cols, num_entries;
    host_ell_adaptor<decltype(std::declval<const M &>().ell)> ell;
    decltype(std::declval<const M &>().coo) const &coo;
    explicit host_hyb_adaptor(const M &mm) : num_rows(mm.num_rows), num_cols(mm.num_cols), num_entries(mm.num_entries), ell(mm.ell), coo(mm.coo) {}
};

template <typename A, typename X, typename Y, typename I, typename C, typename R, typename F>
void host_dispatch(const A &a, const X &x, Y &y, I i, C c, R r, F f) { host_multiply(a, x, y, i, c, r, f); }
template <typename A, typename X, typename Y, typename I, typename C, typename R>
void host_dispatch(const A &a, const X &x, Y &y, I i, C c, R r, ell_format) { host_multiply(host_ell_adaptor<A>(a), x, y, i, c, r, ell_format()); }
template <typename A, typename X, typename Y, typename I, typename C, typename R>
void host_dispatch(const A &a, const X &x, Y &y, I i, C c, R r, dia_format) { host_multiply(host_dia_adaptor<A>(a), x, y, i, c, r, dia_format()); }
template <typename A, typename X, typename Y, typename I, typename C, typename R>
void host_dispatch(const A &a, const X &x, Y &y, I i, C c, R r, hyb_format) { host_multiply(host_hyb_adaptor<A>(a), x, y, i, c, r, hyb_format()); }

template <typename A, typename X, typename Y> void check_shapes(const A &a, const X &x, const Y &y)
{
    // reference: cusp::invalid_input_exception on mismatched dimensions
    if (x.size() != a.num_cols || y.size() != a.num_rows) throw cusp::invalid_input_exception("cusp::multiply: vector sizes do not match the matrix shape");
}

template <typename A, typename X, typename Y, typename I, typename C, typename R>
void multiply_in_space(const A &a, const X &x, Y &y, I init, C comb, R red, void * /*stream*/, host_memory)
{
    host_dispatch(a, x, y, init, comb, red, typename A::format());
}
template <typename A, typename X, typename Y, typename I, typename C, typename R>
void multiply_in_space(const A &a, const X &x, Y &y, I init, C comb, R red, void *stream, device_memory)
{
    typedef typename Y::value_type V;
    const int acc = device_functors<V, I, C, R>::accumulate(init);
    device_multiply(a, x, y, acc, stream, typename A::format());
}

} // namespace detail

// ---- public overloads (reference cusp/multiply.h:40,101,113,190) ----------------------------------

// y = A*x with explicit functors: y[i] = reduce(initialize(y[i]), ... combine(A_ij, x_j))
template <typename LinearOperator, typename Vector1, typename Vector2, typename UnaryFunction, typename BinaryFunction1, typename BinaryFunction2,
          typename = typename std::enable_if<detail::has_format<LinearOperator>::value>::type>
void multiply(const LinearOperator &A, const Vector1 &x, Vector2 &y, UnaryFunction initialize, BinaryFunction1 combine, BinaryFunction2 reduce)
{
    static_assert(std::is_same<typename LinearOperator::memory_space, typename Vector2::memory_space>::value &&
                      std::is_same<typename Vector1::memory_space, typename Vector2::memory_space>::value,
                  "cusp::multiply: A, x and y must live in the same memory space");
    detail::check_shapes(A, x, y);
    detail::multiply_in_space(A, x, y, initialize, combine, reduce, nullptr, typename LinearOperator::memory_space());
}

// y = A*x (initialize = 0, combine = *, reduce = +: generic/multiply.inl:104-110)
template <typename LinearOperator, typename Vector1, typename Vector2, typename = typename std::enable_if<detail::has_format<LinearOperator>::value>::type>
void multiply(const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    typedef typename Vector2::value_type V;
    cusp::multiply(A, x, y, constant_functor<V>(V(0)), multiplies<V>(), plus<V>());
}
// an operator WITHOUT a storage format -- cusp::linear_operator's children: identity_operator, precond::diagonal, a user's matrix-free operator --
// is applied through its own operator()(x, y) (reference cusp/detail/multiply.inl: the unknown_format branch calls A(x, y)).  Excluded:
// execution policies (the policy-first overloads below) and the sharded operator of cusp/distributed/multiply.h, which has its own overload.
template <typename LinearOperator, typename Vector1, typename Vector2,
          typename = typename std::enable_if<!detail::has_format<LinearOperator>::value && !std::is_base_of<cusp::execution_policy<LinearOperator>, LinearOperator>::value>::type,
          typename = decltype(std::declval<const LinearOperator &>()(std::declval<const Vector1 &>(), std::declval<Vector2 &>()))>
void multiply(const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    A(x, y);
}
// views are cheap handles and are often passed as temporaries
template <typename LinearOperator, typename Vector1, typename T, typename M, typename = typename std::enable_if<detail::has_format<LinearOperator>::value>::type>
void multiply(const LinearOperator &A, const Vector1 &x, array1d_view<T, M> &&y)
{
    cusp::multiply(A, x, y);
}

// with the library's execution policy: selects the HIP stream (device) / is ignored (host)
template <typename LinearOperator, typename Vector1, typename Vector2, typename UnaryFunction, typename BinaryFunction1, typename BinaryFunction2>
void multiply(const cusp::hip::execution_policy &exec, const LinearOperator &A, const Vector1 &x, Vector2 &y, UnaryFunction initialize,
              BinaryFunction1 combine, BinaryFunction2 reduce)
{
    detail::check_shapes(A, x, y);
    detail::multiply_in_space(A, x, y, initialize, combine, reduce, exec.stream(), typename LinearOperator::memory_space());
}
template <typename LinearOperator, typename Vector1, typename Vector2>
void multiply(const cusp::hip::execution_policy &exec, const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    typedef typename Vector2::value_type V;
    cusp::multiply(exec, A, x, y, constant_functor<V>(V(0)), multiplies<V>(), plus<V>());
}

// cusp::omp::par: the reference's OpenMP host backend.  CSR: cusp/system/omp/detail/multiply/csr_spmv.h:51-86 -- rows split
// over the threads (#pragma omp parallel for, :67), per-row arithmetic the sequential kernel's (so the result is
// bit-identical to the sequential multiply); every other format: the sequential loops (omp/detail/multiply.h:26-40).
namespace detail {
template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red>
void omp_multiply(const A &a, const X &x, Y &y, Init initialize, Comb combine, Red reduce, csr_format)
{
    typedef typename A::index_type I;
    typedef typename Y::value_type V;
    const long long n = static_cast<long long>(a.num_rows);
#if defined(_OPENMP)
#pragma omp parallel for
#endif
    for (long long i = 0; i < n; i++) {
        V acc = initialize(y[i]);
        for (I jj = a.row_offsets[i]; jj < a.row_offsets[i + 1]; jj++) acc = reduce(acc, combine(a.values[jj], x[a.column_indices[jj]]));
        y[i] = acc;
    }
}
template <typename A, typename X, typename Y, typename Init, typename Comb, typename Red, typename F>
void omp_multiply(const A &a, const X &x, Y &y, Init i, Comb c, Red r, F f) { host_dispatch(a, x, y, i, c, r, f); }
} // namespace detail

template <typename LinearOperator, typename Vector1, typename Vector2, typename UnaryFunction, typename BinaryFunction1, typename BinaryFunction2>
void multiply(const cusp::omp::execution_policy &, const LinearOperator &A, const Vector1 &x, Vector2 &y, UnaryFunction initialize,
              BinaryFunction1 combine, BinaryFunction2 reduce)
{
    static_assert(std::is_same<typename LinearOperator::memory_space, host_memory>::value, "cusp::omp::par runs on host_memory containers");
    detail::check_shapes(A, x, y);
    detail::omp_multiply(A, x, y, initialize, combine, reduce, typename LinearOperator::format());
}
template <typename LinearOperator, typename Vector1, typename Vector2>
void multiply(const cusp::omp::execution_policy &exec, const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    typedef typename Vector2::value_type V;
    cusp::multiply(exec, A, x, y, constant_functor<V>(V(0)), multiplies<V>(), plus<V>());
}

// z = y + A*x (reference cusp/multiply.h:301,377 generalized_spmv with combine = *, reduce = +)
template <typename LinearOperator, typename Vector1, typename Vector2, typename Vector3, typename BinaryFunction1, typename BinaryFunction2>
void generalized_spmv(const LinearOperator &A, const Vector1 &x, const Vector2 &y, Vector3 &z, BinaryFunction1 combine, BinaryFunction2 reduce)
{
    typedef typename Vector3::value_type V;
    if (static_cast<const void *>(y.data()) != static_cast<const void *>(z.data())) cusp::copy_array(y, z);
    cusp::multiply(A, x, z, identity_function<V>(), combine, reduce);
}

} // namespace cusp

// ---- user-derived execution policies ---------------------------------------------------------------
namespace cusp {
namespace detail {
namespace policy_default {
// what a policy without its own multiply overload gets: the memory-space dispatch above
template <typename Derived, typename LinearOperator, typename Vector1, typename Vector2,
          typename = typename std::enable_if<std::is_base_of<cusp::execution_policy<Derived>, Derived>::value>::type>
void multiply(Derived &, const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    cusp::multiply(A, x, y);
}
} // namespace policy_default
} // namespace detail

// cusp::multiply(exec, A, x, y) for any policy derived from cusp::execution_policy<Derived>
// (reference cusp/detail/multiply.inl:27-39): unqualified call with the DERIVED policy, by reference.
template <typename Derived, typename LinearOperator, typename Vector1, typename Vector2>
void multiply(const cusp::execution_policy<Derived> &exec, const LinearOperator &A, const Vector1 &x, Vector2 &y)
{
    using cusp::detail::policy_default::multiply;
    multiply(const_cast<Derived &>(exec.derived()), A, x, y);
}

} // namespace cusp
