// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never linked by
// the product).
//
// Builds the REFERENCE's own host arithmetic for the SpMV hot path into
// oracle/_ref/libcusp_ref.so.  The five sequential kernel headers
//     cusp/system/detail/sequential/multiply/{csr,coo,ell,dia,hyb}_spmv.h
// are included from where they lie under /root/reference (nothing is copied),
// and instantiated with the duck-typed, non-owning containers below: the
// reference kernels are templates that only need num_rows/num_cols/num_entries,
// index_type/value_type, operator[] on the 1-D arrays, operator()(i,j) +
// num_cols on the column-major 2-D arrays, a static invalid_index and the
// ell/coo members of a HYB matrix.  The public cusp::multiply and the CUSP
// containers themselves do not build in this image (they need KTT's <Ktt.h>,
// <cuda.h> and Thrust 1.x internals -- see DESIGN.md), so the kernels are called
// at the backend-overload level, exactly as generic::multiply would
// (cusp/system/detail/generic/multiply.inl:173-191).
//
// Build (oracle/Makefile, target ref): host code only, no GPU needed:
//   hipcc -std=c++17 -O3 -ffp-contract=off -DTHRUST_DEVICE_SYSTEM=THRUST_DEVICE_SYSTEM_CPP
//         -I/root/reference -shared -fPIC oracle/ref_shim.cpp -o oracle/_ref/libcusp_ref.so
#include <cstddef>
#include <cstdint>

#include <thrust/functional.h>
#include <thrust/system/cpp/execution_policy.h>

#include <cusp/system/detail/sequential/multiply/csr_spmv.h>
#include <cusp/system/detail/sequential/multiply/coo_spmv.h>
#include <cusp/system/detail/sequential/multiply/ell_spmv.h>
#include <cusp/system/detail/sequential/multiply/dia_spmv.h>
#include <cusp/system/detail/sequential/multiply/hyb_spmv.h>

namespace shim {

template <typename T> struct span1d {
    typedef T value_type;
    T *p;
    size_t n;
    T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};

// column-major 2-D view, element (i,j) at j*pitch+i (what cusp::array2d<...,column_major> does)
template <typename T> struct span2d {
    typedef T value_type;
    T *p;
    size_t num_rows, num_cols, pitch;
    T &operator()(size_t i, size_t j) const { return p[j * pitch + i]; }
};

template <typename I, typename V> struct csr_view {
    typedef I index_type; typedef V value_type;
    size_t num_rows, num_cols, num_entries;
    span1d<const I> row_offsets, column_indices;
    span1d<const V> values;
};
template <typename I, typename V> struct coo_view {
    typedef I index_type; typedef V value_type;
    size_t num_rows, num_cols, num_entries;
    span1d<const I> row_indices, column_indices;
    span1d<const V> values;
};
template <typename I, typename V> struct ell_view {
    typedef I index_type; typedef V value_type;
    static const I invalid_index = static_cast<I>(-1);
    size_t num_rows, num_cols, num_entries;
    span2d<const I> column_indices;
    span2d<const V> values;
};
template <typename I, typename V> struct dia_view {
    typedef I index_type; typedef V value_type;
    size_t num_rows, num_cols, num_entries;
    span1d<const I> diagonal_offsets;
    span2d<const V> values;
};
template <typename I, typename V> struct hyb_view {
    typedef I index_type; typedef V value_type;
    size_t num_rows, num_cols, num_entries;
    ell_view<I, V> ell;
    coo_view<I, V> coo;
};

template <typename V> struct zero_init { V operator()(const V &) const { return V(0); } };
template <typename V> struct ident_init { V operator()(const V &v) const { return v; } };

template <typename A, typename V, typename Format>
void run(const A &a, const V *x, size_t nx, V *y, size_t ny, int accumulate, Format fmt)
{
    namespace seq = cusp::system::detail::sequential;
    thrust::cpp::tag exec;
    span1d<const V> xs{x, nx};
    span1d<V> ys{y, ny};
    if (accumulate)
        seq::multiply(exec, a, xs, ys, ident_init<V>(), thrust::multiplies<V>(), thrust::plus<V>(), fmt,
                      cusp::array1d_format(), cusp::array1d_format());
    else
        seq::multiply(exec, a, xs, ys, zero_init<V>(), thrust::multiplies<V>(), thrust::plus<V>(), fmt,
                      cusp::array1d_format(), cusp::array1d_format());
}

} // namespace shim

#define REF_API extern "C" __attribute__((visibility("default")))

#define REF_DEFINE(T, SUF)                                                                              \
    REF_API void ref_spmv_csr_##SUF(int64_t rows, int64_t cols, int64_t nnz, const int32_t *Ap,         \
                                    const int32_t *Aj, const T *Ax, const T *x, T *y, int accumulate)   \
    {                                                                                                   \
        shim::csr_view<int32_t, T> A{(size_t)rows, (size_t)cols, (size_t)nnz,                           \
                                     {Ap, (size_t)rows + 1}, {Aj, (size_t)nnz}, {Ax, (size_t)nnz}};     \
        shim::run(A, x, (size_t)cols, y, (size_t)rows, accumulate, cusp::csr_format());                 \
    }                                                                                                   \
    REF_API void ref_spmv_coo_##SUF(int64_t rows, int64_t cols, int64_t nnz, const int32_t *Ai,         \
                                    const int32_t *Aj, const T *Ax, const T *x, T *y, int accumulate)   \
    {                                                                                                   \
        shim::coo_view<int32_t, T> A{(size_t)rows, (size_t)cols, (size_t)nnz,                           \
                                     {Ai, (size_t)nnz}, {Aj, (size_t)nnz}, {Ax, (size_t)nnz}};          \
        shim::run(A, x, (size_t)cols, y, (size_t)rows, accumulate, cusp::coo_format());                 \
    }                                                                                                   \
    REF_API void ref_spmv_ell_##SUF(int64_t rows, int64_t cols, int64_t width, int64_t pitch,           \
                                    const int32_t *Aj, const T *Ax, const T *x, T *y, int accumulate)   \
    {                                                                                                   \
        shim::ell_view<int32_t, T> A{(size_t)rows, (size_t)cols, 0,                                     \
                                     {Aj, (size_t)rows, (size_t)width, (size_t)pitch},                  \
                                     {Ax, (size_t)rows, (size_t)width, (size_t)pitch}};                 \
        shim::run(A, x, (size_t)cols, y, (size_t)rows, accumulate, cusp::ell_format());                 \
    }                                                                                                   \
    REF_API void ref_spmv_dia_##SUF(int64_t rows, int64_t cols, int64_t ndiag, int64_t pitch,           \
                                    const int32_t *offsets, const T *vals, const T *x, T *y,            \
                                    int accumulate)                                                     \
    {                                                                                                   \
        shim::dia_view<int32_t, T> A{(size_t)rows, (size_t)cols, 0, {offsets, (size_t)ndiag},           \
                                     {vals, (size_t)rows, (size_t)ndiag, (size_t)pitch}};               \
        shim::run(A, x, (size_t)cols, y, (size_t)rows, accumulate, cusp::dia_format());                 \
    }                                                                                                   \
    REF_API void ref_spmv_hyb_##SUF(int64_t rows, int64_t cols, int64_t width, int64_t pitch,           \
                                    const int32_t *ell_Aj, const T *ell_Ax, int64_t coo_nnz,            \
                                    const int32_t *coo_Ai, const int32_t *coo_Aj, const T *coo_Ax,      \
                                    const T *x, T *y, int accumulate)                                   \
    {                                                                                                   \
        shim::hyb_view<int32_t, T> A{(size_t)rows, (size_t)cols, 0,                                     \
            {(size_t)rows, (size_t)cols, 0, {ell_Aj, (size_t)rows, (size_t)width, (size_t)pitch},       \
                                            {ell_Ax, (size_t)rows, (size_t)width, (size_t)pitch}},      \
            {(size_t)rows, (size_t)cols, (size_t)coo_nnz, {coo_Ai, (size_t)coo_nnz},                    \
                                            {coo_Aj, (size_t)coo_nnz}, {coo_Ax, (size_t)coo_nnz}}};     \
        shim::run(A, x, (size_t)cols, y, (size_t)rows, accumulate, cusp::hyb_format());                 \
    }

REF_DEFINE(double, f64)
REF_DEFINE(float, f32)

REF_API const char *ref_describe(void)
{
    return "reference cusp/system/detail/sequential/multiply/{csr,coo,ell,dia,hyb}_spmv.h, "
           "compiled from /root/reference, -O3 -ffp-contract=off, single thread";
}
