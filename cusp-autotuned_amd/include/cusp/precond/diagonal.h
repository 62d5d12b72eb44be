// cusp/precond/diagonal.h -- cusp::precond::diagonal<ValueType, MemorySpace>: the Jacobi preconditioner, y <- D^-1 x
// (reference cusp/precond/diagonal.h:85-108, detail/diagonal.inl: extract_diagonal, reciprocals, operator() = blas::xmy).
// Built once from any of the five formats (set-up: the diagonal is read from a host copy of the matrix and the reciprocals are placed in
// MemorySpace); applying it is ONE elementwise kernel (cmi_blas_xmy_*), which is what makes cusp::krylov::cg(A, x, b, monitor, M) with this M
// a device-resident solve: the multiply through A's plan + the library's BLAS-1 per iteration.
#pragma once
#include "../blas/blas.h"
#include "../csr_matrix.h"
#include "../format_utils.h"
#include "../linear_operator.h"

namespace cusp {
namespace precond {

template <typename ValueType, typename MemorySpace> class diagonal : public cusp::linear_operator<ValueType, MemorySpace> {
    typedef cusp::linear_operator<ValueType, MemorySpace> Parent;
    cusp::array1d<ValueType, MemorySpace> diagonal_reciprocals;

public:
    template <typename MatrixType> diagonal(const MatrixType &A) : Parent(A.num_rows, A.num_cols, A.num_rows), diagonal_reciprocals(A.num_rows)
    {
        build(A, std::integral_constant<bool, on_device<MatrixType>::value>());
    }
private:
    // a device_memory matrix of int indices and this value type: the diagonal is read ON the device (cmi_csr_diagonal_*; the other formats are converted
    // to CSR in HBM first); everything else: from a host copy
    template <typename MatrixType> struct on_device {
        static const bool value = std::is_same<MemorySpace, cusp::device_memory>::value && std::is_same<typename MatrixType::memory_space, cusp::device_memory>::value &&
                                  std::is_same<typename MatrixType::index_type, int>::value && std::is_same<typename MatrixType::value_type, ValueType>::value &&
                                  (std::is_same<ValueType, double>::value || std::is_same<ValueType, float>::value) &&
                                  !std::is_same<typename MatrixType::format, cusp::array2d_format>::value;
    };
    static int diag_call(int64_t n, const int *Ap, const int *Aj, const double *Ax, double *d) { return cmi_csr_diagonal_f64(n, Ap, Aj, Ax, d, 1, nullptr); }
    static int diag_call(int64_t n, const int *Ap, const int *Aj, const float *Ax, float *d) { return cmi_csr_diagonal_f32(n, Ap, Aj, Ax, d, 1, nullptr); }
    template <typename T> static int diag_call(int64_t, const int *, const int *, const T *, T *) { return -1; }
    template <typename Csr> void from_device_csr(const Csr &C)
    {
        cusp::detail::check(diag_call((int64_t)C.num_rows, C.row_offsets.data(), C.column_indices.data(), C.values.data(), diagonal_reciprocals.data()));
        cusp::detail::check(cmi_stream_synchronize(nullptr));
    }
    template <typename MatrixType> void build_device(const MatrixType &A, cusp::csr_format) { from_device_csr(A); }
    template <typename MatrixType, typename Format> void build_device(const MatrixType &A, Format)
    {
        cusp::csr_matrix<int, ValueType, cusp::device_memory> C(A); // (device conversions: the matrix never visits the host)
        from_device_csr(C);
    }
    template <typename MatrixType> void build(const MatrixType &A, std::true_type) { build_device(A, typename MatrixType::format()); }
    template <typename MatrixType> void build(const MatrixType &A, std::false_type)
    {
        cusp::array1d<ValueType, cusp::host_memory> d(A.num_rows);
        cusp::extract_diagonal(A, d);
        for (size_t i = 0; i < d.size(); i++) d[i] = ValueType(1) / d[i]; // (a zero on the diagonal gives inf, as in the reference)
        diagonal_reciprocals = d;
    }

public:
    // 1 / a_ii (what cusp::krylov::cg's fused Jacobi path on device_memory reads instead of storing z = D^-1 r)
    const cusp::array1d<ValueType, MemorySpace> &reciprocals() const { return diagonal_reciprocals; }
    template <typename VectorType1, typename VectorType2> void operator()(const VectorType1 &x, VectorType2 &y) const
    {
        cusp::blas::xmy(diagonal_reciprocals, x, y);
    }
};

} // namespace precond
} // namespace cusp
