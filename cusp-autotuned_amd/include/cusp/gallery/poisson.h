// cusp/gallery/poisson.h -- cusp::gallery::poisson5pt(matrix, m, n)
// (reference cusp/gallery/poisson.h, detail/poisson.inl:29-47, detail/stencil.inl:118-206):
// the 5-point stencil (0,-1),(-1,0),(0,0),(1,0),(0,1) = -1,-1,4,-1,-1 on an m x n grid, strides
// (1, m) => diagonals [-m,-1,0,1,m]; a DIA matrix with pitch = num_rows whose entries are the stencil
// value where the neighbour is inside the grid, else 0; other formats by conversion.
// device_memory CSR / DIA targets are built directly in HBM by the C-ABI (cmi_poisson5pt_*), ELL / COO
// through the on-device CSR conversions; results are identical to the host path.
#pragma once
#include <vector>

#include "../convert.h"
#include "../detail/matrices.h"

namespace cusp {
namespace gallery {

namespace detail {

template <typename I, typename V> void poisson5pt_host_dia(dia_matrix<I, V, host_memory> &dia, size_t m, size_t n)
{
    const size_t N = m * n;
    dia.resize(N, N, 0, 5, 1); // alignment 1: pitch = num_rows (stencil.inl:174)
    const long long dx[5] = {0, -1, 0, 1, 0}, dy[5] = {-1, 0, 0, 0, 1};
    const V sv[5] = {V(-1), V(-1), V(4), V(-1), V(-1)};
    size_t nnz = 0;
    for (int d = 0; d < 5; d++) {
        dia.diagonal_offsets[d] = static_cast<I>(dx[d] + dy[d] * static_cast<long long>(m));
        for (size_t r = 0; r < N; r++) {
            const long long ix = static_cast<long long>(r % m) + dx[d], iy = static_cast<long long>(r / m) + dy[d];
            const bool inside = ix >= 0 && ix < static_cast<long long>(m) && iy >= 0 && iy < static_cast<long long>(n);
            dia.values(r, d) = inside ? sv[d] : V(0);
            nnz += inside;
        }
    }
    dia.num_entries = nnz;
}

inline int build_csr(int64_t m, int64_t n, int *Ap, int *Aj, double *Ax) { return cmi_poisson5pt_csr_f64(m, n, 0, m * n, Ap, Aj, Ax, nullptr); }
inline int build_csr(int64_t m, int64_t n, int *Ap, int *Aj, float *Ax) { return cmi_poisson5pt_csr_f32(m, n, 0, m * n, Ap, Aj, Ax, nullptr); }
inline int build_dia(int64_t m, int64_t n, int64_t pitch, int *off, double *v) { return cmi_poisson5pt_dia_f64(m, n, pitch, off, v, nullptr); }
inline int build_dia(int64_t m, int64_t n, int64_t pitch, int *off, float *v) { return cmi_poisson5pt_dia_f32(m, n, pitch, off, v, nullptr); }

template <typename V> void poisson5pt_device_csr(csr_matrix<int, V, device_memory> &A, size_t m, size_t n)
{
    const size_t N = m * n, nnz = static_cast<size_t>(cmi_poisson5pt_num_entries(m, n));
    A.resize(N, N, nnz);
    cusp::detail::check(build_csr(m, n, A.row_offsets.data(), A.column_indices.data(), A.values.data()));
    cusp::detail::check(cmi_stream_synchronize(nullptr));
}

// generic: host DIA then convert (what the reference does for every format)
template <typename Matrix> void poisson5pt_impl(Matrix &A, size_t m, size_t n, host_memory)
{
    dia_matrix<typename Matrix::index_type, typename Matrix::value_type, host_memory> dia;
    poisson5pt_host_dia(dia, m, n);
    cusp::convert(dia, A);
}
template <typename V> void poisson5pt_impl(csr_matrix<int, V, device_memory> &A, size_t m, size_t n, device_memory) { poisson5pt_device_csr(A, m, n); }
template <typename V> void poisson5pt_impl(dia_matrix<int, V, device_memory> &A, size_t m, size_t n, device_memory)
{
    const size_t N = m * n;
    A.resize(N, N, static_cast<size_t>(cmi_poisson5pt_num_entries(m, n)), 5, 1);
    cusp::detail::check(build_dia(m, n, A.values.pitch, A.diagonal_offsets.data(), A.values.values.data()));
    cusp::detail::check(cmi_stream_synchronize(nullptr));
}
template <typename Matrix> void poisson5pt_impl(Matrix &A, size_t m, size_t n, device_memory)
{
    csr_matrix<int, typename Matrix::value_type, device_memory> csr;
    poisson5pt_device_csr(csr, m, n);
    cusp::convert(csr, A); // ELL / COO: on-device; HYB: host heuristic + split
}

} // namespace detail

template <typename MatrixType> void poisson5pt(MatrixType &matrix, size_t m, size_t n)
{
    detail::poisson5pt_impl(matrix, m, n, typename MatrixType::memory_space());
}

// ---- generic stencils (reference cusp/gallery/stencil.h, detail/stencil.inl:143-206) ----------------
// A stencil point = (offset in each grid dimension, value).  Built as a host DIA matrix (one diagonal per
// stencil point, in stencil order, pitch = num_rows) and converted to the requested type -- exactly the
// reference's route.  Used here for the FEM-like surrogates of the SuiteSparse set (27-point: ~27
// entries/row like nlpkkt120; 7- and 9-point: thermal2-like row lengths).
struct stencil_point { long long dx, dy, dz; double value; };

template <typename MatrixType>
void generate_matrix_from_stencil(MatrixType &matrix, const std::vector<stencil_point> &stencil, size_t nx, size_t ny, size_t nz = 1)
{
    typedef typename MatrixType::index_type I;
    typedef typename MatrixType::value_type V;
    const size_t N = nx * ny * nz;
    dia_matrix<I, V, host_memory> dia;
    dia.resize(N, N, 0, stencil.size(), 1);
    size_t nnz = 0;
    for (size_t d = 0; d < stencil.size(); d++) {
        const stencil_point &sp = stencil[d];
        dia.diagonal_offsets[d] = static_cast<I>(sp.dx + sp.dy * (long long)nx + sp.dz * (long long)(nx * ny));
        for (size_t r = 0; r < N; r++) {
            const long long ix = (long long)(r % nx) + sp.dx, iy = (long long)((r / nx) % ny) + sp.dy, iz = (long long)(r / (nx * ny)) + sp.dz;
            const bool inside = ix >= 0 && ix < (long long)nx && iy >= 0 && iy < (long long)ny && iz >= 0 && iz < (long long)nz;
            dia.values(r, d) = inside ? static_cast<V>(sp.value) : V(0);
            nnz += inside && sp.value != 0.0;
        }
    }
    dia.num_entries = nnz;
    cusp::convert(dia, matrix);
}

// reference cusp/gallery/detail/poisson.inl:49-120
template <typename MatrixType> void poisson9pt(MatrixType &matrix, size_t m, size_t n)
{
    std::vector<stencil_point> st;
    for (long long j = -1; j <= 1; j++)
        for (long long i = -1; i <= 1; i++) st.push_back({i, j, 0, (i == 0 && j == 0) ? 8.0 : -1.0});
    generate_matrix_from_stencil(matrix, st, m, n);
}
template <typename MatrixType> void poisson7pt(MatrixType &matrix, size_t m, size_t n, size_t k)
{
    const std::vector<stencil_point> st = {{0, 0, -1, -1.0}, {0, -1, 0, -1.0}, {-1, 0, 0, -1.0}, {0, 0, 0, 6.0}, {1, 0, 0, -1.0}, {0, 1, 0, -1.0}, {0, 0, 1, -1.0}};
    generate_matrix_from_stencil(matrix, st, m, n, k);
}
template <typename MatrixType> void poisson27pt(MatrixType &matrix, size_t m, size_t n, size_t l)
{
    std::vector<stencil_point> st;
    for (long long k = -1; k <= 1; k++)
        for (long long j = -1; j <= 1; j++)
            for (long long i = -1; i <= 1; i++) st.push_back({i, j, k, (i == 0 && j == 0 && k == 0) ? 26.0 : -1.0});
    generate_matrix_from_stencil(matrix, st, m, n, l);
}

} // namespace gallery
} // namespace cusp
