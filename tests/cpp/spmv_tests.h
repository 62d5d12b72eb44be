// Tests of the SpMV path written the way the reference's own tests are (same matrices, same
// protocol), parametrised on TEST_SPACE so the host build and the device build share them:
//   testing/multiply.cu:383-512,569-645  TestSparseMatrixVectorMultiply / Scaled...
//   testing/generalized_spmv.cu:20-70    known answer z = [183,74,325,510,131]
//   testing/convert.cu:65-215,402-497    the 4x4 conversion example in every format
//   testing/ell_matrix.cu:5-22           constructor pitch / alignment
//   testing/poisson.cu:6-25              poisson5pt(2,3) as a dense matrix
//   testing/cg.cu:46-99                  CG on poisson5pt(10,10) and the zero-residual start
//   testing/multiply.cu:792-858          user execution policy reaches a user overload
#pragma once
#include <cstdlib>
#include <map>
#include <cusp/array1d.h>
#include <cusp/array2d.h>
#include <cusp/coo_matrix.h>
#include <cusp/csr_matrix.h>
#include <cusp/dia_matrix.h>
#include <cusp/ell_matrix.h>
#include <cusp/hyb_matrix.h>
#include <cusp/gallery/poisson.h>
#include <cusp/copy.h>
#include <cusp/io/matrix_market.h>
#include <cusp/print.h>
#include <cusp/krylov/cg.h>
#include <cusp/krylov/bicgstab.h>
#include <cusp/krylov/cr.h>
#include <cusp/krylov/bicg.h>
#include <cusp/transpose.h>
#include <cusp/krylov/gmres.h>
#include <cusp/precond/diagonal.h>
#include <cusp/ktt/ktt.h>
#include <cusp/monitor.h>
#include <cusp/sort.h>
#include <cusp/format_utils.h>
#include <cusp/verify.h>
#include <cusp/multiply.h>

#include "unittest.h"

template <typename A, typename B> bool arrays_equal(const A &a, const B &b) { return cusp::equal(a, b); }
#define ASSERT_ARRAYS_EQUAL(a, b) ASSERT_TRUE(arrays_equal(a, b))

// ------------------------------------------------------------------------------------------------
// testing/multiply.cu:383-432
template <typename SparseMatrixType, typename DenseMatrixType> void CompareSparseMatrixVectorMultiply(DenseMatrixType A)
{
    typedef typename SparseMatrixType::value_type ValueType;
    typedef typename SparseMatrixType::memory_space MemorySpace;
    cusp::array1d<ValueType, cusp::host_memory> x(A.num_cols);
    cusp::array1d<ValueType, cusp::host_memory> y(A.num_rows, 10);
    for (size_t i = 0; i < x.size(); i++) x[i] = i % 10;
    cusp::multiply(A, x, y); // dense host reference
    { // container
        SparseMatrixType _A(A);
        cusp::array1d<ValueType, MemorySpace> _x(x);
        cusp::array1d<ValueType, MemorySpace> _y(A.num_rows, 10);
        cusp::multiply(_A, _x, _y);
        ASSERT_ARRAYS_EQUAL(_y, y);
    }
    { // matrix view
        SparseMatrixType _A(A);
        cusp::array1d<ValueType, MemorySpace> _x(x);
        cusp::array1d<ValueType, MemorySpace> _y(A.num_rows, 10);
        typename SparseMatrixType::view _V(_A);
        cusp::multiply(_V, _x, _y);
        ASSERT_ARRAYS_EQUAL(_y, y);
    }
    { // array views
        SparseMatrixType _A(A);
        cusp::array1d<ValueType, MemorySpace> _x(x);
        cusp::array1d<ValueType, MemorySpace> _y(A.num_rows, 10);
        typename cusp::array1d<ValueType, MemorySpace>::view _Vx(_x), _Vy(_y);
        cusp::multiply(_A, _Vx, _Vy);
        ASSERT_ARRAYS_EQUAL(_Vy, y);
    }
    { // explicit policy on the default stream
        SparseMatrixType _A(A);
        cusp::array1d<ValueType, MemorySpace> _x(x);
        cusp::array1d<ValueType, MemorySpace> _y(A.num_rows, 10);
        cusp::multiply(cusp::hip::par, _A, _x, _y);
        ASSERT_ARRAYS_EQUAL(_y, y);
    }
}

// testing/multiply.cu:514-567
template <typename SparseMatrixType, typename DenseMatrixType> void CompareScaledSparseMatrixVectorMultiply(DenseMatrixType A)
{
    typedef typename SparseMatrixType::value_type ValueType;
    typedef typename SparseMatrixType::memory_space MemorySpace;
    cusp::array1d<ValueType, cusp::host_memory> x(A.num_cols);
    cusp::array1d<ValueType, cusp::host_memory> y(A.num_rows, 10);
    for (size_t i = 0; i < x.size(); i++) x[i] = i % 10;
    cusp::identity_function<ValueType> initialize;
    cusp::multiplies<ValueType> combine;
    cusp::plus<ValueType> reduce;
    cusp::multiply(A, x, y, initialize, combine, reduce);
    SparseMatrixType _A(A);
    cusp::array1d<ValueType, MemorySpace> _x(x);
    cusp::array1d<ValueType, MemorySpace> _y(A.num_rows, 10);
    cusp::multiply(_A, _x, _y, initialize, combine, reduce);
    ASSERT_ARRAYS_EQUAL(_y, y);
}

template <typename ValueType, typename Fn> void for_each_reference_matrix(Fn f)
{
    typedef cusp::array2d<ValueType, cusp::host_memory> Dense;
    Dense A(5, 4);
    const ValueType a[5][4] = {{13, 80, 0, 0}, {0, 27, 0, 0}, {55, 0, 24, 42}, {0, 69, 0, 83}, {0, 0, 27, 0}};
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) A(i, j) = a[i][j];
    Dense B(2, 4);
    const ValueType b[2][4] = {{0, 2, 3, 4}, {5, 0, 0, 8}};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) B(i, j) = b[i][j];
    Dense C(2, 2); C(0, 0) = 0; C(0, 1) = 0; C(1, 0) = 3; C(1, 1) = 5;
    Dense D(2, 1); D(0, 0) = 2; D(1, 0) = 3;
    Dense E(2, 2, ValueType(0));
    Dense F(2, 3); F(0, 0) = 0; F(0, 1) = 1.5; F(0, 2) = 3.0; F(1, 0) = 0.5; F(1, 1) = 0; F(1, 2) = 0;
    Dense G; cusp::gallery::poisson5pt(G, 4, 6);
    Dense H; cusp::gallery::poisson5pt(H, 8, 3);
    f(A); f(B); f(C); f(D); f(E); f(F); f(G); f(H);
}

template <class TestMatrix> void TestSparseMatrixVectorMultiply()
{
    typedef typename TestMatrix::value_type V;
    for_each_reference_matrix<V>([](const cusp::array2d<V, cusp::host_memory> &M) { CompareSparseMatrixVectorMultiply<TestMatrix>(M); });
}
DECLARE_SPARSE_MATRIX_UNITTEST(TestSparseMatrixVectorMultiply);

template <class TestMatrix> void TestScaledSparseMatrixVectorMultiply()
{
    typedef typename TestMatrix::value_type V;
    for_each_reference_matrix<V>([](const cusp::array2d<V, cusp::host_memory> &M) { CompareScaledSparseMatrixVectorMultiply<TestMatrix>(M); });
}
DECLARE_SPARSE_MATRIX_UNITTEST(TestScaledSparseMatrixVectorMultiply);

// testing/generalized_spmv.cu:20-70
template <class TestMatrix> void TestGeneralizedSpMVKnownAnswer()
{
    typedef typename TestMatrix::value_type V;
    typedef typename TestMatrix::memory_space MemorySpace;
    cusp::array2d<V, cusp::host_memory> A(5, 4);
    const V a[5][4] = {{13, 80, 0, 0}, {0, 27, 0, 0}, {55, 0, 24, 42}, {0, 69, 0, 83}, {0, 0, 27, 0}};
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) A(i, j) = a[i][j];
    TestMatrix test_matrix = A;
    cusp::array1d<V, MemorySpace> x(4), y(5), z(5, -1);
    x[0] = 1; x[1] = 2; x[2] = 3; x[3] = 4;
    y[0] = 10; y[1] = 20; y[2] = 30; y[3] = 40; y[4] = 50;
    cusp::generalized_spmv(test_matrix, x, y, z, cusp::multiplies<V>(), cusp::plus<V>());
    ASSERT_EQUAL(V(z[0]), V(183)); ASSERT_EQUAL(V(z[1]), V(74)); ASSERT_EQUAL(V(z[2]), V(325));
    ASSERT_EQUAL(V(z[3]), V(510)); ASSERT_EQUAL(V(z[4]), V(131));
}
DECLARE_SPARSE_MATRIX_UNITTEST(TestGeneralizedSpMVKnownAnswer);

// testing/generalized_spmv.cu:72-143: multiply vs generalized_spmv on poisson grids incl. odd sizes
template <class TestMatrix> void TestPoissonSizesAgainstHostCsr()
{
    typedef typename TestMatrix::value_type V;
    typedef typename TestMatrix::memory_space MemorySpace;
    const size_t sizes[][2] = {{5, 5}, {10, 10}, {117, 113}, {313, 444}};
    for (auto &s : sizes) {
        cusp::csr_matrix<int, V, cusp::host_memory> H;
        cusp::gallery::poisson5pt(H, s[0], s[1]);
        const size_t N = s[0] * s[1];
        ASSERT_EQUAL(H.num_entries, 5 * N - 2 * s[0] - 2 * s[1]);
        cusp::array1d<V, cusp::host_memory> x(N), y(N, 10);
        for (size_t i = 0; i < N; i++) x[i] = V(int(i % 21) - 10); // performance/spmv/benchmark.h x pattern
        cusp::multiply(H, x, y);
        TestMatrix A;
        cusp::gallery::poisson5pt(A, s[0], s[1]); // device targets: built in HBM
        cusp::array1d<V, MemorySpace> _x(x), _y(N, 10);
        cusp::multiply(A, _x, _y);
        ASSERT_ARRAYS_EQUAL(_y, y); // small integers: exact in every summation order
    }
}
DECLARE_SPARSE_MATRIX_UNITTEST(TestPoissonSizesAgainstHostCsr);

// ------------------------------------------------------------------------------------------------
// testing/convert.cu:65-215: the canonical 4x4 example
template <typename M> void initialize_conversion_example(cusp::csr_matrix<int, float, M> &csr)
{
    csr.resize(4, 4, 7);
    const int ro[5] = {0, 2, 3, 6, 7}, ci[7] = {0, 1, 2, 0, 2, 3, 1};
    const float v[7] = {10.25f, 11.00f, 12.50f, 13.75f, 14.00f, 15.25f, 16.50f};
    for (int i = 0; i < 5; i++) csr.row_offsets[i] = ro[i];
    for (int i = 0; i < 7; i++) { csr.column_indices[i] = ci[i]; csr.values[i] = v[i]; }
}

template <typename Space> void TestConversionExampleAllFormats()
{
    cusp::csr_matrix<int, float, Space> csr;
    initialize_conversion_example(csr);
    const int X = cusp::ell_matrix<int, float, Space>::invalid_index;
    ASSERT_EQUAL(X, -1);
    { // CSR -> COO
        cusp::coo_matrix<int, float, Space> coo(csr);
        const int ri[7] = {0, 0, 1, 2, 2, 2, 3};
        ASSERT_EQUAL(coo.num_entries, size_t(7));
        for (int i = 0; i < 7; i++) { ASSERT_EQUAL(int(coo.row_indices[i]), ri[i]); ASSERT_EQUAL(int(coo.column_indices[i]), int(csr.column_indices[i])); }
        ASSERT_TRUE(coo.is_sorted_by_row() && coo.is_sorted_by_row_and_column());
        cusp::csr_matrix<int, float, Space> back(coo);
        ASSERT_ARRAYS_EQUAL(back.row_offsets, csr.row_offsets);
        ASSERT_ARRAYS_EQUAL(back.values, csr.values);
    }
    { // CSR -> DIA with alignment 1 (testing/convert.cu:402-438)
        cusp::dia_matrix<int, float, cusp::host_memory> dia;
        cusp::csr_matrix<int, float, cusp::host_memory> h(csr);
        cusp::detail::from_host_csr(h, dia, cusp::dia_format(), 1);
        ASSERT_EQUAL(dia.num_entries, size_t(7));
        ASSERT_EQUAL(int(dia.diagonal_offsets[0]), -2); ASSERT_EQUAL(int(dia.diagonal_offsets[1]), 0); ASSERT_EQUAL(int(dia.diagonal_offsets[2]), 1);
        const float e[12] = {0, 0, 13.75f, 16.50f, 10.25f, 0, 14.00f, 0, 11.00f, 12.50f, 15.25f, 0};
        for (int i = 0; i < 12; i++) ASSERT_EQUAL(float(dia.values.values[i]), e[i]);
        cusp::dia_matrix<int, float, Space> d2(dia); // H->D (or copy)
        cusp::csr_matrix<int, float, Space> back(d2);
        ASSERT_ARRAYS_EQUAL(back.column_indices, csr.column_indices);
        ASSERT_ARRAYS_EQUAL(back.values, csr.values);
    }
    { // CSR -> ELL with alignment 1 (testing/convert.cu:440-497)
        cusp::ell_matrix<int, float, cusp::host_memory> ell;
        cusp::csr_matrix<int, float, cusp::host_memory> h(csr);
        cusp::detail::from_host_csr(h, ell, cusp::ell_format(), 3, 1);
        ASSERT_EQUAL(ell.column_indices.num_rows, size_t(4)); ASSERT_EQUAL(ell.column_indices.num_cols, size_t(3));
        const int ec[12] = {0, 2, 0, 1, 1, X, 2, X, X, X, 3, X};
        const float ev[12] = {10.25f, 12.50f, 13.75f, 16.50f, 11.00f, 0, 14.00f, 0, 0, 0, 15.25f, 0};
        for (int i = 0; i < 12; i++) { ASSERT_EQUAL(int(ell.column_indices.values[i]), ec[i]); ASSERT_EQUAL(float(ell.values.values[i]), ev[i]); }
    }
    { // default conversions keep the matrix: every format -> CSR gives the example back
        cusp::ell_matrix<int, float, Space> ell(csr);
        ASSERT_EQUAL(ell.column_indices.pitch, size_t(32)); // default alignment 32
        cusp::hyb_matrix<int, float, Space> hyb(csr);
        cusp::dia_matrix<int, float, Space> dia(csr);
        cusp::coo_matrix<int, float, Space> coo(csr);
        cusp::csr_matrix<int, float, Space> a(ell), b(hyb), c(dia), d(coo);
        for (auto *m : {&a, &b, &c, &d}) {
            ASSERT_ARRAYS_EQUAL(m->row_offsets, csr.row_offsets);
            ASSERT_ARRAYS_EQUAL(m->column_indices, csr.column_indices);
            ASSERT_ARRAYS_EQUAL(m->values, csr.values);
        }
        // HYB with ELL width 1: testing/convert.cu:177-200
        cusp::hyb_matrix<int, float, cusp::host_memory> h1;
        cusp::csr_matrix<int, float, cusp::host_memory> h(csr);
        cusp::detail::from_host_csr(h, h1, cusp::hyb_format(), 1, 1);
        ASSERT_EQUAL(h1.ell.num_entries, size_t(4)); ASSERT_EQUAL(h1.coo.num_entries, size_t(3));
        const int cr[3] = {0, 2, 2}, cc[3] = {1, 2, 3};
        for (int i = 0; i < 3; i++) { ASSERT_EQUAL(int(h1.coo.row_indices[i]), cr[i]); ASSERT_EQUAL(int(h1.coo.column_indices[i]), cc[i]); }
    }
}
DECLARE_SPACE_UNITTEST(TestConversionExampleAllFormats);

// CSR -> DIA in the matrix's own memory space (device: cmi_csr_diagonals + cmi_csr_to_dia, only the sorted
// offsets visit the host) must produce the arrays of the host conversion (csr_to_other.h:73-153)
template <typename Space> void TestCsrToDiaMatchesHostConversion()
{
    cusp::csr_matrix<int, double, cusp::host_memory> h;
    cusp::gallery::poisson5pt(h, 37, 23);
    // a rectangular banded matrix with an empty row and an empty diagonal in between
    cusp::csr_matrix<int, double, cusp::host_memory> r(50, 70, 0);
    {
        std::vector<int> ro(1, 0), ci; std::vector<double> v;
        for (int i = 0; i < 50; i++) {
            if (i != 17)
                for (int o : {-3, 0, 2, 25}) { const int j = i + o; if (j >= 0 && j < 70) { ci.push_back(j); v.push_back(1.0 + i + 0.25 * o); } }
            ro.push_back(int(ci.size()));
        }
        r.resize(50, 70, ci.size());
        for (size_t k = 0; k < ro.size(); k++) r.row_offsets[k] = ro[k];
        for (size_t k = 0; k < ci.size(); k++) { r.column_indices[k] = ci[k]; r.values[k] = v[k]; }
    }
    for (const auto *src : {&h, &r}) {
        cusp::dia_matrix<int, double, cusp::host_memory> want(*src);
        cusp::csr_matrix<int, double, Space> s(*src);
        cusp::dia_matrix<int, double, Space> got(s);
        ASSERT_EQUAL(got.num_rows, want.num_rows); ASSERT_EQUAL(got.num_cols, want.num_cols); ASSERT_EQUAL(got.num_entries, want.num_entries);
        ASSERT_EQUAL(got.values.pitch, want.values.pitch);
        ASSERT_ARRAYS_EQUAL(got.diagonal_offsets, want.diagonal_offsets);
        ASSERT_ARRAYS_EQUAL(got.values.values, want.values.values);
    }
    // too many diagonals for the fill-in guard: the same exception on either path
    cusp::csr_matrix<int, double, cusp::host_memory> wide(3000, 3000, 3000 * 2);
    for (int i = 0; i <= 3000; i++) wide.row_offsets[i] = 2 * i;
    for (int i = 0; i < 3000; i++) {
        const int a = (i * 7) % 3000, b = (i * 13 + 5) % 3000;
        wide.column_indices[2 * i] = std::min(a, b); wide.column_indices[2 * i + 1] = std::max(a, b) + (a == b);
        if (wide.column_indices[2 * i + 1] >= 3000) wide.column_indices[2 * i + 1] = 2999, wide.column_indices[2 * i] = 2998;
        wide.values[2 * i] = 1; wide.values[2 * i + 1] = 2;
    }
    cusp::csr_matrix<int, double, Space> ws(wide);
    cusp::dia_matrix<int, double, Space> wd;
    ASSERT_THROWS(cusp::convert(ws, wd), cusp::format_conversion_exception);
}
DECLARE_SPACE_UNITTEST(TestCsrToDiaMatchesHostConversion);

// COO -> CSR in the matrix's own memory space: row-sorted entries take the device pass (cmi_coo_row_offsets: offsets from
// the row indices, order checked on the way), unsorted ones the general path, which sorts -- the same CSR either way
template <typename Space> void TestCooToCsrSortedAndUnsorted()
{
    cusp::csr_matrix<int, double, cusp::host_memory> h;
    cusp::gallery::poisson5pt(h, 31, 17);
    cusp::coo_matrix<int, double, cusp::host_memory> hc(h);
    { // sorted, with empty rows at both ends and in the middle: append 5 empty rows, blank two rows
        cusp::coo_matrix<int, double, cusp::host_memory> g(hc.num_rows + 5, hc.num_cols, 0);
        std::vector<int> ri, ci; std::vector<double> v;
        for (size_t k = 0; k < hc.num_entries; k++) {
            const int r = hc.row_indices[k];
            if (r == 0 || r == 100 || r == 101) continue;
            ri.push_back(r); ci.push_back(hc.column_indices[k]); v.push_back(hc.values[k]);
        }
        g.resize(hc.num_rows + 5, hc.num_cols, ri.size());
        for (size_t k = 0; k < ri.size(); k++) { g.row_indices[k] = ri[k]; g.column_indices[k] = ci[k]; g.values[k] = v[k]; }
        cusp::csr_matrix<int, double, cusp::host_memory> want(g);
        cusp::coo_matrix<int, double, Space> dg(g);
        cusp::csr_matrix<int, double, Space> got(dg);
        ASSERT_ARRAYS_EQUAL(got.row_offsets, want.row_offsets);
        ASSERT_ARRAYS_EQUAL(got.column_indices, want.column_indices);
        ASSERT_ARRAYS_EQUAL(got.values, want.values);
        ASSERT_EQUAL(int(want.row_offsets[1]), 0); ASSERT_EQUAL(int(want.row_offsets[102]) - int(want.row_offsets[100]), 0);
    }
    { // rows in reverse order (each row's entries in their own order): not sorted by row -> general path, a stable sort by row
        cusp::coo_matrix<int, double, cusp::host_memory> rev(hc.num_rows, hc.num_cols, hc.num_entries);
        size_t k = 0;
        for (size_t r = h.num_rows; r-- > 0;)
            for (int q = h.row_offsets[r]; q < h.row_offsets[r + 1]; q++, k++) {
                rev.row_indices[k] = int(r); rev.column_indices[k] = h.column_indices[q]; rev.values[k] = h.values[q];
            }
        cusp::coo_matrix<int, double, Space> dr(rev);
        cusp::csr_matrix<int, double, Space> got(dr);
        ASSERT_ARRAYS_EQUAL(got.row_offsets, h.row_offsets);
        ASSERT_ARRAYS_EQUAL(got.column_indices, h.column_indices);
        ASSERT_ARRAYS_EQUAL(got.values, h.values);
    }
}
DECLARE_SPACE_UNITTEST(TestCooToCsrSortedAndUnsorted);

// every ordered pair of formats converts inside one memory space (on the device: directly, or through a CSR matrix that stays
// in HBM) and the matrix survives: back in CSR it is the gallery matrix again
template <typename Space, typename Src> void convert_to_every_format(const Src &src, const cusp::csr_matrix<int, double, cusp::host_memory> &want)
{
    auto same = [&](const cusp::csr_matrix<int, double, Space> &got) {
        ASSERT_ARRAYS_EQUAL(got.row_offsets, want.row_offsets);
        ASSERT_ARRAYS_EQUAL(got.column_indices, want.column_indices);
        ASSERT_ARRAYS_EQUAL(got.values, want.values);
    };
    { cusp::coo_matrix<int, double, Space> d(src); cusp::csr_matrix<int, double, Space> c(d); same(c); }
    { cusp::ell_matrix<int, double, Space> d(src); cusp::csr_matrix<int, double, Space> c(d); same(c); }
    { cusp::hyb_matrix<int, double, Space> d(src); cusp::csr_matrix<int, double, Space> c(d); same(c); }
    { cusp::dia_matrix<int, double, Space> d(src); cusp::csr_matrix<int, double, Space> c(d); same(c); }
    { cusp::csr_matrix<int, double, Space> c(src); same(c); }
}
template <typename Space> void TestEveryFormatPairConverts()
{
    cusp::csr_matrix<int, double, cusp::host_memory> h;
    cusp::gallery::poisson5pt(h, 19, 13);
    cusp::csr_matrix<int, double, Space> a(h);
    convert_to_every_format<Space>(a, h);
    { cusp::coo_matrix<int, double, Space> s(a); convert_to_every_format<Space>(s, h); }
    { cusp::ell_matrix<int, double, Space> s(a); convert_to_every_format<Space>(s, h); }
    { cusp::hyb_matrix<int, double, Space> s(a); convert_to_every_format<Space>(s, h); }
    { cusp::dia_matrix<int, double, Space> s(a); convert_to_every_format<Space>(s, h); }
}
DECLARE_SPACE_UNITTEST(TestEveryFormatPairConverts);

// ------------------------------------------------------------------------------------------------
// testing/array1d.cu:7-193 (push_back, cross-space construction and assignment, std::vector interop,
// iterator-range construction, equality across spaces); Thrust vectors are not part of this layer.  The
// "other" space is host_memory <-> Space, so the host build (no GPU) stays on the host
template <typename Space> void TestArray1dBasics()
{
    cusp::array1d<int, Space> a(4);
    ASSERT_EQUAL(a.size(), size_t(4));
    for (int i = 0; i < 4; i++) a[i] = i;
    a.push_back(4);
    ASSERT_EQUAL(a.size(), size_t(5));
    for (int i = 0; i < 5; i++) ASSERT_EQUAL(int(a[i]), i);

    cusp::array1d<int, Space> b(2);
    b[0] = 0; b[1] = 1;
    cusp::array1d<int, cusp::host_memory> h(b);
    cusp::array1d<int, Space> d(b);
    ASSERT_EQUAL(h.size(), size_t(2)); ASSERT_EQUAL(int(h[1]), 1);
    ASSERT_EQUAL(d.size(), size_t(2)); ASSERT_EQUAL(int(d[0]), 0); ASSERT_EQUAL(int(d[1]), 1);
    const cusp::array1d<int, cusp::host_memory> ch(2, 10);
    const cusp::array1d<int, Space> cd(ch);
    ASSERT_EQUAL(cd.size(), size_t(2)); ASSERT_EQUAL(int(cd[0]), 10); ASSERT_EQUAL(int(cd[1]), 10);

    std::vector<int> v(2, 10);
    cusp::array1d<int, Space> fromv(v), assigned = v, ranged(v.begin(), v.end());
    for (auto *p : {&fromv, &assigned, &ranged}) { ASSERT_EQUAL(p->size(), size_t(2)); ASSERT_EQUAL(int((*p)[0]), 10); ASSERT_EQUAL(int((*p)[1]), 10); }

    cusp::array1d<int, cusp::host_memory> h2 = b;
    cusp::array1d<int, Space> d2 = b;
    ASSERT_EQUAL(int(h2[1]), 1); ASSERT_EQUAL(int(d2[1]), 1);
    b = ch;
    ASSERT_EQUAL(int(b[0]), 10); ASSERT_EQUAL(int(b[1]), 10);
    const cusp::array1d<int, Space> cd20(2, 20);
    b = cd20;
    ASSERT_EQUAL(b.size(), size_t(2)); ASSERT_EQUAL(int(b[0]), 20); ASSERT_EQUAL(int(b[1]), 20);

    cusp::array1d<int, Space> A(2);
    A[0] = 10; A[1] = 20;
    cusp::array1d<int, cusp::host_memory> eh(A);
    cusp::array1d<int, Space> ed(A);
    std::vector<int> ev = {10, 20};
    ASSERT_TRUE(A == eh); ASSERT_TRUE(A == ed); ASSERT_TRUE(A == ev);
    eh.push_back(30); ed.push_back(30); ev.push_back(30);
    ASSERT_TRUE(A != eh); ASSERT_TRUE(A != ed); ASSERT_TRUE(A != ev);
    // resize keeps the leading elements; views alias the storage
    A.resize(4, 7);
    ASSERT_EQUAL(int(A[1]), 20); ASSERT_EQUAL(int(A[3]), 7);
    typename cusp::array1d<int, Space>::view w(A);
    w[0] = -1;
    ASSERT_EQUAL(int(A[0]), -1); ASSERT_EQUAL(w.size(), size_t(4));
    typename cusp::array1d<int, Space>::view sub = A.subarray(1, 2);
    ASSERT_EQUAL(sub.size(), size_t(2)); ASSERT_EQUAL(int(sub[0]), 20); ASSERT_EQUAL(int(sub[1]), 7);
}
DECLARE_SPACE_UNITTEST(TestArray1dBasics);

// testing/array2d.cu:100-296: element layout of both orientations with trivial and padded pitch, mixed-orientation
// assignment, resize (pitch smaller than the leading dimension is an error), swap
template <typename Space> void TestArray2dLayouts()
{
    const float v[2][3] = {{10, 20, 30}, {40, 50, 60}};
    {
        cusp::array2d<float, Space, cusp::row_major> A(2, 3);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A(i, j) = v[i][j];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) ASSERT_EQUAL(float(A(i, j)), v[i][j]);
        const float e[6] = {10, 20, 30, 40, 50, 60};
        for (int k = 0; k < 6; k++) ASSERT_EQUAL(float(A.values[k]), e[k]);
        A.resize(2, 3, 4);
        cusp::blas::fill(A.values, 0.0f);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A(i, j) = v[i][j];
        const float p[8] = {10, 20, 30, 0, 40, 50, 60, 0};
        for (int k = 0; k < 8; k++) ASSERT_EQUAL(float(A.values[k]), p[k]);
    }
    {
        cusp::array2d<float, Space, cusp::column_major> A(2, 3);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A(i, j) = v[i][j];
        const float e[6] = {10, 40, 20, 50, 30, 60};
        for (int k = 0; k < 6; k++) ASSERT_EQUAL(float(A.values[k]), e[k]);
        A.resize(2, 3, 4);
        cusp::blas::fill(A.values, 0.0f);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) A(i, j) = v[i][j];
        const float p[12] = {10, 40, 0, 0, 20, 50, 0, 0, 30, 60, 0, 0};
        for (int k = 0; k < 12; k++) ASSERT_EQUAL(float(A.values[k]), p[k]);
    }
    {
        cusp::array2d<float, Space, cusp::row_major> R(2, 3);
        cusp::array2d<float, Space, cusp::column_major> C(2, 3);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) R(i, j) = v[i][j];
        C = R;
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) ASSERT_EQUAL(float(C(i, j)), v[i][j]);
        cusp::blas::fill(R.values, 0.0f);
        R = C;
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) ASSERT_EQUAL(float(R(i, j)), v[i][j]);
    }
    {
        cusp::array2d<float, Space> A;
        A.resize(3, 2);
        ASSERT_EQUAL(A.num_rows, size_t(3)); ASSERT_EQUAL(A.num_cols, size_t(2)); ASSERT_EQUAL(A.pitch, size_t(2));
        ASSERT_EQUAL(A.num_entries, size_t(6)); ASSERT_EQUAL(A.values.size(), size_t(6));
        A.resize(3, 2, 4);
        ASSERT_EQUAL(A.pitch, size_t(4)); ASSERT_EQUAL(A.num_entries, size_t(6)); ASSERT_EQUAL(A.values.size(), size_t(12));
        ASSERT_THROWS(A.resize(3, 2, 1), cusp::invalid_input_exception);
    }
    {
        cusp::array2d<float, Space> A(2, 2), B(3, 1);
        A(0, 0) = 10; A(0, 1) = 20; A(1, 0) = 30; A(1, 1) = 40;
        B(0, 0) = 50; B(1, 0) = 60; B(2, 0) = 70;
        cusp::array2d<float, Space> A_copy(A), B_copy(B);
        A.swap(B);
        ASSERT_EQUAL(A.num_rows, size_t(3)); ASSERT_EQUAL(A.num_cols, size_t(1)); ASSERT_ARRAYS_EQUAL(A.values, B_copy.values);
        ASSERT_EQUAL(B.num_rows, size_t(2)); ASSERT_EQUAL(B.num_cols, size_t(2)); ASSERT_ARRAYS_EQUAL(B.values, A_copy.values);
    }
}
DECLARE_SPACE_UNITTEST(TestArray2dLayouts);

// ------------------------------------------------------------------------------------------------
// containers: testing/{csr,coo,ell,dia,hyb}_matrix.cu -- BasicConstructor, CopyConstructor, Resize, Swap, Rebind
template <typename Space> void TestContainerShapes()
{
    { // csr_matrix.cu:4-16, 57-70
        cusp::csr_matrix<int, float, Space> m(3, 2, 6), r;
        r.resize(3, 2, 6);
        for (auto *p : {&m, &r}) {
            ASSERT_EQUAL(p->num_rows, size_t(3)); ASSERT_EQUAL(p->num_cols, size_t(2)); ASSERT_EQUAL(p->num_entries, size_t(6));
            ASSERT_EQUAL(p->row_offsets.size(), size_t(4)); ASSERT_EQUAL(p->column_indices.size(), size_t(6)); ASSERT_EQUAL(p->values.size(), size_t(6));
        }
    }
    { // coo_matrix.cu:4-16
        cusp::coo_matrix<int, float, Space> m(3, 2, 6);
        ASSERT_EQUAL(m.row_indices.size(), size_t(6)); ASSERT_EQUAL(m.column_indices.size(), size_t(6)); ASSERT_EQUAL(m.values.size(), size_t(6));
    }
    { // ell_matrix.cu:4-21, 99-117
        cusp::ell_matrix<int, float, Space> m(3, 2, 6, 2, 4), r;
        r.resize(3, 2, 6, 2, 4);
        for (auto *p : {&m, &r}) {
            ASSERT_EQUAL(p->num_rows, size_t(3)); ASSERT_EQUAL(p->num_cols, size_t(2)); ASSERT_EQUAL(p->num_entries, size_t(6));
            ASSERT_EQUAL(p->column_indices.num_cols, size_t(2)); ASSERT_EQUAL(p->column_indices.num_rows, size_t(3));
            ASSERT_EQUAL(p->column_indices.pitch, size_t(4)); ASSERT_EQUAL(p->column_indices.num_entries, size_t(6));
            ASSERT_EQUAL(p->values.num_cols, size_t(2)); ASSERT_EQUAL(p->values.num_rows, size_t(3)); ASSERT_EQUAL(p->values.pitch, size_t(4));
        }
    }
    { // dia_matrix.cu:4-17, 57-71
        cusp::dia_matrix<int, float, Space> m(4, 5, 7, 3, 8), r;
        r.resize(4, 5, 7, 3, 8);
        for (auto *p : {&m, &r}) {
            ASSERT_EQUAL(p->num_rows, size_t(4)); ASSERT_EQUAL(p->num_cols, size_t(5)); ASSERT_EQUAL(p->num_entries, size_t(7));
            ASSERT_EQUAL(p->diagonal_offsets.size(), size_t(3));
            ASSERT_EQUAL(p->values.num_rows, size_t(4)); ASSERT_EQUAL(p->values.num_cols, size_t(3)); ASSERT_EQUAL(p->values.pitch, size_t(8));
        }
    }
    { // hyb_matrix.cu:4-30, 92-119
        cusp::hyb_matrix<int, float, Space> m(10, 10, 42, 13, 5, 16), r;
        r.resize(10, 10, 42, 13, 5, 16);
        for (auto *p : {&m, &r}) {
            ASSERT_EQUAL(p->num_rows, size_t(10)); ASSERT_EQUAL(p->num_cols, size_t(10)); ASSERT_EQUAL(p->num_entries, size_t(55));
            ASSERT_EQUAL(p->ell.num_rows, size_t(10)); ASSERT_EQUAL(p->ell.num_entries, size_t(42));
            ASSERT_EQUAL(p->ell.column_indices.num_rows, size_t(10)); ASSERT_EQUAL(p->ell.column_indices.num_cols, size_t(5)); ASSERT_EQUAL(p->ell.column_indices.pitch, size_t(16));
            ASSERT_EQUAL(p->ell.values.num_cols, size_t(5)); ASSERT_EQUAL(p->ell.values.pitch, size_t(16));
            ASSERT_EQUAL(p->coo.num_rows, size_t(10)); ASSERT_EQUAL(p->coo.num_entries, size_t(13));
            ASSERT_EQUAL(p->coo.row_indices.size(), size_t(13)); ASSERT_EQUAL(p->coo.column_indices.size(), size_t(13)); ASSERT_EQUAL(p->coo.values.size(), size_t(13));
        }
    }
}
DECLARE_SPACE_UNITTEST(TestContainerShapes);

template <typename Space> void TestContainerCopySwapRebind()
{
    // csr_matrix.cu:18-54, 73-117 (the 3x2 and 1x2 / 3x1 examples)
    cusp::csr_matrix<int, float, Space> m(3, 2, 6);
    for (int i = 0; i < 4; i++) m.row_offsets[i] = 2 * i;
    for (int i = 0; i < 6; i++) { m.column_indices[i] = i % 2; m.values[i] = float(i); }
    cusp::csr_matrix<int, float, Space> c(m);
    ASSERT_EQUAL(c.num_rows, size_t(3)); ASSERT_EQUAL(c.num_cols, size_t(2)); ASSERT_EQUAL(c.num_entries, size_t(6));
    ASSERT_ARRAYS_EQUAL(c.row_offsets, m.row_offsets); ASSERT_ARRAYS_EQUAL(c.column_indices, m.column_indices); ASSERT_ARRAYS_EQUAL(c.values, m.values);

    cusp::csr_matrix<int, float, Space> A(1, 2, 2), B(3, 1, 3);
    A.row_offsets[0] = 0; A.row_offsets[1] = 2;
    A.column_indices[0] = 0; A.values[0] = 0; A.column_indices[1] = 1; A.values[1] = 1;
    for (int i = 0; i < 4; i++) B.row_offsets[i] = i;
    for (int i = 0; i < 3; i++) { B.column_indices[i] = 0; B.values[i] = float(i); }
    cusp::csr_matrix<int, float, Space> A_copy(A), B_copy(B);
    A.swap(B);
    ASSERT_EQUAL(A.num_rows, size_t(3)); ASSERT_EQUAL(A.num_cols, size_t(1)); ASSERT_EQUAL(A.num_entries, size_t(3));
    ASSERT_ARRAYS_EQUAL(A.row_offsets, B_copy.row_offsets); ASSERT_ARRAYS_EQUAL(A.column_indices, B_copy.column_indices); ASSERT_ARRAYS_EQUAL(A.values, B_copy.values);
    ASSERT_EQUAL(B.num_rows, size_t(1)); ASSERT_EQUAL(B.num_cols, size_t(2)); ASSERT_EQUAL(B.num_entries, size_t(2));
    ASSERT_ARRAYS_EQUAL(B.row_offsets, A_copy.row_offsets); ASSERT_ARRAYS_EQUAL(B.values, A_copy.values);

    // the other formats: copy through the same space, swap, and rebind to the OTHER memory space (…_matrix.cu Rebind)
    cusp::csr_matrix<int, float, cusp::host_memory> h;
    cusp::gallery::poisson5pt(h, 4, 3);
    cusp::coo_matrix<int, float, Space> coo(h), coo2; coo2.swap(coo);
    ASSERT_EQUAL(coo.num_entries, size_t(0)); ASSERT_EQUAL(coo2.num_entries, h.num_entries);
    cusp::ell_matrix<int, float, Space> ell(h), ell2(ell);
    ASSERT_ARRAYS_EQUAL(ell2.column_indices.values, ell.column_indices.values); ASSERT_ARRAYS_EQUAL(ell2.values.values, ell.values.values);
    cusp::dia_matrix<int, float, Space> dia(h), dia2; dia2.swap(dia);
    ASSERT_EQUAL(dia2.diagonal_offsets.size(), size_t(5)); ASSERT_EQUAL(dia.diagonal_offsets.size(), size_t(0));
    cusp::hyb_matrix<int, float, Space> hyb(h), hyb2(hyb);
    ASSERT_EQUAL(hyb2.num_entries, h.num_entries);
    typedef typename cusp::csr_matrix<int, float, cusp::host_memory>::template rebind<Space>::type Rebound;
    Rebound there(h);
    ASSERT_EQUAL(there.num_entries, h.num_entries);
    typedef typename cusp::ell_matrix<int, float, Space>::template rebind<cusp::host_memory>::type EllHost;
    EllHost back(ell2);
    cusp::csr_matrix<int, float, cusp::host_memory> h2(back);
    ASSERT_ARRAYS_EQUAL(h2.column_indices, h.column_indices); ASSERT_ARRAYS_EQUAL(h2.values, h.values);
}
DECLARE_SPACE_UNITTEST(TestContainerCopySwapRebind);

// testing/csr_matrix_view.cu:6-194 (views here are typed by element + memory space, not by Thrust iterator:
// `typename Matrix::view`; the tests compare the addresses the views alias)
template <typename Space> void TestCsrMatrixViews()
{
    typedef cusp::csr_matrix<int, float, Space> Matrix;
    typedef typename Matrix::view View;
    Matrix M(3, 2, 6);
    auto same = [&](const View &v) {
        ASSERT_EQUAL(v.num_rows, size_t(3)); ASSERT_EQUAL(v.num_cols, size_t(2)); ASSERT_EQUAL(v.num_entries, size_t(6));
        ASSERT_TRUE(v.row_offsets.data() == M.row_offsets.data() && v.row_offsets.size() == M.row_offsets.size());
        ASSERT_TRUE(v.column_indices.data() == M.column_indices.data() && v.column_indices.size() == M.column_indices.size());
        ASSERT_TRUE(v.values.data() == M.values.data() && v.values.size() == M.values.size());
    };
    View V(3, 2, 6, cusp::make_array1d_view(M.row_offsets), cusp::make_array1d_view(M.column_indices), cusp::make_array1d_view(M.values));
    same(V);
    View W(M); same(W);
    View A = M; same(A);             // :52-92 assignment from matrix, from view
    View B = A; same(B);
    same(cusp::make_csr_matrix_view(M));
    View X = cusp::make_csr_matrix_view(M);
    View Y = cusp::make_csr_matrix_view(X);
    Y.row_offsets[0] = 0; Y.column_indices[0] = 1; Y.values[0] = 2; // writes through the view reach the matrix
    same(Y);
    ASSERT_EQUAL(int(M.column_indices[0]), 1); ASSERT_EQUAL(float(M.values[0]), 2.0f);
    const Matrix C(3, 2, 6);
    ASSERT_EQUAL(cusp::make_csr_matrix_view(C).num_entries, size_t(6));
    ASSERT_TRUE(cusp::make_csr_matrix_view(C).values.data() == C.values.data());
    // a view multiplies like its matrix
    cusp::csr_matrix<int, float, Space> P;
    cusp::gallery::poisson5pt(P, 4, 4);
    cusp::array1d<float, Space> x(16, 1.0f), y1(16, 9.0f), y2(16, -9.0f);
    cusp::multiply(P, x, y1);
    typename cusp::csr_matrix<int, float, Space>::view PV(P);
    cusp::multiply(PV, x, y2);
    ASSERT_ARRAYS_EQUAL(y1, y2);
    typedef cusp::coo_matrix<int, float, Space> Coo;
    Coo K(P);
    typename Coo::view KV = cusp::make_coo_matrix_view(K);
    cusp::multiply(KV, x, y2);
    ASSERT_ARRAYS_EQUAL(y1, y2);
}
DECLARE_SPACE_UNITTEST(TestCsrMatrixViews);

// cusp::copy (same format, any memory spaces) and cusp::print (reference cusp/copy.h, cusp/print.h)
template <typename Space> void TestCopyAndPrint()
{
    cusp::csr_matrix<int, float, cusp::host_memory> h;
    cusp::gallery::poisson5pt(h, 3, 2);
    cusp::csr_matrix<int, float, Space> d;
    cusp::copy(h, d);
    cusp::csr_matrix<int, float, cusp::host_memory> back;
    cusp::copy(d, back);
    ASSERT_ARRAYS_EQUAL(back.row_offsets, h.row_offsets); ASSERT_ARRAYS_EQUAL(back.column_indices, h.column_indices); ASSERT_ARRAYS_EQUAL(back.values, h.values);
    cusp::array1d<float, Space> v(3, 2.5f), w;
    cusp::copy(v, w);
    ASSERT_ARRAYS_EQUAL(v, w);
    std::ostringstream os;
    cusp::print(d, os);                       // any sparse format prints as its COO triplets
    const std::string text = os.str();
    ASSERT_TRUE(text.rfind("sparse matrix <6, 6> with 20 entries\n", 0) == 0);
    ASSERT_TRUE(text.find("              0              0        (4)\n") != std::string::npos);
    std::ostringstream oa;
    cusp::print(v, oa);
    ASSERT_EQUAL(oa.str(), std::string("array1d <3>\n        (2.5)\n        (2.5)\n        (2.5)\n"));
}
DECLARE_SPACE_UNITTEST(TestCopyAndPrint);

// coo_matrix.cu:130-267: sort_by_row, sort_by_row_and_column, is_sorted_*
template <typename Space> void TestCooMatrixSorting()
{
    {
        cusp::coo_matrix<int, float, Space> A(5, 5, 4);
        const int r[4] = {3, 4, 1, 2}, c[4] = {1, 2, 3, 4};
        for (int i = 0; i < 4; i++) { A.row_indices[i] = r[i]; A.column_indices[i] = c[i]; A.values[i] = float(i + 1); }
        ASSERT_EQUAL(A.is_sorted_by_row(), false); ASSERT_EQUAL(A.is_sorted_by_row_and_column(), false);
        A.sort_by_row();
        const int er[4] = {1, 2, 3, 4}, ec[4] = {3, 4, 1, 2}; const float ev[4] = {3, 4, 1, 2};
        for (int i = 0; i < 4; i++) { ASSERT_EQUAL(int(A.row_indices[i]), er[i]); ASSERT_EQUAL(int(A.column_indices[i]), ec[i]); ASSERT_EQUAL(float(A.values[i]), ev[i]); }
        ASSERT_EQUAL(A.is_sorted_by_row(), true);
    }
    {
        cusp::coo_matrix<int, float, Space> A(5, 5, 7);
        const int r[7] = {3, 4, 1, 2, 1, 0, 2}, c[7] = {1, 2, 3, 2, 2, 3, 1};
        for (int i = 0; i < 7; i++) { A.row_indices[i] = r[i]; A.column_indices[i] = c[i]; A.values[i] = float(i + 1); }
        A.sort_by_row_and_column();
        const int er[7] = {0, 1, 1, 2, 2, 3, 4}, ec[7] = {3, 2, 3, 1, 2, 1, 2}; const float ev[7] = {6, 5, 3, 7, 4, 1, 2};
        for (int i = 0; i < 7; i++) { ASSERT_EQUAL(int(A.row_indices[i]), er[i]); ASSERT_EQUAL(int(A.column_indices[i]), ec[i]); ASSERT_EQUAL(float(A.values[i]), ev[i]); }
        ASSERT_EQUAL(A.is_sorted_by_row_and_column(), true);
    }
}
DECLARE_SPACE_UNITTEST(TestCooMatrixSorting);

// cusp/sort.h:231,302 (the free functions the container's methods call; testing/sort.cu covers the counting sorts, which are not on this path):
// stable by row -- entries of a row keep their order -- and by (row, column); with and without an execution policy
template <typename Space> void TestSortByRowFreeFunctions()
{
    const int r[9] = {2, 0, 2, 1, 0, 2, 1, 0, 2}, c[9] = {5, 1, 0, 3, 0, 4, 2, 1, 0};
    cusp::array1d<int, Space> rows(9), cols(9);
    cusp::array1d<double, Space> vals(9);
    for (int i = 0; i < 9; i++) { rows[i] = r[i]; cols[i] = c[i]; vals[i] = double(i); }
    cusp::sort_by_row(rows, cols, vals);
    const int er[9] = {0, 0, 0, 1, 1, 2, 2, 2, 2}, ec[9] = {1, 0, 1, 3, 2, 5, 0, 4, 0}; const double ev[9] = {1, 4, 7, 3, 6, 0, 2, 5, 8};
    for (int i = 0; i < 9; i++) { ASSERT_EQUAL(int(rows[i]), er[i]); ASSERT_EQUAL(int(cols[i]), ec[i]); ASSERT_EQUAL(double(vals[i]), ev[i]); }
    for (int i = 0; i < 9; i++) { rows[i] = r[i]; cols[i] = c[i]; vals[i] = double(i); }
    cusp::sort_by_row_and_column(cusp::hip::par, rows, cols, vals);
    const int fr[9] = {0, 0, 0, 1, 1, 2, 2, 2, 2}, fc[9] = {0, 1, 1, 2, 3, 0, 0, 4, 5}; const double fv[9] = {4, 1, 7, 6, 3, 2, 8, 5, 0};
    for (int i = 0; i < 9; i++) { ASSERT_EQUAL(int(rows[i]), fr[i]); ASSERT_EQUAL(int(cols[i]), fc[i]); ASSERT_EQUAL(double(vals[i]), fv[i]); }
    cusp::array1d<int, Space> short_cols(3);
    ASSERT_THROWS(cusp::sort_by_row(rows, short_cols, vals), cusp::invalid_input_exception);
}
DECLARE_SPACE_UNITTEST(TestSortByRowFreeFunctions);

// cusp/format_utils.h:83,133 (testing/format_utils.cu: TestOffsetsToIndices / TestIndicesToOffsets): row offsets <-> row indices, empty rows at both
// ends and inside; on device_memory through the C-ABI builders
template <typename Space> void TestOffsetsAndIndices()
{
    const int off[8] = {0, 0, 2, 2, 3, 6, 6, 6};
    cusp::array1d<int, Space> offsets(8), indices, back(8, -1);
    for (int i = 0; i < 8; i++) offsets[i] = off[i];
    cusp::offsets_to_indices(offsets, indices);
    ASSERT_EQUAL(indices.size(), size_t(6));
    const int want[6] = {1, 1, 3, 4, 4, 4};
    for (int i = 0; i < 6; i++) ASSERT_EQUAL(int(indices[i]), want[i]);
    cusp::indices_to_offsets(indices, back);
    for (int i = 0; i < 8; i++) ASSERT_EQUAL(int(back[i]), off[i]);
    ASSERT_EQUAL(cusp::compute_max_entries_per_row(offsets), size_t(3));
    // indices in any order: counted all the same (the reference's host path; device arrays fall back to it)
    cusp::array1d<int, Space> shuffled(6);
    const int sh[6] = {4, 1, 4, 3, 1, 4};
    for (int i = 0; i < 6; i++) shuffled[i] = sh[i];
    cusp::array1d<int, Space> again(8, -1);
    cusp::indices_to_offsets(shuffled, again);
    for (int i = 0; i < 8; i++) ASSERT_EQUAL(int(again[i]), off[i]);
}
DECLARE_SPACE_UNITTEST(TestOffsetsAndIndices);

// testing/verify.cu:13-304 (TestIsValidMatrix{Coo,Csr,Dia,Ell,Hyb,Array2d}, TestAssertIsValidMatrix): the same 3 x 3 matrix, the same
// corruptions, the same verdicts
template <typename Space> void TestIsValidMatrix()
{
    cusp::array2d<float, Space> D(3, 3, 0.0f);
    D(0, 1) = 1; D(1, 0) = 1; D(1, 2) = 1; D(2, 1) = 1;
    { cusp::coo_matrix<int, float, Space> M(D); ASSERT_EQUAL(cusp::is_valid_matrix(M), true); }
    { cusp::coo_matrix<int, float, Space> M(D); M.row_indices[0] = 1; M.row_indices[1] = 0; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::coo_matrix<int, float, Space> M(D); M.column_indices[2] = -1; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::coo_matrix<int, float, Space> M(D); M.column_indices[2] = 4; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::csr_matrix<int, float, Space> M(D); ASSERT_EQUAL(cusp::is_valid_matrix(M), true); }
    { cusp::csr_matrix<int, float, Space> M(D); M.row_offsets[1] = 1; M.row_offsets[2] = 5; M.row_offsets[3] = 4; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::csr_matrix<int, float, Space> M(D); M.column_indices[2] = -1; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::csr_matrix<int, float, Space> M(D); M.column_indices[2] = 4; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::dia_matrix<int, float, Space> M(D); ASSERT_EQUAL(cusp::is_valid_matrix(M), true); }
    { cusp::dia_matrix<int, float, Space> M(D); M.values.num_rows = 2; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::ell_matrix<int, float, Space> M(D); ASSERT_EQUAL(cusp::is_valid_matrix(M), true); }
    { cusp::ell_matrix<int, float, Space> M(D); M.values.num_cols = M.values.num_cols + 1; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::ell_matrix<int, float, Space> M(D); M.column_indices.num_rows = 2; M.values.num_rows = 2; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::ell_matrix<int, float, Space> M(D); M.column_indices(0, 0) = cusp::ell_matrix<int, float, Space>::invalid_index; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::ell_matrix<int, float, Space> M(D); M.column_indices(0, 0) = -2; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::ell_matrix<int, float, Space> M(D); M.column_indices(0, 0) = 3; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::hyb_matrix<int, float, Space> M(D); ASSERT_EQUAL(cusp::is_valid_matrix(M), true); }
    { cusp::hyb_matrix<int, float, Space> M(D); M.ell.num_rows = 4; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::hyb_matrix<int, float, Space> M(D); M.coo.num_rows = 4; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    { cusp::hyb_matrix<int, float, Space> M(D); M.num_entries = 5; ASSERT_EQUAL(cusp::is_valid_matrix(M), false); }
    ASSERT_EQUAL(cusp::is_valid_matrix(D), true);
    { cusp::array2d<float, Space> E(D); E.num_entries = 8; ASSERT_EQUAL(cusp::is_valid_matrix(E), false); }
    // assert_is_valid_matrix throws format_exception with the reason; assert_same_dimensions throws invalid_input_exception
    cusp::csr_matrix<int, float, Space> bad(D);
    bad.row_offsets[3] = 9;
    ASSERT_THROWS(cusp::assert_is_valid_matrix(bad), cusp::format_exception);
    std::ostringstream why;
    ASSERT_EQUAL(cusp::is_valid_matrix(bad, why), false);
    ASSERT_TRUE(why.str().find("row_offsets ends at 9") != std::string::npos);
    cusp::csr_matrix<int, float, Space> good(D);
    cusp::assert_is_valid_matrix(good);
    cusp::array1d<float, Space> a3(3), b3(3), c4(4);
    cusp::assert_same_dimensions(a3, b3);
    ASSERT_THROWS(cusp::assert_same_dimensions(a3, b3, c4), cusp::invalid_input_exception);
}
DECLARE_SPACE_UNITTEST(TestIsValidMatrix);

// testing/monitor.cu:5-68, statement by statement
template <typename Space> void TestMonitorSimple()
{
    cusp::array1d<float, Space> b(2), r(2);
    b[0] = 10; b[1] = 0; r[0] = 10; r[1] = 0;
    cusp::monitor<float> monitor(b, 5, 0.5, 1.0);
    ASSERT_EQUAL(monitor.finished(r), false);
    ASSERT_EQUAL(monitor.iteration_count(), size_t(0)); ASSERT_EQUAL(monitor.iteration_limit(), size_t(5));
    ASSERT_EQUAL(monitor.relative_tolerance(), 0.5f); ASSERT_EQUAL(monitor.absolute_tolerance(), 1.0f); ASSERT_EQUAL(monitor.tolerance(), 6.0f);
    ++monitor;
    ASSERT_EQUAL(monitor.finished(r), false); ASSERT_EQUAL(monitor.iteration_count(), size_t(1)); ASSERT_EQUAL(monitor.residual_norm(), 10.0f);
    r[0] = 2;
    ASSERT_EQUAL(monitor.finished(r), true); ASSERT_EQUAL(monitor.iteration_count(), size_t(1)); ASSERT_EQUAL(monitor.residual_norm(), 2.0f);
    r[0] = 7;
    ASSERT_EQUAL(monitor.finished(r), false); ASSERT_EQUAL(monitor.iteration_count(), size_t(1)); ASSERT_EQUAL(monitor.residual_norm(), 7.0f);
    ++monitor;
    ASSERT_EQUAL(monitor.finished(r), false); ASSERT_EQUAL(monitor.iteration_count(), size_t(2)); ASSERT_EQUAL(monitor.residual_norm(), 7.0f);
    ++monitor; ++monitor;
    ASSERT_EQUAL(monitor.finished(r), false); ASSERT_EQUAL(monitor.iteration_count(), size_t(4)); ASSERT_EQUAL(monitor.residual_norm(), 7.0f);
    ++monitor;
    ASSERT_EQUAL(monitor.finished(r), true); ASSERT_EQUAL(monitor.iteration_count(), size_t(5)); ASSERT_EQUAL(monitor.residual_norm(), 7.0f);
    monitor.reset(r);
    ASSERT_EQUAL(monitor.finished(r), false); ASSERT_EQUAL(monitor.iteration_count(), size_t(0)); ASSERT_EQUAL(monitor.residual_norm(), 7.0f);
}
DECLARE_SPACE_UNITTEST(TestMonitorSimple);

// the BLAS-1 routines cg calls, with the vectors and answers of testing/blas.cu:56-140 (axpy, axpby), 252-276
// (copy), 278-303 (dot; dotc on real data), 335-357 (fill), 407-425 (nrm2), containers and views, size checks
template <typename Space> void TestBlasKnownAnswers()
{
    typedef cusp::array1d<float, Space> Array;
    typedef typename Array::view View;
    const float xs[4] = {7.0f, 5.0f, 4.0f, -3.0f}, ys[4] = {0.0f, -2.0f, 0.0f, 5.0f};
    Array x(4), y(4), z(4, 0), w(3);
    for (int i = 0; i < 4; i++) { x[i] = xs[i]; y[i] = ys[i]; }
    cusp::blas::axpy(x, y, 2.0f);
    { const float e[4] = {14, 8, 8, -1}; for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(y[i]), e[i]); }
    { View vx(x), vy(y); cusp::blas::axpy(vx, vy, 2.0f); }
    { const float e[4] = {28, 18, 16, -7}; for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(y[i]), e[i]); }
    ASSERT_THROWS(cusp::blas::axpy(x, w, 1.0f), cusp::invalid_input_exception);

    for (int i = 0; i < 4; i++) y[i] = ys[i];
    cusp::blas::axpby(x, y, z, 2.0f, 1.0f);
    { const float e[4] = {14, 8, 8, -1}; for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(z[i]), e[i]); }
    cusp::blas::fill(z, 0.0f);
    { View vx(x), vy(y), vz(z); cusp::blas::axpby(vx, vy, vz, 2.0f, 1.0f); }
    { const float e[4] = {14, 8, 8, -1}; for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(z[i]), e[i]); }
    ASSERT_THROWS(cusp::blas::axpby(x, y, w, 2.0f, 1.0f), cusp::invalid_input_exception);

    { Array c(4, -1); cusp::blas::copy(x, c); ASSERT_ARRAYS_EQUAL(x, c); }
    { Array c(4, -1); View vx(x), vc(c); cusp::blas::copy(vx, vc); ASSERT_ARRAYS_EQUAL(x, c); }
    ASSERT_THROWS(cusp::blas::copy(w, x), cusp::invalid_input_exception);

    const float x6[6] = {7, 5, 4, -3, 0, 4}, y6[6] = {0, -2, 0, 5, 6, 1};
    Array a(6), b(6);
    for (int i = 0; i < 6; i++) { a[i] = x6[i]; b[i] = y6[i]; }
    ASSERT_EQUAL(cusp::blas::dot(a, b), -21.0f);
    ASSERT_EQUAL(cusp::blas::dot(View(a), View(b)), -21.0f);
    ASSERT_EQUAL(cusp::blas::dotc(a, b), -21.0f);
    ASSERT_THROWS(cusp::blas::dot(a, w), cusp::invalid_input_exception);

    cusp::blas::fill(x, 2.0f);
    for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(x[i]), 2.0f);
    { View vx(x); cusp::blas::fill(vx, 1.0f); }
    for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(x[i]), 1.0f);

    a[5] = 1.0f; // 7 5 4 -3 0 1
    ASSERT_EQUAL(cusp::blas::nrm2(a), 10.0f);
    ASSERT_EQUAL(cusp::blas::nrm2(View(a)), 10.0f);
}
DECLARE_SPACE_UNITTEST(TestBlasKnownAnswers);

// testing/ell_matrix.cu:5-22
template <typename Space> void TestEllMatrixBasicConstructor()
{
    cusp::ell_matrix<int, float, Space> matrix(3, 2, 6, 2, 4);
    ASSERT_EQUAL(matrix.num_rows, size_t(3)); ASSERT_EQUAL(matrix.num_cols, size_t(2)); ASSERT_EQUAL(matrix.num_entries, size_t(6));
    ASSERT_EQUAL(matrix.column_indices.num_cols, size_t(2)); ASSERT_EQUAL(matrix.column_indices.num_rows, size_t(3));
    ASSERT_EQUAL(matrix.column_indices.pitch, size_t(4)); ASSERT_EQUAL(matrix.column_indices.num_entries, size_t(6));
    ASSERT_EQUAL(matrix.values.pitch, size_t(4)); ASSERT_EQUAL(matrix.values.num_entries, size_t(6));
    cusp::ell_matrix<int, double, Space> big(9998244, 9998244, 0, 0);
    ASSERT_EQUAL(big.column_indices.pitch, size_t(9998272)); // pitch of the 3162^2 ELL
    typedef cusp::array2d<float, Space, cusp::column_major> A2;
    A2 bad;
    ASSERT_THROWS(bad.resize(10, 2, 8), cusp::invalid_input_exception); // pitch < minor dimension
}
DECLARE_SPACE_UNITTEST(TestEllMatrixBasicConstructor);

// testing/poisson.cu:6-25
template <typename Space> void TestPoisson5pt()
{
    cusp::dia_matrix<int, float, Space> matrix;
    cusp::gallery::poisson5pt(matrix, 2, 3);
    ASSERT_EQUAL(matrix.num_entries, size_t(20));
    cusp::array2d<float, cusp::host_memory> R(matrix);
    const float E[6][6] = {{4, -1, -1, 0, 0, 0}, {-1, 4, 0, -1, 0, 0}, {-1, 0, 4, -1, -1, 0}, {0, -1, -1, 4, 0, -1}, {0, 0, -1, 0, 4, -1}, {0, 0, 0, -1, -1, 4}};
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) ASSERT_EQUAL(R(i, j), E[i][j]);
    // HYB heuristic on 5-pt Poisson: K = 5, COO part empty
    cusp::csr_matrix<int, double, cusp::host_memory> big;
    cusp::gallery::poisson5pt(big, 100, 100);
    ASSERT_EQUAL(cusp::compute_optimal_entries_per_row(big.row_offsets), size_t(5));
    cusp::hyb_matrix<int, double, Space> hyb(big);
    ASSERT_EQUAL(hyb.ell.column_indices.num_cols, size_t(5)); ASSERT_EQUAL(hyb.coo.num_entries, size_t(0));
}
DECLARE_SPACE_UNITTEST(TestPoisson5pt);

// ------------------------------------------------------------------------------------------------
// testing/cg.cu:46-99 + docs/quickstart.md:72-87
template <typename Space> void TestConjugateGradient()
{
    cusp::csr_matrix<int, double, Space> A;
    cusp::gallery::poisson5pt(A, 10, 10);
    cusp::array1d<double, Space> x(A.num_rows, 0), b(A.num_rows, 1);
    cusp::monitor<double> monitor(b, 20, 1e-4);
    cusp::krylov::cg(A, x, b, monitor);
    cusp::array1d<double, Space> residual(A.num_rows, 0);
    cusp::multiply(A, x, residual);
    cusp::blas::axpby(residual, b, residual, -1.0, 1.0);
    ASSERT_TRUE(monitor.converged());
    ASSERT_TRUE(monitor.residual_norm() < 1e-4 * cusp::blas::nrm2(b));
    ASSERT_TRUE(cusp::blas::nrm2(residual) < 1e-4 * cusp::blas::nrm2(b));
}
DECLARE_SPACE_UNITTEST(TestConjugateGradient);

// testing/cg.cu:46-67 verbatim protocol: float, monitor(b, 20, 1e-4)
template <typename Space> void TestConjugateGradientFloat()
{
    cusp::csr_matrix<int, float, Space> A;
    cusp::gallery::poisson5pt(A, 10, 10);
    cusp::array1d<float, Space> x(A.num_rows, 0.0f), b(A.num_rows, 1.0f);
    cusp::monitor<float> monitor(b, 20, 1e-4);
    cusp::krylov::cg(A, x, b, monitor);
    cusp::array1d<float, Space> residual(A.num_rows, 0.0f);
    cusp::multiply(A, x, residual);
    cusp::blas::axpby(residual, b, residual, -1.0f, 1.0f);
    ASSERT_EQUAL(monitor.converged(), true);
    ASSERT_TRUE(monitor.residual_norm() < 1e-4 * cusp::blas::nrm2(b));
    ASSERT_TRUE(cusp::blas::nrm2(residual) < 1e-4 * cusp::blas::nrm2(b));
}
DECLARE_SPACE_UNITTEST(TestConjugateGradientFloat);

template <typename Space> void TestConjugateGradientQuickstartTrace()
{
    // docs/quickstart.md:72-87 (examples/Solvers/cg.cu): poisson5pt(10,10), b = 1, x0 = 0, rel-tol 1e-3:
    // converged after 12 iterations with this residual history (printed to 7 significant digits)
    const double trace[13] = {1.0e+01, 1.414214e+01, 1.093707e+01, 8.949319e+00, 6.190055e+00, 3.835189e+00, 1.745481e+00,
                              5.963546e-01, 2.371134e-01, 1.152524e-01, 3.134467e-02, 1.144415e-02, 1.824176e-03};
    cusp::csr_matrix<int, double, Space> A;
    cusp::gallery::poisson5pt(A, 10, 10);
    cusp::array1d<double, Space> x(A.num_rows, 0), b(A.num_rows, 1);
    cusp::monitor<double> monitor(b, 100, 1e-3);
    cusp::krylov::cg(A, x, b, monitor);
    ASSERT_TRUE(monitor.converged());
    ASSERT_EQUAL(monitor.iteration_count(), size_t(12));
    ASSERT_EQUAL(monitor.residuals.size(), size_t(13));
    for (int i = 0; i < 13; i++) ASSERT_TRUE(std::fabs(monitor.residuals[i] - trace[i]) <= 2e-6 * trace[i]);
}
DECLARE_SPACE_UNITTEST(TestConjugateGradientQuickstartTrace);

template <typename Space> void TestConjugateGradientZeroResidual()
{
    cusp::csr_matrix<int, double, Space> A;
    cusp::gallery::poisson5pt(A, 10, 10);
    cusp::array1d<double, Space> x(A.num_rows, 1), b(A.num_rows);
    cusp::multiply(A, x, b);
    cusp::monitor<double> monitor(b, 20, 0);
    cusp::krylov::cg(A, x, b, monitor);
    ASSERT_EQUAL(monitor.converged(), true);
    ASSERT_EQUAL(monitor.iteration_count(), size_t(0));
}
DECLARE_SPACE_UNITTEST(TestConjugateGradientZeroResidual);

// ------------------------------------------------------------------------------------------------
// testing/multiply.cu:792-858 with testing/unittest/special_types.h:108-141
struct my_system : cusp::execution_policy<my_system> {
    explicit my_system(int) : correctly_dispatched(false), num_copies(0) {}
    my_system(const my_system &o) : cusp::execution_policy<my_system>(o), correctly_dispatched(false), num_copies(o.num_copies + 1) {}
    void validate_dispatch() { correctly_dispatched = (num_copies == 0); }
    bool is_valid() { return correctly_dispatched; }
    bool correctly_dispatched;
    unsigned num_copies;
};
template <typename MatrixType, typename ArrayType1, typename ArrayType2>
void multiply(my_system &system, const MatrixType &, const ArrayType1 &, ArrayType2 &) { system.validate_dispatch(); }

struct plain_system : cusp::execution_policy<plain_system> {}; // no overload of its own: default path

template <typename Space> void TestMultiplyDispatch()
{
    cusp::csr_matrix<int, float, Space> A;
    cusp::gallery::poisson5pt(A, 3, 3);
    cusp::array1d<float, Space> x(9, 1), y(9, 10);
    my_system sys(0);
    cusp::multiply(sys, A, x, y); // must reach ::multiply(my_system&, ...) by ADL, policy not copied
    ASSERT_EQUAL(true, sys.is_valid());
    ASSERT_EQUAL(float(y[0]), 10.0f); // the user overload did nothing
    plain_system plain;
    cusp::multiply(plain, A, x, y);
    ASSERT_EQUAL(float(y[0]), 2.0f); ASSERT_EQUAL(float(y[4]), 0.0f);
}
DECLARE_SPACE_UNITTEST(TestMultiplyDispatch);

// cusp::omp::par (reference cusp/system/omp): the OpenMP CSR loop gives the sequential multiply's bits, every format works
void TestMultiplyOmpPolicy()
{
    cusp::csr_matrix<int, double, cusp::host_memory> A;
    cusp::gallery::poisson5pt(A, 57, 43);
    const size_t N = A.num_rows;
    cusp::array1d<double, cusp::host_memory> x(N), y0(N, 10.0), y1(N, -3.0);
    for (size_t i = 0; i < N; i++) x[i] = double((i * 2654435761u) % 1000u) / 997.0 - 0.5;
    cusp::multiply(A, x, y0);
    cusp::multiply(cusp::omp::par, A, x, y1);
    for (size_t i = 0; i < N; i++) ASSERT_EQUAL(y0[i], y1[i]);
    cusp::ell_matrix<int, double, cusp::host_memory> E(A);
    cusp::coo_matrix<int, double, cusp::host_memory> C(A);
    cusp::array1d<double, cusp::host_memory> y2(N, 7.0), y3(N, 7.0);
    cusp::multiply(cusp::omp::par, E, x, y2);
    cusp::multiply(cusp::omp::par, C, x, y3);
    for (size_t i = 0; i < N; i++) { ASSERT_EQUAL(y0[i], y2[i]); ASSERT_EQUAL(y0[i], y3[i]); }
    // y += A x through the policy overload with functors
    cusp::array1d<double, cusp::host_memory> z(N, 1.5);
    cusp::multiply(cusp::omp::par, A, x, z, cusp::identity_function<double>(), cusp::multiplies<double>(), cusp::plus<double>());
    for (size_t i = 0; i < N; i += 97) {
        double acc = 1.5;
        for (int jj = A.row_offsets[i]; jj < A.row_offsets[i + 1]; jj++) acc = acc + A.values[jj] * x[A.column_indices[jj]];
        ASSERT_EQUAL(z[i], acc);
    }
}
DECLARE_UNITTEST(TestMultiplyOmpPolicy);

// Per-matrix plan options of the device containers: the 16-bit column copy (csr_matrix::compress_indices) and HYB's
// one-launch plan give the host loops' bits; both are re-made when the structure changes.
template <typename Space> void TestDevicePlanOptions()
{
    if constexpr (std::is_same<Space, cusp::device_memory>::value) {
        cusp::csr_matrix<int, double, cusp::host_memory> Ah;
        cusp::gallery::poisson5pt(Ah, 97, 61);
        const size_t N = Ah.num_rows;
        cusp::array1d<double, cusp::host_memory> xh(N), want(N, 10.0);
        for (size_t i = 0; i < N; i++) xh[i] = double((i * 2654435761u) % 1000u) / 997.0 - 0.5;
        cusp::multiply(Ah, xh, want);
        cusp::csr_matrix<int, double, Space> A(Ah);
        cusp::array1d<double, Space> x(xh), y(N, 10.0);
        A.compress_indices(true);
        cmi_config c;
        ASSERT_EQUAL(cmi_plan_config(A.plan(), &c), 0);
        ASSERT_EQUAL(c.kernel, (int)CMI_CSR_STREAM_C16); // a 5-point stencil: every tile spans a few hundred columns
        cusp::multiply(A, x, y);
        cusp::array1d<double, cusp::host_memory> got(y);
        for (size_t i = 0; i < N; i++) ASSERT_EQUAL(got[i], want[i]);
        // the solver runs on the compressed matrix too (fused SpMV + dot through the same plan)
        cusp::array1d<double, Space> sol(N, 0.0), b(N, 1.0);
        cusp::monitor<double> mon(b, 400, 1e-10);
        cusp::krylov::cg(A, sol, b, mon);
        ASSERT_EQUAL(mon.converged(), true);
        A.compress_indices(false);
        ASSERT_EQUAL(cmi_plan_config(A.plan(), &c), 0);
        ASSERT_EQUAL(c.kernel == (int)CMI_CSR_STREAM_C16, false);
        // HYB: one launch through the container's plan
        cusp::hyb_matrix<int, double, cusp::host_memory> Hh;
        cusp::convert(Ah, Hh);
        cusp::hyb_matrix<int, double, Space> H(Ah);
        cusp::array1d<double, cusp::host_memory> wh(N, 10.0);
        cusp::multiply(Hh, xh, wh);
        cusp::array1d<double, Space> yh(N, 10.0);
        cusp::multiply(H, x, yh);
        cusp::array1d<double, cusp::host_memory> goth(yh);
        for (size_t i = 0; i < N; i++) ASSERT_EQUAL(goth[i], wh[i]);
        if (H.coo.num_entries > 0) {
            int sorted = -1, exact = -1;
            ASSERT_EQUAL(cmi_plan_info(H.plan(), nullptr, nullptr, &sorted, &exact), 0);
            ASSERT_EQUAL(sorted, 1);
            ASSERT_EQUAL(exact, 1);
        }
    }
}
DECLARE_SPACE_UNITTEST(TestDevicePlanOptions);

// Round 4: a matrix whose columns come in runs of three (three degrees of freedom per node, 11.5 M entries): the device containers make their
// plan WITH the column indices, and that plan multiplies from the run-compressed column copy (CMI_CSR_STREAM_WAVER) -- the host loop's bits
// (reference arithmetic: cusp/system/detail/sequential/multiply/csr_spmv.h:42-74), for csr_matrix and for the same matrix held in coo_matrix;
// values refreshed in place are seen by the next multiply (nothing of them is cached); CG runs its fused SpMV + dot through the same plan.
template <typename Space> void TestRunCompressedPlanOfFemBlocks()
{
    if constexpr (std::is_same<Space, cusp::device_memory>::value) {
        const size_t nodes = 160000, N = 3 * nodes;
        cusp::csr_matrix<int, double, cusp::host_memory> Ah(N, N, 0);
        std::vector<int> cols;
        std::vector<double> vals;
        const long nb[8] = {-401, -400, -399, -1, 0, 1, 400, 401}; // a node couples to itself and seven neighbours of a 400-wide mesh
        for (size_t node = 0; node < nodes; node++)
            for (int d = 0; d < 3; d++) {
                Ah.row_offsets[3 * node + d] = (int)cols.size();
                for (long o : nb) {
                    const long other = (long)node + o;
                    if (other < 0 || other >= (long)nodes) continue;
                    for (int e = 0; e < 3; e++) {
                        cols.push_back((int)(3 * other + e));
                        vals.push_back(o == 0 && e == d ? 30.0 + double(node % 7) : -0.25 - double(((3 * node + d) + (size_t)(3 * other + e)) % 5) / 16.0); // symmetric in (row, column)
                    }
                }
            }
        Ah.row_offsets[N] = (int)cols.size();
        Ah.resize(N, N, cols.size());
        for (size_t k = 0; k < cols.size(); k++) { Ah.column_indices[k] = cols[k]; Ah.values[k] = vals[k]; }
        ASSERT_EQUAL(Ah.num_entries > 10000000, true);
        cusp::array1d<double, cusp::host_memory> xh(N), want(N, 10.0);
        for (size_t i = 0; i < N; i++) xh[i] = double((i * 2654435761u) % 1000u) / 997.0 - 0.5;
        cusp::multiply(Ah, xh, want);
        cusp::csr_matrix<int, double, Space> A(Ah);
        cusp::array1d<double, Space> x(xh), y(N, 10.0);
        cmi_config c;
        ASSERT_EQUAL(cmi_plan_config(A.plan(), &c), 0);
        ASSERT_EQUAL(c.kernel, (int)CMI_CSR_STREAM_WAVER);
        int64_t owned = 0;
        ASSERT_EQUAL(cmi_plan_device_bytes(A.plan(), &owned), 0);
        ASSERT_EQUAL(owned > (int64_t)(Ah.num_entries / 3) * 4 && owned < (int64_t)Ah.num_entries * 3, true); // ~ 4 bytes per piece of three
        cusp::multiply(A, x, y);
        cusp::array1d<double, cusp::host_memory> got(y);
        for (size_t i = 0; i < N; i++) ASSERT_EQUAL(got[i], want[i]);
        int valid = 0;
        ASSERT_EQUAL(cmi_plan_validate(A.plan(), A.row_offsets.data(), A.column_indices.data(), nullptr, &valid), 0);
        ASSERT_EQUAL(valid, 1);
        // values refreshed IN PLACE: the same plan, the new values' result
        cusp::blas::scal(A.values, 0.5);
        for (size_t k = 0; k < Ah.num_entries; k++) Ah.values[k] *= 0.5;
        cusp::multiply(Ah, xh, want);
        cusp::multiply(A, x, y);
        got = y;
        for (size_t i = 0; i < N; i++) ASSERT_EQUAL(got[i], want[i]);
        // the same matrix in COO: its plan's CSR sub-plan is made with the columns too
        cusp::coo_matrix<int, double, Space> C(Ah);
        ASSERT_EQUAL(cmi_plan_config(C.plan(), &c), 0);
        ASSERT_EQUAL(c.kernel, (int)CMI_CSR_STREAM_WAVER);
        cusp::array1d<double, Space> yc(N, 10.0);
        cusp::multiply(C, x, yc);
        got = yc;
        for (size_t i = 0; i < N; i++) ASSERT_EQUAL(got[i], want[i]);
        // the fused CG iteration through the same plan (diagonally dominant: converges)
        cusp::array1d<double, Space> sol(N, 0.0), b(N, 1.0);
        cusp::monitor<double> mon(b, 200, 1e-10);
        cusp::krylov::cg(A, sol, b, mon);
        ASSERT_EQUAL(mon.converged(), true);
    }
}
DECLARE_SPACE_UNITTEST(TestRunCompressedPlanOfFemBlocks);

// testing/cg.cu:11-44: cg(policy, ...) reaches a user overload by ADL; a policy without one solves
template <class LinearOperator, class VectorType1, class VectorType2, class Monitor, class Preconditioner>
void cg(my_system &system, const LinearOperator &, VectorType1 &, const VectorType2 &, Monitor &, Preconditioner &) { system.validate_dispatch(); }

template <typename Space> void TestConjugateGradientDispatch()
{
    cusp::csr_matrix<int, float, Space> A;
    cusp::gallery::poisson5pt(A, 10, 10);
    cusp::array1d<float, Space> x(A.num_rows, 0.0f);
    cusp::monitor<float> monitor(x, 20, 1e-4);
    cusp::identity_operator<float, Space> M(A.num_rows, A.num_cols);
    my_system sys(0);
    cusp::krylov::cg(sys, A, x, x, monitor, M);
    ASSERT_EQUAL(true, sys.is_valid());
    plain_system plain;
    cusp::array1d<float, Space> b(A.num_rows, 1.0f);
    cusp::monitor<float> m2(b, 100, 1e-4);
    cusp::krylov::cg(plain, A, x, b, m2);
    ASSERT_EQUAL(m2.converged(), true);
    cusp::krylov::cg(cusp::hip::par, A, x, b);
}
DECLARE_SPACE_UNITTEST(TestConjugateGradientDispatch);

// testing/blas.cu:752-1208 for the routines cg uses: blas::f(policy, ...) reaches the user's overload by ADL
template <typename A1, typename A2, typename S> void axpy(my_system &s, const A1 &, A2 &, const S) { s.validate_dispatch(); }
template <typename A1, typename A2, typename A3, typename S1, typename S2> void axpby(my_system &s, const A1 &, const A2 &, A3 &, S1, S2) { s.validate_dispatch(); }
template <typename A1, typename A2> void copy(my_system &s, const A1 &, A2 &) { s.validate_dispatch(); }
template <typename A1, typename S> void fill(my_system &s, A1 &, const S) { s.validate_dispatch(); }
template <typename A1, typename A2> typename A1::value_type dot(my_system &s, const A1 &, const A2 &) { s.validate_dispatch(); return 0; }
template <typename A1, typename A2> typename A1::value_type dotc(my_system &s, const A1 &, const A2 &) { s.validate_dispatch(); return 0; }
template <typename A1> typename A1::value_type nrm2(my_system &s, const A1 &) { s.validate_dispatch(); return 0; }

template <typename Space> void TestBlasDispatch()
{
    cusp::array1d<float, Space> x(4, 1.0f), y(4, 2.0f), z(4, 0.0f);
    { my_system sys(0); cusp::blas::axpy(sys, x, y, 2.0f); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::axpby(sys, x, y, z, 2.0f, 1.0f); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::copy(sys, x, y); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::fill(sys, x, 3.0f); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::dot(sys, x, y); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::dotc(sys, x, y); ASSERT_EQUAL(true, sys.is_valid()); }
    { my_system sys(0); cusp::blas::nrm2(sys, x); ASSERT_EQUAL(true, sys.is_valid()); }
    ASSERT_EQUAL(float(y[0]), 2.0f); // the user overloads did nothing
    plain_system plain;              // a policy without overloads: the real routines
    cusp::blas::axpy(plain, x, y, 2.0f);
    ASSERT_EQUAL(float(y[3]), 4.0f);
    ASSERT_EQUAL(cusp::blas::dot(cusp::hip::par, x, y), 16.0f);
    ASSERT_EQUAL(cusp::blas::nrm2(plain, x), 2.0f);
}
DECLARE_SPACE_UNITTEST(TestBlasDispatch);

// ------------------------------------------------------------------------------------------------
// error behaviour of the boundary
template <typename Space> void TestMultiplyErrors()
{
    cusp::csr_matrix<int, double, Space> A;
    cusp::gallery::poisson5pt(A, 4, 4);
    cusp::array1d<double, Space> x(15), y(16);
    ASSERT_THROWS(cusp::multiply(A, x, y), cusp::invalid_input_exception);
    cusp::array1d<double, Space> x2(16, 1.0);
    cusp::multiply(A, x2, y);
    ASSERT_EQUAL(double(y[5]), 0.0);
    cusp::csr_matrix<int, double, Space> empty(3, 3, 0);
    cusp::array1d<double, Space> z(3, 7.0), w(3, 1.0);
    cusp::array1d<int, cusp::host_memory> zeros(4, 0);
    empty.row_offsets = zeros;
    cusp::multiply(empty, w, z);
    ASSERT_EQUAL(double(z[1]), 0.0);
}
DECLARE_SPACE_UNITTEST(TestMultiplyErrors);

// testing/blas.cu:9-28 (amax), 147-199 (axpbypcz), 203-249 (xmy), 389-408 (nrm1), 479-498 (nrmmax), 524-560 (scal): the vectors and answers of
// the reference's tests, containers and views, size checking
template <typename Space> void TestBlasRestOfTheSet()
{
    typedef cusp::array1d<float, Space> Array;
    typedef typename Array::view View;
    {
        Array x(6); const float xv[6] = {0, -5, 4, -3, 7, 1};
        for (int i = 0; i < 6; i++) x[i] = xv[i];
        View vx(x);
        ASSERT_EQUAL(cusp::blas::amax(x), 4); ASSERT_EQUAL(cusp::blas::amax(vx), 4);
        ASSERT_EQUAL(cusp::blas::nrmmax(x), 7.0f); ASSERT_EQUAL(cusp::blas::nrmmax(vx), 7.0f);
    }
    {
        Array x(6); const float xv[6] = {7, 5, 4, -3, 0, 1};
        for (int i = 0; i < 6; i++) x[i] = xv[i];
        View vx(x);
        ASSERT_EQUAL(cusp::blas::nrm1(x), 20.0f); ASSERT_EQUAL(cusp::blas::nrm1(vx), 20.0f);
    }
    {
        Array x(4), y(4), z(4), w(4, 0);
        const float xv[4] = {7, 5, 4, -3}, yv[4] = {0, -2, 0, 5}, zv[4] = {1, 0, 3, -2};
        for (int i = 0; i < 4; i++) { x[i] = xv[i]; y[i] = yv[i]; z[i] = zv[i]; }
        cusp::blas::axpbypcz(x, y, z, w, 2.0f, 1.0f, 3.0f);
        const float want[4] = {17, 8, 17, -7};
        for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(w[i]), want[i]);
        cusp::blas::fill(w, 0.0f);
        View vx(x), vy(y), vz(z), vw(w);
        cusp::blas::axpbypcz(vx, vy, vz, vw, 2.0f, 1.0f, 3.0f);
        for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(w[i]), want[i]);
        Array output(3);
        ASSERT_THROWS(cusp::blas::axpbypcz(x, y, z, output, 2.0f, 1.0f, 3.0f), cusp::invalid_input_exception);
        cusp::blas::xmy(x, y, w);
        const float prod[4] = {0, -10, 0, -15};
        for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(w[i]), prod[i]);
        cusp::blas::fill(w, 0.0f);
        cusp::blas::xmy(vx, vy, vw);
        for (int i = 0; i < 4; i++) ASSERT_EQUAL(float(w[i]), prod[i]);
        ASSERT_THROWS(cusp::blas::xmy(x, y, output), cusp::invalid_input_exception);
    }
    {
        Array x(6); const float xv[6] = {7, 5, 4, -3, 0, 4};
        for (int i = 0; i < 6; i++) x[i] = xv[i];
        cusp::blas::scal(x, 4.0f);
        const float a[6] = {28, 20, 16, -12, 0, 16};
        for (int i = 0; i < 6; i++) ASSERT_EQUAL(float(x[i]), a[i]);
        View vx(x);
        cusp::blas::scal(vx, 2.0f);
        for (int i = 0; i < 6; i++) ASSERT_EQUAL(float(x[i]), 2 * a[i]);
        cusp::blas::scal(View(x), 0.5f); // a view passed as a temporary
        for (int i = 0; i < 6; i++) ASSERT_EQUAL(float(x[i]), a[i]);
    }
    { // double, long enough for several workgroups: ties take the first position; an empty vector
        const size_t n = 300001;
        cusp::array1d<double, cusp::host_memory> h(n);
        for (size_t i = 0; i < n; i++) h[i] = double(int((unsigned(i) * 2654435761u) % 2001u) - 1000) / 8.0;
        h[123457] = -999.0; h[250000] = 999.0;
        cusp::array1d<double, Space> x(h);
        double s = 0;
        for (size_t i = 0; i < n; i++) s += std::fabs(h[i]);
        ASSERT_EQUAL(cusp::blas::amax(x), 123457);
        ASSERT_EQUAL(cusp::blas::nrmmax(x), 999.0);
        ASSERT_TRUE(std::fabs(cusp::blas::nrm1(x) - s) <= 1e-9 * s);
        cusp::array1d<double, Space> e;
        ASSERT_EQUAL(cusp::blas::nrm1(e), 0.0); ASSERT_EQUAL(cusp::blas::nrmmax(e), 0.0); ASSERT_EQUAL(cusp::blas::amax(e), 0);
    }
}
DECLARE_SPACE_UNITTEST(TestBlasRestOfTheSet);

// testing/diagonal.cu:12-78: the Jacobi preconditioner built from every format: D(x, y) and cusp::multiply(D, x, y) give 1 / a_ii
template <typename Matrix> void diagonal_preconditioner_from()
{
    typedef typename Matrix::value_type V;
    typedef typename Matrix::memory_space Space;
    cusp::array2d<V, cusp::host_memory> A(5, 5, V(0));
    A(0, 0) = 1.0; A(0, 1) = 1.0; A(0, 2) = 2.0; A(1, 0) = 3.0; A(1, 1) = 2.0; A(1, 4) = 5.0; A(2, 2) = 0.5;
    A(3, 1) = 6.0; A(3, 2) = 7.0; A(3, 3) = 4.0; A(4, 1) = 8.0; A(4, 4) = 0.25;
    cusp::csr_matrix<int, V, cusp::host_memory> H(A);
    Matrix Mx(H);
    cusp::precond::diagonal<V, Space> D(Mx);
    ASSERT_EQUAL(D.num_rows, size_t(5)); ASSERT_EQUAL(D.num_cols, size_t(5)); ASSERT_EQUAL(D.num_entries, size_t(5));
    cusp::array1d<V, Space> input(5, V(1)), output(5, V(0));
    const V expected[5] = {V(1.00), V(0.50), V(2.00), V(0.25), V(4.00)};
    D(input, output);
    for (int i = 0; i < 5; i++) ASSERT_EQUAL(V(output[i]), expected[i]);
    cusp::blas::fill(output, V(0));
    cusp::multiply(D, input, output);
    for (int i = 0; i < 5; i++) ASSERT_EQUAL(V(output[i]), expected[i]);
}
template <typename Space> void TestDiagonalPreconditioner()
{
    diagonal_preconditioner_from<cusp::csr_matrix<int, double, Space>>();
    diagonal_preconditioner_from<cusp::coo_matrix<int, double, Space>>();
    diagonal_preconditioner_from<cusp::ell_matrix<int, float, Space>>();
    diagonal_preconditioner_from<cusp::dia_matrix<int, float, Space>>();
    diagonal_preconditioner_from<cusp::hyb_matrix<int, double, Space>>();
}
DECLARE_SPACE_UNITTEST(TestDiagonalPreconditioner);

// examples/Preconditioning/diagonal.cu + testing/cg.cu's protocol: cg with the Jacobi preconditioner on a badly SCALED Poisson matrix (rows and
// columns scaled by s_i: D^-1 undoes most of it) converges in far fewer iterations than without, to the same solution
template <typename Space> void TestPreconditionedCg()
{
    cusp::csr_matrix<int, double, cusp::host_memory> H;
    cusp::gallery::poisson5pt(H, 40, 30);
    const size_t N = H.num_rows;
    std::vector<double> sc(N);
    for (size_t i = 0; i < N; i++) sc[i] = 1.0 + double((i * 7919) % 97);
    for (size_t i = 0; i < N; i++)
        for (int jj = H.row_offsets[i]; jj < H.row_offsets[i + 1]; jj++) H.values[jj] = H.values[jj] * sc[i] * sc[H.column_indices[jj]];
    cusp::csr_matrix<int, double, Space> A(H);
    cusp::array1d<double, Space> b(N, 1.0), x0(N, 0.0), x1(N, 0.0);
    cusp::monitor<double> plain(b, 20000, 1e-10), jacobi(b, 20000, 1e-10);
    cusp::krylov::cg(A, x0, b, plain);
    cusp::precond::diagonal<double, Space> M(A);
    cusp::krylov::cg(A, x1, b, jacobi, M);
    ASSERT_TRUE(plain.converged() && jacobi.converged());
    ASSERT_TRUE(jacobi.iteration_count() * 3 < plain.iteration_count());
    cusp::array1d<double, Space> r(N);
    cusp::multiply(A, x1, r);
    cusp::blas::axpby(b, r, r, 1.0, -1.0);
    ASSERT_TRUE(cusp::blas::nrm2(r) <= 1e-9 * cusp::blas::nrm2(b));
    // device_memory runs the FUSED Jacobi iteration (cmi_pcg_*_jacobi_*); $CMI_CG_FUSED_JACOBI=0 forces the operation-by-operation path: the
    // same method, so the same iteration count to within rounding and the same solution
    setenv("CMI_CG_FUSED_JACOBI", "0", 1);
    cusp::array1d<double, Space> x2(N, 0.0);
    cusp::monitor<double> generic(b, 20000, 1e-10);
    cusp::krylov::cg(A, x2, b, generic, M);
    unsetenv("CMI_CG_FUSED_JACOBI");
    ASSERT_TRUE(generic.converged());
    const long long d = (long long)generic.iteration_count() - (long long)jacobi.iteration_count();
    ASSERT_TRUE(d >= -3 && d <= 3);
    cusp::blas::axpy(x1, x2, -1.0);
    ASSERT_TRUE(cusp::blas::nrmmax(x2) <= 1e-8 * cusp::blas::nrmmax(x1));
    // float instance
    cusp::csr_matrix<int, float, cusp::host_memory> Hf;
    cusp::gallery::poisson5pt(Hf, 20, 20);
    for (size_t i = 0; i < Hf.num_rows; i++)
        for (int jj = Hf.row_offsets[i]; jj < Hf.row_offsets[i + 1]; jj++) Hf.values[jj] = Hf.values[jj] * float(1 + i % 4) * float(1 + Hf.column_indices[jj] % 4);
    cusp::csr_matrix<int, float, Space> Af(Hf);
    cusp::precond::diagonal<float, Space> Mf(Af);
    cusp::array1d<float, Space> bf(Af.num_rows, 1.0f), xf(Af.num_rows, 0.0f), rf(Af.num_rows);
    cusp::monitor<float> mf(bf, 2000, 1e-5);
    cusp::krylov::cg(Af, xf, bf, mf, Mf);
    ASSERT_TRUE(mf.converged());
    cusp::multiply(Af, xf, rf);
    cusp::blas::axpby(bf, rf, rf, 1.0f, -1.0f);
    ASSERT_TRUE(cusp::blas::nrm2(rf) <= 1e-4f * cusp::blas::nrm2(bf));
}
DECLARE_SPACE_UNITTEST(TestPreconditionedCg);

// testing/cr.cu:36-88 verbatim protocol (float, monitor(b, 20, 1e-4) on poisson5pt(10, 10); the zero-residual start)
template <typename Space> void TestConjugateResidual()
{
    {
        cusp::csr_matrix<int, float, Space> A;
        cusp::gallery::poisson5pt(A, 10, 10);
        cusp::array1d<float, Space> x(A.num_rows, 0.0f), b(A.num_rows, 1.0f);
        cusp::monitor<float> monitor(b, 20, 1e-4);
        cusp::krylov::cr(A, x, b, monitor);
        cusp::array1d<float, Space> residual(A.num_rows, 0.0f);
        cusp::multiply(A, x, residual);
        cusp::blas::axpby(residual, b, residual, -1.0f, 1.0f);
        ASSERT_EQUAL(cusp::blas::nrm2(residual) < 1e-4 * cusp::blas::nrm2(b), true);
    }
    {
        cusp::array2d<float, cusp::host_memory> D(2, 2, 0.0f);
        D(0, 0) = 8; D(1, 1) = 4;
        cusp::csr_matrix<int, float, cusp::host_memory> H(D);
        cusp::csr_matrix<int, float, Space> A(H);
        cusp::array1d<float, Space> x(A.num_rows, 1.0f), b(A.num_rows);
        cusp::multiply(A, x, b);
        cusp::monitor<float> monitor(b, 20, 0.0f);
        cusp::krylov::cr(A, x, b, monitor);
        cusp::array1d<float, Space> residual(A.num_rows, 0.0f);
        cusp::multiply(A, x, residual);
        cusp::blas::axpby(residual, b, residual, -1.0f, 1.0f);
        ASSERT_EQUAL(monitor.converged(), true);
        ASSERT_EQUAL(monitor.iteration_count(), size_t(0));
        ASSERT_EQUAL(cusp::blas::nrm2(residual), 0.0f);
    }
}
DECLARE_SPACE_UNITTEST(TestConjugateResidual);

// cr on device_memory runs a FUSED iteration (cmi_cr_*); $CMI_CR_FUSED=0 forces the operation-by-operation path: the same method -- incl. the residual
// rebuilt every 8th iteration -- so the same iteration count to within rounding and the same solution (double and float)
template <typename Space> void TestConjugateResidualFusedAgainstGeneric()
{
    cusp::csr_matrix<int, double, Space> A;
    cusp::gallery::poisson5pt(A, 60, 45);
    const size_t N = A.num_rows;
    cusp::array1d<double, Space> b(N, 1.0), x1(N, 0.0), x2(N, 0.0), r(N);
    cusp::monitor<double> m1(b, 2000, 1e-10), m2(b, 2000, 1e-10);
    cusp::krylov::cr(A, x1, b, m1);
    setenv("CMI_CR_FUSED", "0", 1);
    cusp::krylov::cr(A, x2, b, m2);
    unsetenv("CMI_CR_FUSED");
    ASSERT_TRUE(m1.converged() && m2.converged());
    ASSERT_TRUE(m1.iteration_count() > 20); // (the every-8th-iteration branch was taken several times)
    const long long d = (long long)m1.iteration_count() - (long long)m2.iteration_count();
    ASSERT_TRUE(d >= -3 && d <= 3);
    cusp::multiply(A, x1, r);
    cusp::blas::axpby(b, r, r, 1.0, -1.0);
    ASSERT_TRUE(cusp::blas::nrm2(r) <= 1e-9 * cusp::blas::nrm2(b));
    cusp::blas::axpy(x1, x2, -1.0);
    ASSERT_TRUE(cusp::blas::nrmmax(x2) <= 1e-7 * cusp::blas::nrmmax(x1));
    cusp::ell_matrix<int, float, Space> Ef;
    cusp::gallery::poisson5pt(Ef, 30, 20);
    cusp::array1d<float, Space> bf(Ef.num_rows, 1.0f), xf(Ef.num_rows, 0.0f), rf(Ef.num_rows);
    cusp::monitor<float> mf(bf, 500, 1e-5);
    cusp::krylov::cr(Ef, xf, bf, mf);
    ASSERT_TRUE(mf.converged());
    cusp::multiply(Ef, xf, rf);
    cusp::blas::axpby(bf, rf, rf, 1.0f, -1.0f);
    ASSERT_TRUE(cusp::blas::nrm2(rf) <= 1e-4f * cusp::blas::nrm2(bf));
}
DECLARE_SPACE_UNITTEST(TestConjugateResidualFusedAgainstGeneric);

// bicgstab (no test file in the reference's testing/; examples/Solvers/bicgstab.cu protocol): a NON-symmetric matrix -- 5-point diffusion plus an
// upwind convection term -- with and without the Jacobi preconditioner; the residual is checked from scratch; every format multiplies the same way
template <typename Space> void TestBicgstab()
{
    cusp::csr_matrix<int, double, cusp::host_memory> H;
    cusp::gallery::poisson5pt(H, 30, 20);
    const size_t N = H.num_rows;
    for (size_t i = 0; i < N; i++)
        for (int jj = H.row_offsets[i]; jj < H.row_offsets[i + 1]; jj++) {
            const size_t j = H.column_indices[jj];
            if (j + 1 == i) H.values[jj] -= 0.8;      // west neighbour: convection makes A non-symmetric
            if (j == i) H.values[jj] += 0.8;
        }
    cusp::csr_matrix<int, double, Space> A(H);
    cusp::array1d<double, Space> b(N, 1.0), x(N, 0.0), r(N);
    cusp::monitor<double> monitor(b, 500, 1e-9);
    cusp::krylov::bicgstab(A, x, b, monitor);
    ASSERT_TRUE(monitor.converged());
    cusp::multiply(A, x, r);
    cusp::blas::axpby(b, r, r, 1.0, -1.0);
    ASSERT_TRUE(cusp::blas::nrm2(r) <= 1e-8 * cusp::blas::nrm2(b));
    // device_memory ran the FUSED iteration (cmi_bicgstab_*); $CMI_BICGSTAB_FUSED=0 forces the operation-by-operation path: same method, same count
    // to within rounding, same solution; a float instance through the fused path as well
    setenv("CMI_BICGSTAB_FUSED", "0", 1);
    cusp::array1d<double, Space> xg(N, 0.0);
    cusp::monitor<double> generic(b, 500, 1e-9);
    cusp::krylov::bicgstab(A, xg, b, generic);
    unsetenv("CMI_BICGSTAB_FUSED");
    ASSERT_TRUE(generic.converged());
    const long long dit = (long long)generic.iteration_count() - (long long)monitor.iteration_count();
    ASSERT_TRUE(dit >= -3 && dit <= 3);
    cusp::blas::axpy(x, xg, -1.0);
    ASSERT_TRUE(cusp::blas::nrmmax(xg) <= 1e-6);
    {
        cusp::csr_matrix<int, float, cusp::host_memory> Hf(H.num_rows, H.num_cols, H.num_entries);
        for (size_t i = 0; i <= N; i++) Hf.row_offsets[i] = H.row_offsets[i];
        for (size_t k = 0; k < H.num_entries; k++) { Hf.column_indices[k] = H.column_indices[k]; Hf.values[k] = float(H.values[k]); }
        cusp::csr_matrix<int, float, Space> Af(Hf);
        cusp::array1d<float, Space> bf(N, 1.0f), xf(N, 0.0f), rf(N);
        cusp::monitor<float> mf(bf, 500, 1e-4);
        cusp::krylov::bicgstab(Af, xf, bf, mf);
        ASSERT_TRUE(mf.converged());
        cusp::multiply(Af, xf, rf);
        cusp::blas::axpby(bf, rf, rf, 1.0f, -1.0f);
        ASSERT_TRUE(cusp::blas::nrm2(rf) <= 1e-3f * cusp::blas::nrm2(bf));
        // the zero-residual start: no iteration, x untouched
        cusp::array1d<float, Space> x1(N, 1.0f), b1(N);
        cusp::multiply(Af, x1, b1);
        cusp::monitor<float> m0(b1, 20, 0.0f);
        cusp::krylov::bicgstab(Af, x1, b1, m0);
        ASSERT_EQUAL(m0.iteration_count(), size_t(0));
    }
    cusp::hyb_matrix<int, double, Space> Hy(H);
    cusp::precond::diagonal<double, Space> M(A);
    cusp::array1d<double, Space> x2(N, 0.0);
    cusp::monitor<double> monitor2(b, 500, 1e-9);
    cusp::krylov::bicgstab(Hy, x2, b, monitor2, M);
    ASSERT_TRUE(monitor2.converged());
    cusp::blas::axpy(x, x2, -1.0);
    ASSERT_TRUE(cusp::blas::nrmmax(x2) <= 1e-6);
    // cg refuses nothing here, but the method is for SPD systems: bicgstab's answer must satisfy the non-symmetric system, checked above
}
DECLARE_SPACE_UNITTEST(TestBicgstab);

// testing/gmres.cu:39-61 verbatim protocol (float, restart 20, monitor(b, 20, 1e-4) on poisson5pt(10, 10)); then a non-symmetric system with restarts
// (restart 5: several outer cycles) and the Jacobi preconditioner
template <typename Space> void TestGeneralizedMinRes()
{
    {
        cusp::csr_matrix<int, float, Space> A;
        cusp::gallery::poisson5pt(A, 10, 10);
        cusp::array1d<float, Space> x(A.num_rows, 0.0f), b(A.num_rows, 1.0f);
        cusp::monitor<float> monitor(b, 20, 1e-4);
        cusp::krylov::gmres(A, x, b, 20, monitor);
        cusp::array1d<float, Space> residual(A.num_rows, 0.0f);
        cusp::multiply(A, x, residual);
        cusp::blas::axpby(residual, b, residual, -1.0f, 1.0f);
        ASSERT_EQUAL(cusp::blas::nrm2(residual) < 1e-4 * cusp::blas::nrm2(b), true);
    }
    {
        cusp::csr_matrix<int, double, cusp::host_memory> H;
        cusp::gallery::poisson5pt(H, 20, 15);
        const size_t N = H.num_rows;
        for (size_t i = 0; i < N; i++)
            for (int jj = H.row_offsets[i]; jj < H.row_offsets[i + 1]; jj++) {
                const size_t j = H.column_indices[jj];
                if (j + 1 == i) H.values[jj] -= 0.8;
                if (j == i) H.values[jj] += 0.8 + double(i % 5);
            }
        cusp::ell_matrix<int, double, Space> A(H);
        cusp::precond::diagonal<double, Space> M(A);
        cusp::array1d<double, Space> b(N, 1.0), x(N, 0.0), r(N);
        cusp::monitor<double> monitor(b, 400, 1e-9);
        cusp::krylov::gmres(A, x, b, 5, monitor, M);
        ASSERT_TRUE(monitor.converged());
        ASSERT_TRUE(monitor.iteration_count() > 5); // more than one cycle
        cusp::multiply(A, x, r);
        cusp::blas::axpby(b, r, r, 1.0, -1.0);
        ASSERT_TRUE(cusp::blas::nrm2(r) <= 1e-6 * cusp::blas::nrm2(b)); // (the monitor watches the PRECONDITIONED residual)
    }
}
DECLARE_SPACE_UNITTEST(TestGeneralizedMinRes);

// cusp/transpose.h (testing/transpose.cu: every format against the dense transpose) and testing/bicg.cu:50-101 (verbatim protocol: A = A^T there) plus a
// NON-symmetric system with A^T from cusp::transpose
template <typename Space> void TestTransposeAndBicg()
{
    cusp::array2d<float, cusp::host_memory> D(3, 4, 0.0f);
    D(0, 0) = 10; D(0, 2) = 20; D(1, 3) = 30; D(2, 0) = 40; D(2, 1) = 50; D(2, 3) = 60;
    cusp::csr_matrix<int, float, cusp::host_memory> H(D);
    {
        cusp::csr_matrix<int, float, Space> A(H), At;
        cusp::transpose(A, At);
        ASSERT_EQUAL(At.num_rows, size_t(4)); ASSERT_EQUAL(At.num_cols, size_t(3)); ASSERT_EQUAL(At.num_entries, size_t(6));
        cusp::array2d<float, cusp::host_memory> T(At);
        for (size_t i = 0; i < 3; i++) for (size_t j = 0; j < 4; j++) ASSERT_EQUAL(float(T(j, i)), float(D(i, j)));
        cusp::coo_matrix<int, float, Space> C(A), Ct;
        cusp::transpose(C, Ct);
        ASSERT_TRUE(Ct.is_sorted_by_row_and_column());
        cusp::ell_matrix<int, float, Space> E(A), Et;
        cusp::transpose(E, Et);
        cusp::array2d<float, cusp::host_memory> T2(Et);
        for (size_t i = 0; i < 3; i++) for (size_t j = 0; j < 4; j++) ASSERT_EQUAL(float(T2(j, i)), float(D(i, j)));
    }
    {
        cusp::csr_matrix<int, float, Space> A;
        cusp::gallery::poisson5pt(A, 10, 10);
        cusp::array1d<float, Space> x(A.num_rows, 0.0f), b(A.num_rows, 1.0f);
        cusp::monitor<float> monitor(b, 20, 1e-4);
        cusp::krylov::bicg(A, A, x, b, monitor);
        cusp::array1d<float, Space> residual(A.num_rows, 0.0f);
        cusp::multiply(A, x, residual);
        cusp::blas::axpby(residual, b, residual, -1.0f, 1.0f);
        ASSERT_EQUAL(cusp::blas::nrm2(residual) < 1e-4 * cusp::blas::nrm2(b), true);
        cusp::array1d<float, Space> x1(A.num_rows, 1.0f), b1(A.num_rows);
        cusp::multiply(A, x1, b1);
        cusp::monitor<float> m0(b1, 20, 0.0f);
        cusp::krylov::bicg(A, A, x1, b1, m0);
        ASSERT_EQUAL(m0.converged(), true); ASSERT_EQUAL(m0.iteration_count(), size_t(0));
    }
    {
        cusp::csr_matrix<int, double, cusp::host_memory> G;
        cusp::gallery::poisson5pt(G, 25, 16);
        const size_t N = G.num_rows;
        for (size_t i = 0; i < N; i++)
            for (int jj = G.row_offsets[i]; jj < G.row_offsets[i + 1]; jj++) {
                const size_t j = G.column_indices[jj];
                if (j + 1 == i) G.values[jj] -= 0.6;
                if (j == i) G.values[jj] += 0.6;
            }
        cusp::csr_matrix<int, double, Space> A(G), At;
        cusp::transpose(A, At);
        cusp::array1d<double, Space> b(N, 1.0), x(N, 0.0), r(N);
        cusp::monitor<double> monitor(b, 1000, 1e-9);
        cusp::krylov::bicg(A, At, x, b, monitor);
        ASSERT_TRUE(monitor.converged());
        cusp::multiply(A, x, r);
        cusp::blas::axpby(b, r, r, 1.0, -1.0);
        ASSERT_TRUE(cusp::blas::nrm2(r) <= 1e-8 * cusp::blas::nrm2(b));
        cusp::precond::diagonal<double, Space> M(A), Mt(At);
        cusp::array1d<double, Space> x2(N, 0.0);
        cusp::monitor<double> monitor2(b, 1000, 1e-9);
        cusp::krylov::bicg(A, At, x2, b, monitor2, M, Mt);
        ASSERT_TRUE(monitor2.converged());
        cusp::blas::axpy(x, x2, -1.0);
        ASSERT_TRUE(cusp::blas::nrmmax(x2) <= 1e-6);
    }
}
DECLARE_SPACE_UNITTEST(TestTransposeAndBicg);

// every solver on the same seeded, irregular, strictly diagonally dominant systems (rows of 2..12 scattered off-diagonal entries): the non-symmetric one
// through bicgstab / gmres / bicg, its symmetric part (SPD) through cg / cr / Jacobi-cg as well -- in four storage formats -- must all find the solution
// a host cg / bicgstab finds, to the tolerance asked for
template <typename Space> void TestSolversAgreeOnRandomSystems()
{
    const size_t N = 3000;
    unsigned s = 2024u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffffu; };
    std::vector<std::vector<std::pair<int, double>>> rows(N);
    for (size_t i = 0; i < N; i++) {
        const int k = 2 + int(rnd() % 11);
        for (int e = 0; e < k; e++) {
            const int j = int(rnd() % N);
            if (size_t(j) != i) rows[i].push_back({j, double(int(rnd() % 2001) - 1000) / 1000.0});
        }
    }
    auto build = [&](bool symmetric) {
        std::vector<std::map<int, double>> m(N);
        for (size_t i = 0; i < N; i++)
            for (auto &e : rows[i]) {
                if (symmetric) { m[i][e.first] += 0.5 * e.second; m[e.first][int(i)] += 0.5 * e.second; }
                else m[i][e.first] += e.second;
            }
        size_t nnz = 0;
        for (size_t i = 0; i < N; i++) { double a = 0; for (auto &e : m[i]) a += std::fabs(e.second); m[i][int(i)] = a + 1.0; nnz += m[i].size(); }
        cusp::csr_matrix<int, double, cusp::host_memory> H(N, N, nnz);
        size_t p = 0;
        for (size_t i = 0; i < N; i++) { H.row_offsets[i] = int(p); for (auto &e : m[i]) { H.column_indices[p] = e.first; H.values[p] = e.second; p++; } }
        H.row_offsets[N] = int(p);
        return H;
    };
    cusp::array1d<double, cusp::host_memory> hb(N);
    for (size_t i = 0; i < N; i++) hb[i] = double(int(rnd() % 201) - 100) / 50.0;
    const double tol = 1e-10;
    auto close_to = [&](const cusp::array1d<double, Space> &x, const cusp::array1d<double, cusp::host_memory> &ref) {
        cusp::array1d<double, cusp::host_memory> h(x);
        double worst = 0, scale = 0;
        for (size_t i = 0; i < N; i++) { worst = std::max(worst, std::fabs(h[i] - ref[i])); scale = std::max(scale, std::fabs(ref[i])); }
        return worst <= 1e-7 * scale;
    };
    { // SPD
        const auto H = build(true);
        cusp::array1d<double, cusp::host_memory> ref(N, 0.0);
        { cusp::monitor<double> m(hb, 5000, tol); cusp::krylov::cg(H, ref, hb, m); ASSERT_TRUE(m.converged()); }
        cusp::array1d<double, Space> b(hb);
        cusp::csr_matrix<int, double, Space> A(H);
        cusp::ell_matrix<int, double, Space> E(H);
        cusp::hyb_matrix<int, double, Space> Y(H);
        cusp::coo_matrix<int, double, Space> C(H);
        cusp::precond::diagonal<double, Space> M(A);
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::cg(A, x, b, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::cg(E, x, b, m, M); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::cr(Y, x, b, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::bicgstab(C, x, b, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::gmres(A, x, b, 30, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
    }
    { // non-symmetric
        const auto H = build(false);
        cusp::array1d<double, cusp::host_memory> ref(N, 0.0);
        { cusp::monitor<double> m(hb, 5000, tol); cusp::krylov::bicgstab(H, ref, hb, m); ASSERT_TRUE(m.converged()); }
        cusp::array1d<double, Space> b(hb);
        cusp::csr_matrix<int, double, Space> A(H), At;
        cusp::transpose(A, At);
        cusp::ell_matrix<int, double, Space> E(H);
        cusp::hyb_matrix<int, double, Space> Y(H);
        cusp::precond::diagonal<double, Space> M(A);
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::bicgstab(A, x, b, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::bicgstab(E, x, b, m, M); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::gmres(Y, x, b, 25, m, M); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
        { cusp::array1d<double, Space> x(N, 0.0); cusp::monitor<double> m(b, 5000, tol); cusp::krylov::bicg(A, At, x, b, m); ASSERT_TRUE(m.converged() && close_to(x, ref)); }
    }
}
DECLARE_SPACE_UNITTEST(TestSolversAgreeOnRandomSystems);

// ELLR (the fork's container, testing/ktt.cu:26-43 runs its kernels on dia / ell / ellr)
template <typename Space> void TestEllrMatrix()
{
    cusp::csr_matrix<int, double, cusp::host_memory> H;
    cusp::gallery::poisson5pt(H, 7, 5);
    cusp::ktt::ellr_matrix<int, double, Space> R(H);
    cusp::array1d<int, cusp::host_memory> len(R.row_lengths);
    for (size_t i = 0; i < H.num_rows; i++) ASSERT_EQUAL(len[i], H.row_offsets[i + 1] - H.row_offsets[i]);
    cusp::array1d<double, cusp::host_memory> x(35), y(35, 10);
    for (int i = 0; i < 35; i++) x[i] = i % 10;
    cusp::multiply(H, x, y);
    cusp::array1d<double, Space> _x(x), _y(35, 10);
    cusp::ktt::multiply(R, _x, _y);
    ASSERT_ARRAYS_EQUAL(_y, y);
}
DECLARE_SPACE_UNITTEST(TestEllrMatrix);
