// Host build of the cusp:: layer tests: host_memory only, runs without a GPU (-m "not gpu").
#define TEST_SPACE cusp::host_memory
#define TEST_SPACE_NAME "host_memory"
#include "spmv_tests.h"

// the fixture the reference's tests hold: testing/data/laplacian/5pt_10x10.mtx == poisson5pt(10,10)
static std::string g_golden_dir;
void TestMatrixMarketFixture()
{
    cusp::csr_matrix<int, double, cusp::host_memory> F, G;
    cusp::io::read_matrix_market_file(F, g_golden_dir + "/5pt_10x10.mtx");
    cusp::gallery::poisson5pt(G, 10, 10);
    ASSERT_EQUAL(F.num_rows, size_t(100)); ASSERT_EQUAL(F.num_entries, size_t(460));
    ASSERT_ARRAYS_EQUAL(F.row_offsets, G.row_offsets);
    ASSERT_ARRAYS_EQUAL(F.column_indices, G.column_indices);
    ASSERT_ARRAYS_EQUAL(F.values, G.values);
    // write -> read round trip
    std::stringstream ss;
    cusp::io::write_matrix_market_stream(G, ss);
    cusp::coo_matrix<int, double, cusp::host_memory> back;
    cusp::io::read_matrix_market_stream(back, ss);
    cusp::csr_matrix<int, double, cusp::host_memory> b2(back);
    ASSERT_ARRAYS_EQUAL(b2.column_indices, G.column_indices);
    // symmetric + pattern expansion
    std::stringstream sym("%%MatrixMarket matrix coordinate pattern symmetric\n% comment\n3 3 3\n1 1\n3 1\n2 2\n");
    cusp::coo_matrix<int, float, cusp::host_memory> S;
    cusp::io::read_matrix_market_stream(S, sym);
    ASSERT_EQUAL(S.num_entries, size_t(4));
    const int r[4] = {0, 0, 1, 2}, c[4] = {0, 2, 1, 0};
    for (int i = 0; i < 4; i++) { ASSERT_EQUAL(S.row_indices[i], r[i]); ASSERT_EQUAL(S.column_indices[i], c[i]); ASSERT_EQUAL(S.values[i], 1.0f); }
    std::stringstream bad("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n");
    ASSERT_THROWS(cusp::io::read_matrix_market_stream(S, bad), cusp::io_exception);
}
DECLARE_UNITTEST(TestMatrixMarketFixture);

void TestFormatConversionGuards()
{
    // reference csr_to_other.h:97-103,178-184: refuse > 3x fill-in once it exceeds 1e6 slots
    const size_t n = 400000;
    cusp::coo_matrix<int, float, cusp::host_memory> coo(n, n, n + 3);
    for (size_t i = 0; i < n; i++) { coo.row_indices[i] = int(i); coo.column_indices[i] = int(i); coo.values[i] = 1; }
    for (size_t k = 0; k < 3; k++) { coo.row_indices[n + k] = 0; coo.column_indices[n + k] = int(100 + 7 * k); coo.values[n + k] = 2; }
    coo.sort_by_row_and_column();
    cusp::csr_matrix<int, float, cusp::host_memory> csr(coo);
    cusp::ell_matrix<int, float, cusp::host_memory> ell;
    ASSERT_THROWS(ell = csr, cusp::format_conversion_exception); // width 4 x 400000 rows vs 400003 entries
    cusp::dia_matrix<int, float, cusp::host_memory> dia;
    ASSERT_THROWS(dia = csr, cusp::format_conversion_exception);
    cusp::hyb_matrix<int, float, cusp::host_memory> hyb(csr); // HYB always works: width 1 + 3 COO entries
    ASSERT_EQUAL(hyb.ell.column_indices.num_cols, size_t(1)); ASSERT_EQUAL(hyb.coo.num_entries, size_t(3));
}
DECLARE_UNITTEST(TestFormatConversionGuards);

void TestGenericFunctorsOnHost()
{
    // the 7-argument multiply is fully generic on host_memory (cusp/multiply.h:113): max-plus algebra
    cusp::csr_matrix<int, double, cusp::host_memory> A;
    cusp::gallery::poisson5pt(A, 3, 3);
    cusp::array1d<double, cusp::host_memory> x(9), y(9, -100);
    for (int i = 0; i < 9; i++) x[i] = i;
    struct maxf { double operator()(double a, double b) const { return a > b ? a : b; } };
    cusp::multiply(A, x, y, cusp::identity_function<double>(), cusp::plus<double>(), maxf());
    ASSERT_EQUAL(y[4], 8.0); // max over {-1+1, -1+3, 4+4, -1+5, -1+7, -100}
}
DECLARE_UNITTEST(TestGenericFunctorsOnHost);

// reference testing/poisson.cu:27-92: exact dense expectations for the 9-, 7- and 27-point stencils
void TestPoissonOtherStencils()
{
    {
        cusp::dia_matrix<int, float, cusp::host_memory> matrix;
        cusp::gallery::poisson9pt(matrix, 2, 3);
        cusp::array2d<float, cusp::host_memory> R(matrix);
        const float E[6][6] = {{8, -1, -1, -1, 0, 0}, {-1, 8, -1, -1, 0, 0}, {-1, -1, 8, -1, -1, -1}, {-1, -1, -1, 8, -1, -1}, {0, 0, -1, -1, 8, -1}, {0, 0, -1, -1, -1, 8}};
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) ASSERT_EQUAL(R(i, j), E[i][j]);
    }
    {
        cusp::csr_matrix<int, float, cusp::host_memory> matrix;
        cusp::gallery::poisson7pt(matrix, 2, 2, 2);
        cusp::array2d<float, cusp::host_memory> R(matrix);
        const float E[8][8] = {{6, -1, -1, 0, -1, 0, 0, 0}, {-1, 6, 0, -1, 0, -1, 0, 0}, {-1, 0, 6, -1, 0, 0, -1, 0}, {0, -1, -1, 6, 0, 0, 0, -1},
                               {-1, 0, 0, 0, 6, -1, -1, 0}, {0, -1, 0, 0, -1, 6, 0, -1}, {0, 0, -1, 0, -1, 0, 6, -1}, {0, 0, 0, -1, 0, -1, -1, 6}};
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) ASSERT_EQUAL(R(i, j), E[i][j]);
    }
    {
        cusp::coo_matrix<int, double, cusp::host_memory> matrix;
        cusp::gallery::poisson27pt(matrix, 2, 2, 2);
        cusp::array2d<double, cusp::host_memory> R(matrix);
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) ASSERT_EQUAL(R(i, j), i == j ? 26.0 : -1.0);
        cusp::csr_matrix<int, double, cusp::host_memory> big;
        cusp::gallery::poisson27pt(big, 12, 11, 10); // interior rows hold 27 entries
        ASSERT_EQUAL(cusp::compute_max_entries_per_row(big.row_offsets), size_t(27));
    }
}
DECLARE_UNITTEST(TestPoissonOtherStencils);

int main(int argc, char **argv)
{
    g_golden_dir = GOLDEN_DIR;
    return unittest::run_all(argc, argv);
}
