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

// The reference's own data files (testing/data/test/*.mtx) with the dense answers its tests expect
// (testing/matrix_market.cu:60-100, 146-186, 188-215, 217-260)
void TestReadMatrixMarketReferenceFiles()
{
    const std::string dir = g_golden_dir + "/ref_data/test/";
    {
        cusp::coo_matrix<int, float, cusp::host_memory> coo;
        cusp::io::read_matrix_market_file(coo, dir + "coordinate_real_general.mtx");
        cusp::array2d<float, cusp::host_memory> D(coo);
        const float E[5][5] = {{1.0f, 0, 0, 6.0f, 0}, {0, 10.5f, 0, 0, 0}, {0, 0, 0.25f, 0, 0}, {0, 250.5f, 0, -250.0f, 38.75f}, {0, 0, 0, 0, 12.0f}};
        ASSERT_EQUAL(D.num_rows, size_t(5)); ASSERT_EQUAL(D.num_cols, size_t(5));
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) ASSERT_EQUAL(D(i, j), E[i][j]);
        cusp::csr_matrix<int, float, cusp::host_memory> csr;                      // :217-260, straight into CSR
        cusp::io::read_matrix_market_file(csr, dir + "coordinate_real_general.mtx");
        cusp::array2d<float, cusp::host_memory> D2(csr);
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) ASSERT_EQUAL(D2(i, j), E[i][j]);
    }
    {
        cusp::coo_matrix<int, float, cusp::host_memory> coo;
        cusp::io::read_matrix_market_file(coo, dir + "coordinate_pattern_symmetric.mtx");
        cusp::array2d<float, cusp::host_memory> D(coo);
        const float E[5][5] = {{1, 0, 0, 0, 0}, {0, 1, 0, 1, 0}, {0, 0, 1, 0, 0}, {0, 1, 0, 1, 1}, {0, 0, 0, 1, 1}};
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) ASSERT_EQUAL(D(i, j), E[i][j]);
    }
    {
        cusp::coo_matrix<int, float, cusp::host_memory> coo;
        cusp::io::read_matrix_market_file(coo, dir + "array_real_general.mtx");
        cusp::array2d<float, cusp::host_memory> D(coo);
        ASSERT_EQUAL(D.num_rows, size_t(4)); ASSERT_EQUAL(D.num_cols, size_t(3));
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) ASSERT_EQUAL(D(i, j), float(1 + i + 4 * j));
    }
}
DECLARE_UNITTEST(TestReadMatrixMarketReferenceFiles);

// testing/data/laplacian/*.mtx are what cusp::gallery produces: the files pin the generic stencil builder
void TestGalleryAgainstReferenceLaplacianFiles()
{
    const std::string dir = g_golden_dir + "/ref_data/laplacian/";
    auto same = [](const cusp::csr_matrix<int, double, cusp::host_memory> &F, const cusp::csr_matrix<int, double, cusp::host_memory> &G) {
        ASSERT_EQUAL(F.num_rows, G.num_rows); ASSERT_EQUAL(F.num_cols, G.num_cols); ASSERT_EQUAL(F.num_entries, G.num_entries);
        ASSERT_ARRAYS_EQUAL(F.row_offsets, G.row_offsets);
        ASSERT_ARRAYS_EQUAL(F.column_indices, G.column_indices);
        ASSERT_ARRAYS_EQUAL(F.values, G.values);
    };
    cusp::csr_matrix<int, double, cusp::host_memory> F, G;
    cusp::io::read_matrix_market_file(F, dir + "9pt_10x10.mtx");
    cusp::gallery::poisson9pt(G, 10, 10);
    same(F, G);
    cusp::io::read_matrix_market_file(F, dir + "7pt_10x10x10.mtx"); // this file's diagonal is 7, not poisson7pt's 6
    cusp::gallery::generate_matrix_from_stencil(G, {{0, 0, -1, -1.0}, {0, -1, 0, -1.0}, {-1, 0, 0, -1.0}, {0, 0, 0, 7.0}, {1, 0, 0, -1.0}, {0, 1, 0, -1.0}, {0, 0, 1, -1.0}}, 10, 10, 10);
    same(F, G);
    cusp::csr_matrix<int, double, cusp::host_memory> P;
    cusp::gallery::poisson7pt(P, 10, 10, 10);
    ASSERT_ARRAYS_EQUAL(F.row_offsets, P.row_offsets);       // same pattern as the gallery's 7-point matrix
    ASSERT_ARRAYS_EQUAL(F.column_indices, P.column_indices);
    cusp::io::read_matrix_market_file(F, dir + "3pt_100.mtx");
    cusp::gallery::generate_matrix_from_stencil(G, {{-1, 0, 0, -1.0}, {0, 0, 0, 2.0}, {1, 0, 0, -1.0}}, 100, 1);
    same(F, G);
}
DECLARE_UNITTEST(TestGalleryAgainstReferenceLaplacianFiles);

// testing/data/random_10x10/NNN_nonzeros.mtx (0 ... 100 entries in a 10x10 matrix): every format built from
// the file multiplies like the dense matrix (integer entries, x in halves: every sum is exact in any order)
void TestRandom10x10FilesAllFormats()
{
    const char *names[] = {"000", "001", "002", "005", "008", "010", "015", "020", "030", "050", "080", "100"};
    for (const char *nm : names) {
        cusp::coo_matrix<int, double, cusp::host_memory> coo;
        cusp::io::read_matrix_market_file(coo, g_golden_dir + "/ref_data/random_10x10/" + nm + "_nonzeros.mtx");
        ASSERT_EQUAL(coo.num_rows, size_t(10)); ASSERT_EQUAL(coo.num_entries, size_t(std::atoi(nm)));
        cusp::array2d<double, cusp::host_memory> D(coo);
        cusp::array1d<double, cusp::host_memory> x(10), want(10, 0.0);
        for (int j = 0; j < 10; j++) x[j] = 0.5 * j - 2.0;
        for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++) want[i] += D(i, j) * x[j];
        cusp::csr_matrix<int, double, cusp::host_memory> csr(coo);
        cusp::hyb_matrix<int, double, cusp::host_memory> hyb(coo);
        cusp::array1d<double, cusp::host_memory> y(10, -1.0);
        cusp::multiply(coo, x, y); ASSERT_ARRAYS_EQUAL(y, want);
        cusp::multiply(csr, x, y); ASSERT_ARRAYS_EQUAL(y, want);
        cusp::multiply(hyb, x, y); ASSERT_ARRAYS_EQUAL(y, want);
        if (coo.num_entries) {
            cusp::ell_matrix<int, double, cusp::host_memory> ell(coo);
            cusp::dia_matrix<int, double, cusp::host_memory> dia(coo);
            cusp::multiply(ell, x, y); ASSERT_ARRAYS_EQUAL(y, want);
            cusp::multiply(dia, x, y); ASSERT_ARRAYS_EQUAL(y, want);
        }
    }
}
DECLARE_UNITTEST(TestRandom10x10FilesAllFormats);

int main(int argc, char **argv)
{
    g_golden_dir = GOLDEN_DIR;
    return unittest::run_all(argc, argv);
}
