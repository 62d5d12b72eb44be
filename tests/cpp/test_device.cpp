// Device build of the cusp:: layer tests: every test of spmv_tests.h with device_memory containers,
// i.e. through the C-ABI and the gfx950 kernels (-m gpu).
#define TEST_SPACE cusp::device_memory
#define TEST_SPACE_NAME "device_memory"
#include "spmv_tests.h"

void TestDeviceFunctorsNotImplemented()
{
    cusp::csr_matrix<int, double, cusp::device_memory> A;
    cusp::gallery::poisson5pt(A, 3, 3);
    cusp::array1d<double, cusp::device_memory> x(9, 1.0), y(9, 0.0);
    struct maxf { double operator()(double a, double b) const { return a > b ? a : b; } };
    // no silent host fallback for functors the kernels do not implement
    ASSERT_THROWS(cusp::multiply(A, x, y, cusp::identity_function<double>(), cusp::plus<double>(), maxf()), cusp::not_implemented_exception);
}
DECLARE_UNITTEST(TestDeviceFunctorsNotImplemented);

void TestStreamPolicy()
{
    void *stream = nullptr;
    cusp::detail::check(cmi_stream_create(&stream));
    cusp::csr_matrix<int, double, cusp::device_memory> A;
    cusp::gallery::poisson5pt(A, 50, 40);
    cusp::csr_matrix<int, double, cusp::host_memory> H(A);
    cusp::array1d<double, cusp::host_memory> x(2000), y(2000, 10);
    for (int i = 0; i < 2000; i++) x[i] = (i % 21) - 10;
    cusp::multiply(H, x, y);
    cusp::array1d<double, cusp::device_memory> _x(x), _y(2000, 10);
    cusp::multiply(cusp::hip::par.on(stream), A, _x, _y);
    cusp::detail::check(cmi_stream_synchronize(stream));
    ASSERT_ARRAYS_EQUAL(_y, y);
    cusp::detail::check(cmi_stream_destroy(stream));
}
DECLARE_UNITTEST(TestStreamPolicy);

void TestLargePoissonAllFormatsAgreeWithHost()
{
    // 1000 x 1000 grid, fp64: device formats vs the host CSR loop, bit for bit for the one-lane-per-row
    // kernels (CSR default, ELL, DIA), to 1e-6 relative for COO / HYB's COO half
    const size_t m = 1000, n = 1000, N = m * n;
    cusp::csr_matrix<int, double, cusp::device_memory> A;
    cusp::gallery::poisson5pt(A, m, n);
    cusp::csr_matrix<int, double, cusp::host_memory> H(A);
    cusp::array1d<double, cusp::host_memory> x(N), y(N, 10);
    for (size_t i = 0; i < N; i++) x[i] = double((unsigned(i) * 2654435761u) % 1000u) / 997.0 - 0.5;
    cusp::multiply(H, x, y);
    cusp::array1d<double, cusp::device_memory> _x(x), _y(N, 10);
    cusp::multiply(A, _x, _y);
    ASSERT_ARRAYS_EQUAL(_y, y);
    cusp::ell_matrix<int, double, cusp::device_memory> E(A);
    cusp::blas::fill(_y, 10.0); cusp::multiply(E, _x, _y); ASSERT_ARRAYS_EQUAL(_y, y);
    cusp::dia_matrix<int, double, cusp::device_memory> D;
    cusp::gallery::poisson5pt(D, m, n);
    cusp::blas::fill(_y, 10.0); cusp::multiply(D, _x, _y); ASSERT_ARRAYS_EQUAL(_y, y);
    cusp::coo_matrix<int, double, cusp::device_memory> C(A);
    cusp::blas::fill(_y, 10.0); cusp::multiply(C, _x, _y);
    cusp::array1d<double, cusp::host_memory> got(_y);
    for (size_t i = 0; i < N; i++) ASSERT_TRUE(std::fabs(got[i] - y[i]) <= 1e-6 * 8.0);
}
DECLARE_UNITTEST(TestLargePoissonAllFormatsAgreeWithHost);

// cusp::ktt::tune (the fork's entry point, testing/ktt.cu:142-202): every configuration validated, the best
// installed, later multiplies still exact
template <typename Matrix> void tune_one_format(const char *name)
{
    cusp::csr_matrix<int, double, cusp::host_memory> H;
    cusp::gallery::poisson5pt(H, 300, 200);
    Matrix A(H);
    const size_t N = H.num_rows;
    cusp::array1d<double, cusp::host_memory> x(N), y(N, 10);
    for (size_t i = 0; i < N; i++) x[i] = double(i % 21) - 10;
    cusp::multiply(H, x, y);
    cusp::array1d<double, cusp::device_memory> _x(x), _y(N, 10);
    std::vector<cusp::ktt::tuning_result> res = cusp::ktt::tune(A, _x, _y, 5);
    ASSERT_TRUE(res.size() >= 10);
    size_t valid = 0;
    for (auto &r : res) valid += r.valid;
    ASSERT_TRUE(valid >= res.size() - 2); // every configuration of the space computes the right answer
    ASSERT_TRUE(res[0].valid && res[0].milliseconds > 0 && res[0].milliseconds <= res[valid - 1].milliseconds);
    cusp::blas::fill(_y, 10.0);
    cusp::multiply(A, _x, _y); // now runs the configuration tune() installed
    ASSERT_ARRAYS_EQUAL(_y, y);
    std::printf("        tune(%s): %zu configurations, best %.4f ms (kernel %d, block %d), worst valid %.4f ms\n", name, res.size(),
                res[0].milliseconds, res[0].config.kernel, res[0].config.block_size, res[valid - 1].milliseconds);
}
void TestKttTune()
{
    tune_one_format<cusp::csr_matrix<int, double, cusp::device_memory>>("csr");
    tune_one_format<cusp::ell_matrix<int, double, cusp::device_memory>>("ell");
    tune_one_format<cusp::dia_matrix<int, double, cusp::device_memory>>("dia");
    tune_one_format<cusp::coo_matrix<int, double, cusp::device_memory>>("coo");
    cusp::ktt::reset_tuning(); // back to the built-in heuristics (ktt.inl:130-142)
}
DECLARE_UNITTEST(TestKttTune);

// Entries in any order (the reference's device multiply refuses them, coo_flat_spmv.h:139-145): sort_by_row on the device is
// stable, so the sorted matrix through its plan has the bits of the host COO loop on the entries as given; coo -> csr of the
// unsorted device matrix (device sort of a copy) equals the host conversion.
void TestUnsortedCooSortOnDevice()
{
    const size_t rows = 5000, cols = 4000, n = 60000;
    cusp::coo_matrix<int, double, cusp::host_memory> H(rows, cols, n);
    unsigned s = 12345u;
    for (size_t k = 0; k < n; k++) {
        s = s * 1664525u + 1013904223u; H.row_indices[k] = int((s >> 8) % rows);
        s = s * 1664525u + 1013904223u; H.column_indices[k] = int((s >> 8) % cols);
        s = s * 1664525u + 1013904223u; H.values[k] = double(int((s >> 8) % 2001u) - 1000) / 997.0;
    }
    cusp::array1d<double, cusp::host_memory> x(cols), y(rows, 0.0);
    for (size_t i = 0; i < cols; i++) x[i] = double((unsigned(i) * 2654435761u) % 1000u) / 997.0 - 0.5;
    cusp::multiply(H, x, y); // host loop, storage order
    cusp::coo_matrix<int, double, cusp::device_memory> D(H);
    ASSERT_TRUE(!D.is_sorted_by_row());
    cusp::csr_matrix<int, double, cusp::device_memory> fromUnsorted(D); // device: sort a copy, offsets
    cusp::csr_matrix<int, double, cusp::host_memory> hostCsr(H);        // host: stable sort, offsets
    ASSERT_ARRAYS_EQUAL(fromUnsorted.row_offsets, hostCsr.row_offsets);
    ASSERT_ARRAYS_EQUAL(fromUnsorted.column_indices, hostCsr.column_indices);
    ASSERT_ARRAYS_EQUAL(fromUnsorted.values, hostCsr.values);
    ASSERT_TRUE(!D.is_sorted_by_row()); // the source was not touched
    D.sort_by_row();
    ASSERT_TRUE(D.is_sorted_by_row());
    cusp::array1d<double, cusp::device_memory> _x(x), _y(rows, 7.0);
    cusp::multiply(D, _x, _y);
    ASSERT_ARRAYS_EQUAL(_y, y);
    H.sort_by_row();
    ASSERT_ARRAYS_EQUAL(D.row_indices, H.row_indices);
    ASSERT_ARRAYS_EQUAL(D.column_indices, H.column_indices);
    ASSERT_ARRAYS_EQUAL(D.values, H.values);
    // float instance
    cusp::coo_matrix<int, float, cusp::host_memory> Hf(H.num_rows, H.num_cols, n);
    for (size_t k = 0; k < n; k++) { Hf.row_indices[k] = H.row_indices[n - 1 - k]; Hf.column_indices[k] = H.column_indices[n - 1 - k]; Hf.values[k] = float(H.values[n - 1 - k]); }
    cusp::coo_matrix<int, float, cusp::device_memory> Df(Hf);
    Df.sort_by_row_and_column();
    Hf.sort_by_row_and_column();
    ASSERT_TRUE(Df.is_sorted_by_row_and_column());
    ASSERT_ARRAYS_EQUAL(Df.column_indices, Hf.column_indices);
    ASSERT_ARRAYS_EQUAL(Df.values, Hf.values);
}
DECLARE_UNITTEST(TestUnsortedCooSortOnDevice);

int main(int argc, char **argv) { return unittest::run_all(argc, argv); }
