// tools/spmv_bench.cpp -- the reference's SpMV benchmark driver re-hosted on the MI355X engine
// (reference performance/spmv/spmv.cu:46-131 CLI, benchmark.h:17-47 correctness check,
// benchmark.h:84-120 timing protocol, bytes_per_spmv.h:9-72 byte models, utility.h:61-73 error norm).
//
//   spmv_bench                        poisson5pt(512,512), the reference's default input
//   spmv_bench --grid=3162            poisson5pt(3162,3162), the headline matrix
//   spmv_bench --stencil=27 --grid=150   27-point stencil on 150^3 (3.4M rows, ~27 entries/row: an nlpkkt120-like surrogate)
//   spmv_bench my_matrix.mtx          a MatrixMarket file (SuiteSparse: nlpkkt120, ldoor, thermal2 ...)
//   options: --value_type={float,double}  (default double)
//
// For each of COO / CSR / DIA / ELL / HYB: convert the host CSR matrix, copy to the device, check
// y = A x against the host multiply (relative L2 error, x[i] = (i % 21) - 10), then time with the
// reference protocol: 1 warm-up, clamp(3 s / t1, 100, 500) iterations between two events.  Prints
// GFLOP/s = 2 nnz / t and GB/s with BOTH byte models: the reference's (x counted once per entry, y read
// and written) and the compulsory one of SURVEY.md 8(d) (every array once).  A format whose fill-in
// the conversion refuses (format_conversion_exception, as in the reference) is reported and skipped.
//
//   g++ -std=c++17 -O2 -I cusp-autotuned_amd/include tools/spmv_bench.cpp -L cusp-autotuned_amd/lib -lcusp_mi355x -o tools/bin/spmv_bench
#include <cusp/coo_matrix.h>
#include <cusp/csr_matrix.h>
#include <cusp/dia_matrix.h>
#include <cusp/ell_matrix.h>
#include <cusp/hyb_matrix.h>
#include <cusp/gallery/poisson.h>
#include <cusp/io/matrix_market.h>
#include <cusp/multiply.h>

#include <cmath>
#include <cstdio>
#include <iostream>
#include <map>
#include <string>

typedef std::map<std::string, std::string> ArgumentMap;
static ArgumentMap args;

static std::string process_args(int argc, char **argv)
{
    std::string filename;
    for (int i = 1; i < argc; i++) {
        std::string arg(argv[i]);
        if (arg.substr(0, 2) == "--") {
            std::string::size_type n = arg.find('=', 2);
            if (n == std::string::npos) args[arg.substr(2)] = std::string();
            else args[arg.substr(2, n - 2)] = arg.substr(n + 1);
        } else filename = arg;
    }
    return filename;
}

// reference bytes_per_spmv.h (int32 indices)
template <typename V> struct bytes_model {
    template <typename I> static double ref(const cusp::csr_matrix<I, V, cusp::host_memory> &m)
    { return 2.0 * sizeof(I) * m.num_rows + sizeof(I) * m.num_entries + 2.0 * sizeof(V) * m.num_entries + 2.0 * sizeof(V) * m.num_rows; }
    template <typename I> static double ref(const cusp::coo_matrix<I, V, cusp::host_memory> &m)
    { return 2.0 * sizeof(I) * m.num_entries + 2.0 * sizeof(V) * m.num_entries + 2.0 * sizeof(V) * m.num_rows; }
    template <typename I> static double ref(const cusp::ell_matrix<I, V, cusp::host_memory> &m)
    { return 1.0 * sizeof(V) * m.num_rows * m.values.num_cols + (sizeof(I) + sizeof(V)) * (double)m.num_entries + 2.0 * sizeof(V) * m.num_rows; }
    template <typename I> static double ref(const cusp::dia_matrix<I, V, cusp::host_memory> &m)
    { return 2.0 * sizeof(V) * m.num_entries + 2.0 * sizeof(V) * m.num_rows; }
    template <typename I> static double ref(const cusp::hyb_matrix<I, V, cusp::host_memory> &m) { return ref(m.ell) + ref(m.coo); }
    // compulsory: every array once, x once, y once
    template <typename I> static double min(const cusp::csr_matrix<I, V, cusp::host_memory> &m)
    { return sizeof(I) * (m.num_rows + 1.0) + (sizeof(I) + sizeof(V)) * (double)m.num_entries + sizeof(V) * (double)(m.num_rows + m.num_cols); }
    template <typename I> static double min(const cusp::coo_matrix<I, V, cusp::host_memory> &m)
    { return (2.0 * sizeof(I) + sizeof(V)) * m.num_entries + sizeof(V) * (double)(m.num_rows + m.num_cols); }
    template <typename I> static double min(const cusp::ell_matrix<I, V, cusp::host_memory> &m)
    { return (sizeof(I) + sizeof(V)) * (double)m.values.num_cols * m.values.pitch + sizeof(V) * (double)(m.num_rows + m.num_cols); }
    template <typename I> static double min(const cusp::dia_matrix<I, V, cusp::host_memory> &m)
    { return sizeof(V) * (double)m.values.num_cols * m.values.pitch + sizeof(I) * m.values.num_cols + sizeof(V) * (double)(m.num_rows + m.num_cols); }
    template <typename I> static double min(const cusp::hyb_matrix<I, V, cusp::host_memory> &m) { return min(m.ell) + min(m.coo) - sizeof(V) * (double)(m.num_rows + m.num_cols); }
};

template <typename T> double l2_error(size_t N, const T *a, const T *b) // reference utility.h:61-73
{
    double numerator = 0, denominator = 0;
    for (size_t i = 0; i < N; i++) {
        numerator += (double(a[i]) - double(b[i])) * (double(a[i]) - double(b[i]));
        denominator += double(b[i]) * double(b[i]);
    }
    return denominator > 0 ? numerator / denominator : numerator;
}

template <typename HostFormat, typename DeviceFormat, typename HostCsr>
void bench_format(const char *name, const HostCsr &host_csr)
{
    typedef typename HostCsr::value_type V;
    HostFormat host;
    try { host = host_csr; }
    catch (const cusp::format_conversion_exception &e) { std::printf("\t%-4s: Refusing to convert (%s)\n", name, e.what()); return; }
    DeviceFormat dev(host);
    const size_t M = host_csr.num_rows, N = host_csr.num_cols;
    cusp::array1d<V, cusp::host_memory> hx(N), hy(M, V(0));
    for (size_t i = 0; i < N; i++) hx[i] = V(int(i % 21) - 10);
    cusp::array1d<V, cusp::device_memory> dx(hx), dy(M, V(0));
    cusp::multiply(host_csr, hx, hy);
    cusp::multiply(dev, dx, dy);
    cusp::array1d<V, cusp::host_memory> back(dy);
    const double err = l2_error(M, back.data(), hy.data());

    void *e0, *e1;
    cusp::detail::check(cmi_event_create(&e0));
    cusp::detail::check(cmi_event_create(&e1));
    float ms = 0;
    cusp::detail::check(cmi_event_record(e0, nullptr));
    cusp::multiply(dev, dx, dy); // warm-up, timed to size the loop (benchmark.h:84-100)
    cusp::detail::check(cmi_event_record(e1, nullptr));
    cusp::detail::check(cmi_event_elapsed_ms(e0, e1, &ms));
    const double estimated = ms / 1e3;
    int iters = estimated <= 0 ? 500 : (int)std::min(500.0, std::max(100.0, 3.0 / estimated));
    cusp::detail::check(cmi_event_record(e0, nullptr));
    for (int i = 0; i < iters; i++) cusp::multiply(dev, dx, dy);
    cusp::detail::check(cmi_event_record(e1, nullptr));
    cusp::detail::check(cmi_event_elapsed_ms(e0, e1, &ms));
    const double sec = ms / 1e3 / iters;
    const double gflops = 2.0 * host_csr.num_entries / sec / 1e9;
    std::printf("\t%-4s: %9.4f ms  %8.2f GFLOP/s  %8.2f GB/s (reference byte model)  %8.2f GB/s (compulsory bytes)  [L2 error %.3e]%s\n",
                name, sec * 1e3, gflops, bytes_model<V>::ref(host) / sec / 1e9, bytes_model<V>::min(host) / sec / 1e9, err,
                err > 1e-10 ? "  *** RESULT MISMATCH ***" : "");
    FILE *fid = std::fopen("benchmark_output.log", "a"); // same log line as benchmark.h:174-179
    if (fid) { std::fprintf(fid, "kernel=%s gflops=%f gbytes=%f msec=%f\n", name, gflops, bytes_model<V>::ref(host) / sec / 1e9, sec * 1e3); std::fclose(fid); }
    cmi_event_destroy(e0);
    cmi_event_destroy(e1);
}

template <typename I, typename V> int test_all_formats(const std::string &filename)
{
    int ndev = 0;
    cmi_device_count(&ndev);
    if (ndev == 0) { std::fprintf(stderr, "ERROR: no HIP device visible (the engine has no CPU fallback)\n"); return 2; }
    char name[256]; int cus = 0; int64_t mem = 0;
    cusp::detail::check(cmi_device_info(0, name, sizeof(name), &cus, &mem));
    std::printf("Running on device 0: %s, %d CUs, %.0f GB\n\n", name, cus, mem / 1e9);

    cusp::csr_matrix<I, V, cusp::host_memory> host_matrix;
    if (filename.empty()) {
        const size_t g = args.count("grid") ? std::stoul(args["grid"]) : 512;
        const int stencil = args.count("stencil") ? std::stoi(args["stencil"]) : 5;
        if (stencil == 5) { std::printf("Generated matrix (poisson5pt %zux%zu) ", g, g); cusp::gallery::poisson5pt(host_matrix, g, g); }
        else if (stencil == 9) { std::printf("Generated matrix (poisson9pt %zux%zu) ", g, g); cusp::gallery::poisson9pt(host_matrix, g, g); }
        else if (stencil == 7) { std::printf("Generated matrix (poisson7pt %zu^3) ", g); cusp::gallery::poisson7pt(host_matrix, g, g, g); }
        else if (stencil == 27) { std::printf("Generated matrix (poisson27pt %zu^3) ", g); cusp::gallery::poisson27pt(host_matrix, g, g, g); }
        else { std::fprintf(stderr, "ERROR: --stencil must be 5, 7, 9 or 27\n"); return 1; }
    } else {
        cusp::io::read_matrix_market_file(host_matrix, filename);
        std::printf("Read matrix (%s) ", filename.c_str());
    }
    std::printf("with shape (%zu,%zu) and %zu entries (%.2f per row)\n\n", host_matrix.num_rows, host_matrix.num_cols, host_matrix.num_entries,
                double(host_matrix.num_entries) / std::max<size_t>(1, host_matrix.num_rows));
    bench_format<cusp::coo_matrix<I, V, cusp::host_memory>, cusp::coo_matrix<I, V, cusp::device_memory>>("coo", host_matrix);
    bench_format<cusp::csr_matrix<I, V, cusp::host_memory>, cusp::csr_matrix<I, V, cusp::device_memory>>("csr", host_matrix);
    bench_format<cusp::dia_matrix<I, V, cusp::host_memory>, cusp::dia_matrix<I, V, cusp::device_memory>>("dia", host_matrix);
    bench_format<cusp::ell_matrix<I, V, cusp::host_memory>, cusp::ell_matrix<I, V, cusp::device_memory>>("ell", host_matrix);
    bench_format<cusp::hyb_matrix<I, V, cusp::host_memory>, cusp::hyb_matrix<I, V, cusp::device_memory>>("hyb", host_matrix);
    return 0;
}

int main(int argc, char **argv)
{
    std::string filename = process_args(argc, argv);
    if (args.count("help")) {
        std::printf("Usage:\n\t%s\n\t%s my_matrix.mtx\n\t%s --grid=3162 --value_type=double\n", argv[0], argv[0], argv[0]);
        return 0;
    }
    const std::string value_type = args.count("value_type") ? args["value_type"] : "double";
    std::printf("\nComputing SpMV with '%s' values.\n\n", value_type.c_str());
    try {
        if (value_type == "float") return test_all_formats<int, float>(filename);
        if (value_type == "double") return test_all_formats<int, double>(filename);
    } catch (const std::exception &e) { std::fprintf(stderr, "ERROR: %s\n", e.what()); return 1; }
    std::fprintf(stderr, "ERROR: Unsupported type '%s'\n", value_type.c_str());
    return 1;
}
