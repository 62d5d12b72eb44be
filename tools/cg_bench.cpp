// tools/cg_bench.cpp -- CG iterations/s through the C++ layer (cusp::krylov::cg on device_memory), the
// caller of the SpMV hot path: poisson5pt(grid, grid), b = deterministic x pattern, fixed iteration count.
//   cg_bench [--grid=3162] [--iterations=200]
// Prints the fused device path (default: identity preconditioner, double) and, for comparison, the plain
// operation-by-operation path (forced by passing an explicit non-identity-typed preconditioner).
#include <cusp/csr_matrix.h>
#include <cusp/gallery/poisson.h>
#include <cusp/krylov/cg.h>
#include <cusp/monitor.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

// same action as identity_operator, different type: keeps cusp::krylov::cg on its plain path
struct copy_preconditioner {
    template <typename X, typename Y> void operator()(const X &x, Y &y) const { cusp::blas::copy(x, y); }
};

int main(int argc, char **argv)
{
    size_t grid = 3162, iters = 200;
    for (int i = 1; i < argc; i++) {
        if (!std::strncmp(argv[i], "--grid=", 7)) grid = std::strtoul(argv[i] + 7, nullptr, 10);
        if (!std::strncmp(argv[i], "--iterations=", 13)) iters = std::strtoul(argv[i] + 13, nullptr, 10);
    }
    try {
        cusp::csr_matrix<int, double, cusp::device_memory> A;
        cusp::gallery::poisson5pt(A, grid, grid);
        const size_t N = A.num_rows;
        cusp::array1d<double, cusp::host_memory> hb(N);
        for (size_t i = 0; i < N; i++) hb[i] = double((unsigned(i) * 2654435761u) % 1000u) / 997.0 - 0.5;
        cusp::array1d<double, cusp::device_memory> b(hb);
        for (int pass = 0; pass < 4; pass++) {
            const bool fused = pass & 1;
            cusp::array1d<double, cusp::device_memory> x(N, 0.0);
            cusp::monitor<double> monitor(b, iters, 0.0, 0.0);
            cusp::detail::check(cmi_device_synchronize());
            const auto t0 = std::chrono::steady_clock::now();
            if (fused) cusp::krylov::cg(A, x, b, monitor);
            else { copy_preconditioner M; cusp::krylov::cg(A, x, b, monitor, M); }
            cusp::detail::check(cmi_device_synchronize());
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%-5s  %zu iterations in %8.1f ms = %7.0f it/s, %7.1f us/iteration; final ||r|| = %.6e\n", fused ? "fused" : "plain",
                        monitor.iteration_count(), sec * 1e3, monitor.iteration_count() / sec, sec / monitor.iteration_count() * 1e6, monitor.residual_norm());
        }
    } catch (const std::exception &e) { std::fprintf(stderr, "ERROR: %s\n", e.what()); return 1; }
    return 0;
}
