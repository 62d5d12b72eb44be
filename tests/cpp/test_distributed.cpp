// tests/cpp/test_distributed.cpp -- the row-block sharded operator, cusp::multiply and cusp::krylov::cg through the C++ layer ONLY.
// Run as N ranks (tools/bin/cmi_launch -n N -- tests/cpp/bin/test_distributed host|device); every rank checks its own slices
// against the single-process result computed on the host from the same matrix (reference arithmetic:
// cusp/system/detail/sequential/multiply/csr_spmv.h:42-74, cusp/krylov/detail/cg.inl:41-107).
//   host    host_memory operator + vectors over the TCP star: partitions (equal rows / balanced by entries), exchange plans
//           (all-gather, all-gather of unequal pieces, halo), multiply, plain CG -- the logic RCCL carries on the GPUs, at world 2+
//           on CPUs.
//   device  device_memory: the same through cmi_comm / cmi_allgather / cmi_halo_exchange / cmi_allreduce (RCCL).  One rank per GPU --
//           or, with CMI_COMM_STAGED=1, several ranks SHARING one GPU (collectives staged through the host; the one-sided exchange
//           and its CG run for real: IPC mappings between processes, cmi_copy_ranges pulls ordered by the algorithm's reductions).
#include <cusp/csr_matrix.h>
#include <cusp/distributed.h>
#include <cusp/krylov/bicgstab.h>
#include <cusp/gallery/poisson.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

namespace cd = cusp::distributed;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "[rank %d] FAIL %s:%d  ", g_rank, __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); g_fail++; } } while (0)
static int g_rank = 0;

typedef cusp::csr_matrix<int, double, cusp::host_memory> host_csr;

// an irregular banded + scattered symmetric positive definite matrix (diagonally dominant), the same on every rank
static host_csr test_matrix(size_t n, unsigned seed, bool banded)
{
    std::mt19937 gen(seed);
    std::vector<std::vector<std::pair<int, double>>> rows(n);
    for (size_t i = 0; i < n; i++) {
        const int extra = (int)(gen() % 5);
        for (int k = 0; k < extra; k++) {
            const size_t j = banded ? std::min(n - 1, i + 1 + gen() % 9) : gen() % n;
            if (j == i) continue;
            const double v = -0.25 - (double)(gen() % 100) / 400.0;
            rows[i].push_back({(int)j, v});
            rows[j].push_back({(int)i, v});
        }
    }
    host_csr A(n, n, 0);
    std::vector<int> Aj;
    std::vector<double> Ax;
    A.row_offsets[0] = 0;
    for (size_t i = 0; i < n; i++) {
        std::sort(rows[i].begin(), rows[i].end());
        double off = 0;
        std::vector<std::pair<int, double>> merged;
        for (auto &e : rows[i]) { if (!merged.empty() && merged.back().first == e.first) merged.back().second += e.second; else merged.push_back(e); }
        for (auto &e : merged) off += std::fabs(e.second);
        bool placed = false;
        for (auto &e : merged) {
            if (!placed && e.first > (int)i) { Aj.push_back((int)i); Ax.push_back(off + 1.0); placed = true; }
            Aj.push_back(e.first); Ax.push_back(e.second);
        }
        if (!placed) { Aj.push_back((int)i); Ax.push_back(off + 1.0); }
        A.row_offsets[i + 1] = (int)Aj.size();
    }
    A.resize(n, n, Aj.size());
    // (resize keeps the offsets: same row count)
    for (size_t k = 0; k < Aj.size(); k++) { A.column_indices[k] = Aj[k]; A.values[k] = Ax[k]; }
    return A;
}

template <typename Space> static void run(cd::communicator &comm, const char *space_name)
{
    typedef cd::csr_matrix<int, double, Space> dist_csr;
    typedef cd::vector<double, Space> dist_vec;
    const int world = comm.size(), rank = comm.rank();
    struct variant { const char *name; bool banded; bool by_entries; cd::exchange_mode mode; };
    const variant variants[] = {{"scattered/equal-rows/auto", false, false, cd::exchange_mode::automatic},
                                {"scattered/by-entries/allgather", false, true, cd::exchange_mode::allgather},
                                {"banded/equal-rows/auto", true, false, cd::exchange_mode::automatic},
                                {"banded/by-entries/halo", true, true, cd::exchange_mode::halo},
                                {"banded/equal-rows/allgather", true, false, cd::exchange_mode::allgather},
                                {"banded/by-entries/peer", true, true, cd::exchange_mode::peer}};
    for (const variant &v : variants) {
        const bool one_sided = v.mode == cd::exchange_mode::peer;
        if (one_sided && (!std::is_same<Space, cusp::device_memory>::value || world == 1)) continue; // IPC mappings between processes on GPUs
        const size_t n = 4000 + 37 * (size_t)world + 1; // (not a multiple of the world size: the last slice is short)
        const host_csr G = test_matrix(n, 7u + (v.banded ? 1u : 0u), v.banded);
        const std::vector<int64_t> cuts = v.by_entries ? cd::partition_by_entries(G.row_offsets, world) : cd::partition_rows((int64_t)n, world);
        dist_csr A(comm);
        A.scatter(G, cuts, v.mode);
        CHECK(A.num_rows == n && A.num_entries == G.num_entries, "%s: global sizes %zu %zu", v.name, A.num_rows, A.num_entries);
        if (v.mode != cd::exchange_mode::automatic) CHECK(A.mode() == v.mode, "%s: mode", v.name);
        if (world > 1 && v.banded && v.mode == cd::exchange_mode::automatic)
            CHECK(A.mode() == (std::is_same<Space, cusp::device_memory>::value ? cd::exchange_mode::peer : cd::exchange_mode::halo),
                  "%s: a banded matrix should pick the halo exchange (one-sided on device_memory)", v.name);
        if (world > 1 && !v.banded && v.mode == cd::exchange_mode::automatic) CHECK(A.mode() == cd::exchange_mode::allgather, "%s: scattered columns should pick the all-gather", v.name);
        // two-sided halo mode on device_memory: the interior rows are multiplied on a side stream while the halo is in flight
        if (world > 1 && v.mode == cd::exchange_mode::halo && std::is_same<Space, cusp::device_memory>::value)
            CHECK(A.overlapped() && A.interior_last() - A.interior_first() > A.local_rows() / 2 && A.interior_last() - A.interior_first() < A.local_rows(),
                  "%s: banded block of %zu rows: interior rows [%zu, %zu) overlapped %d", v.name, A.local_rows(), A.interior_first(), A.interior_last(), (int)A.overlapped());
        if (v.mode != cd::exchange_mode::halo || !std::is_same<Space, cusp::device_memory>::value) CHECK(!A.overlapped(), "%s: overlap outside the two-sided halo mode", v.name);
        // x, and the single-process y = G x on the host
        cusp::array1d<double, cusp::host_memory> xg(n), yg(n);
        for (size_t i = 0; i < n; i++) xg[i] = double((unsigned(i) * 2654435761u) % 1000u) / 997.0 - 0.5;
        cusp::multiply(G, xg, yg);
        const size_t lo = A.row_begin(), hi = A.row_end();
        cusp::array1d<double, cusp::host_memory> xl(xg.begin() + lo, xg.begin() + hi);
        dist_vec x = A.make_vector(), y = A.make_vector(9.0);
        { auto xv = x.local(); cusp::copy_array(xl, xv); }
        cusp::multiply(A, x, y);
        cusp::array1d<double, cusp::host_memory> got(y.local());
        bool same = got.size() == hi - lo;
        for (size_t i = 0; same && i < got.size(); i++) same = got[i] == yg[lo + i]; // storage-order sums: the host loop's bits
        CHECK(same, "%s [%s]: sharded multiply differs from the single-process result on rows [%zu, %zu)", v.name, space_name, lo, hi);
        // reductions see the whole vector
        double ref_dot = 0;
        for (size_t i = 0; i < n; i++) ref_dot += xg[i] * yg[i];
        const double d = cusp::blas::dot(x, y);
        CHECK(std::fabs(d - ref_dot) <= 1e-12 * std::fabs(ref_dot) + 1e-9, "%s: dot %.17g vs %.17g", v.name, d, ref_dot);
        // the gathered result is the global vector on every rank
        const auto all = y.gather();
        bool gsame = all.size() == n;
        for (size_t i = 0; gsame && i < n; i++) gsame = all[i] == yg[i];
        CHECK(gsame, "%s: gathered y differs", v.name);
        // CG: same iteration count and residual history (to rounding) as the single-process solve
        cusp::array1d<double, cusp::host_memory> bg(n), sg(n, 0.0);
        for (size_t i = 0; i < n; i++) bg[i] = 1.0 + double(i % 7);
        cusp::monitor<double> mon_ref(bg, 200, 1e-10);
        cusp::krylov::cg(G, sg, bg, mon_ref);
        dist_vec bl = A.make_vector(), sol = A.make_vector(0.0);
        { cusp::array1d<double, cusp::host_memory> t(bg.begin() + lo, bg.begin() + hi); auto bv = bl.local(); cusp::copy_array(t, bv); }
        cusp::monitor<double> mon(bl, 200, 1e-10);
        cusp::krylov::cg(A, sol, bl, mon);
        CHECK(mon.converged() && mon_ref.converged(), "%s: CG did not converge (%zu / %zu iterations)", v.name, mon.iteration_count(), mon_ref.iteration_count());
        CHECK(mon.iteration_count() == mon_ref.iteration_count(), "%s [%s]: %zu iterations sharded, %zu single-process", v.name, space_name, mon.iteration_count(), mon_ref.iteration_count());
        const size_t k = std::min(mon.residuals.size(), mon_ref.residuals.size());
        for (size_t i = 0; i < k; i++)
            if (std::fabs(mon.residuals[i] - mon_ref.residuals[i]) > 1e-8 * mon_ref.residuals[0]) { CHECK(false, "%s: residual %zu: %.12e vs %.12e", v.name, i, mon.residuals[i], mon_ref.residuals[i]); break; }
        cusp::array1d<double, cusp::host_memory> sl(sol.local());
        double err = 0;
        for (size_t i = 0; i < sl.size(); i++) err = std::max(err, std::fabs(sl[i] - sg[lo + i]));
        CHECK(err <= 1e-8, "%s: solution differs by %.3e", v.name, err);
        if (rank == 0) std::printf("ok  %-34s [%s, world %d]  mode %s, %lld values per exchange (all-gather %lld), CG %zu iterations\n", v.name, space_name, world,
                                   A.mode_name(), (long long)A.exchange_values(), (long long)A.allgather_values(), mon.iteration_count());
    }
    // bicgstab on a sharded NON-symmetric operator (cusp/distributed/bicgstab.h): the banded matrix with its strictly-upper entries scaled by 0.4 (still
    // strictly diagonally dominant); the single-process host solve is the reference: same iteration count to within rounding, same solution
    {
        const size_t n = 3000 + 13 * (size_t)world;
        host_csr G = test_matrix(n, 21u, true);
        for (size_t i = 0; i < n; i++)
            for (int jj = G.row_offsets[i]; jj < G.row_offsets[i + 1]; jj++)
                if ((size_t)G.column_indices[jj] > i) G.values[jj] *= 0.4;
        const std::vector<int64_t> cuts = cd::partition_by_entries(G.row_offsets, world);
        dist_csr A(comm);
        A.scatter(G, cuts, cd::exchange_mode::automatic);
        cusp::array1d<double, cusp::host_memory> bg(n), sg(n, 0.0);
        for (size_t i = 0; i < n; i++) bg[i] = 1.0 + double(i % 5);
        cusp::monitor<double> mon_ref(bg, 300, 1e-10);
        cusp::krylov::bicgstab(G, sg, bg, mon_ref);
        const size_t lo = A.row_begin(), hi = A.row_end();
        dist_vec bl = A.make_vector(), sol = A.make_vector(0.0);
        { cusp::array1d<double, cusp::host_memory> t(bg.begin() + lo, bg.begin() + hi); auto bv = bl.local(); cusp::copy_array(t, bv); }
        cusp::monitor<double> mon(bl, 300, 1e-10);
        cusp::krylov::bicgstab(A, sol, bl, mon);
        CHECK(mon.converged() && mon_ref.converged(), "sharded bicgstab did not converge (%zu / %zu iterations)", mon.iteration_count(), mon_ref.iteration_count());
        const long long d = (long long)mon.iteration_count() - (long long)mon_ref.iteration_count();
        CHECK(d >= -2 && d <= 2, "sharded bicgstab: %zu iterations, single-process %zu", mon.iteration_count(), mon_ref.iteration_count());
        cusp::array1d<double, cusp::host_memory> sl(sol.local());
        double err = 0;
        for (size_t i = 0; i < sl.size(); i++) err = std::max(err, std::fabs(sl[i] - sg[lo + i]));
        CHECK(err <= 1e-8, "sharded bicgstab: solution differs by %.3e", err);
        if (rank == 0) std::printf("ok  sharded bicgstab, non-symmetric banded [%s, world %d]  mode %s, %zu iterations (single process %zu)\n", space_name, world, A.mode_name(),
                                   mon.iteration_count(), mon_ref.iteration_count());
    }
    // the gallery builder: every rank its own rows of poisson5pt(m, n); against the host gallery
    {
        const size_t m = 61, nn = 47;
        dist_csr A(comm);
        cd::poisson5pt(A, m, nn);
        host_csr G;
        cusp::gallery::poisson5pt(G, m, nn);
        cusp::csr_matrix<int, double, cusp::host_memory> Lh(A.local);
        const size_t lo = A.row_begin();
        bool same = Lh.num_rows == A.local_rows() && A.num_entries == G.num_entries;
        for (size_t i = 0; same && i < Lh.num_rows; i++) {
            const int a = Lh.row_offsets[i], b = Lh.row_offsets[i + 1], ga = G.row_offsets[lo + i], gb = G.row_offsets[lo + i + 1];
            same = (b - a) == (gb - ga);
            for (int k = 0; same && k < b - a; k++) same = Lh.column_indices[a + k] == G.column_indices[ga + k] && Lh.values[a + k] == G.values[ga + k];
        }
        CHECK(same, "sharded poisson5pt(%zu, %zu) differs from the gallery's rows", m, nn);
        // the quickstart protocol (docs/quickstart.md:72-87 shape): b = 1, x0 = 0
        dist_vec b = A.make_vector(1.0), x = A.make_vector(0.0);
        cusp::monitor<double> mon(b, 500, 1e-8);
        cusp::krylov::cg(A, x, b, mon);
        cusp::array1d<double, cusp::host_memory> bg(m * nn, 1.0), xg(m * nn, 0.0);
        cusp::monitor<double> mon_ref(bg, 500, 1e-8);
        cusp::krylov::cg(G, xg, bg, mon_ref);
        CHECK(mon.converged() && mon.iteration_count() == mon_ref.iteration_count(), "poisson CG: %zu vs %zu iterations", mon.iteration_count(), mon_ref.iteration_count());
        if (rank == 0) std::printf("ok  sharded poisson5pt(%zu,%zu) + cg  [%s, world %d]  mode %s, %zu iterations, ||r|| = %.6e\n", m, nn, space_name, world,
                                   A.mode_name(), mon.iteration_count(), (double)mon.residual_norm());
    }
}

// the other formats (cusp/distributed/matrix.h, round 4): the CSR operator's partition + exchange with the rank's block converted into ELL /
// DIA / COO / HYB -- multiply against the single-process host result (one chain per row in every one of these paths: the host loop's bits
// on host_memory; on device_memory the formats' own parity classes), CG against the single-process solve
template <typename Space> static void run_formats(cd::communicator &comm, const char *space_name)
{
    const int world = comm.size(), rank = comm.rank();
    const size_t n = 3000 + 29 * (size_t)world + 1;
    const host_csr G = test_matrix(n, 11u, true); // banded: a handful of diagonals (DIA holds it), a halo exchange
    const std::vector<int64_t> cuts = cd::partition_by_entries(G.row_offsets, world);
    cd::csr_matrix<int, double, Space> A(comm);
    A.scatter(G, cuts, cd::exchange_mode::automatic);
    cusp::array1d<double, cusp::host_memory> xg(n), yg(n);
    for (size_t i = 0; i < n; i++) xg[i] = double((unsigned(i) * 2654435761u) % 1000u) / 997.0 - 0.5;
    cusp::multiply(G, xg, yg);
    const size_t lo = A.row_begin(), hi = A.row_end();
    cusp::array1d<double, cusp::host_memory> bg(n), sg(n, 0.0);
    for (size_t i = 0; i < n; i++) bg[i] = 1.0 + double(i % 7);
    cusp::monitor<double> mon_ref(bg, 200, 1e-10);
    cusp::krylov::cg(G, sg, bg, mon_ref);
    auto check = [&](const char *name, const auto &E) {
        CHECK(E.num_rows == n && E.local_rows() == hi - lo && E.local.num_rows == hi - lo && E.local.num_cols == n, "%s: block shape", name);
        auto x = E.make_vector(), y = E.make_vector(9.0);
        { cusp::array1d<double, cusp::host_memory> t(xg.begin() + lo, xg.begin() + hi); auto xv = x.local(); cusp::copy_array(t, xv); }
        cusp::multiply(E, x, y);
        cusp::array1d<double, cusp::host_memory> got(y.local());
        double worst = 0, scale = 0;
        bool same = got.size() == hi - lo;
        for (size_t i = 0; i < got.size(); i++) { same = same && got[i] == yg[lo + i]; worst = std::max(worst, std::fabs(got[i] - yg[lo + i])); scale = std::max(scale, std::fabs(yg[lo + i])); }
        if (std::is_same<Space, cusp::host_memory>::value) CHECK(same, "%s [%s]: sharded multiply differs from the single-process result", name, space_name);
        else CHECK(worst <= 1e-12 * scale, "%s [%s]: sharded multiply off by %.3e", name, space_name, worst);
        auto bl = E.make_vector(), sol = E.make_vector(0.0);
        { cusp::array1d<double, cusp::host_memory> t(bg.begin() + lo, bg.begin() + hi); auto bv = bl.local(); cusp::copy_array(t, bv); }
        cusp::monitor<double> mon(bl, 200, 1e-10);
        cusp::krylov::cg(E, sol, bl, mon);
        const long long d = (long long)mon.iteration_count() - (long long)mon_ref.iteration_count();
        CHECK(mon.converged() && d >= -1 && d <= 1, "%s [%s]: CG %zu iterations sharded, %zu single-process", name, space_name, mon.iteration_count(), mon_ref.iteration_count());
        cusp::array1d<double, cusp::host_memory> sl(sol.local());
        double err = 0;
        for (size_t i = 0; i < sl.size(); i++) err = std::max(err, std::fabs(sl[i] - sg[lo + i]));
        CHECK(err <= 1e-8, "%s: solution differs by %.3e", name, err);
        if (rank == 0) std::printf("ok  sharded %-4s [%s, world %d]  mode %s, multiply %s, CG %zu iterations\n", name, space_name, world, E.mode_name(), same ? "bit-identical" : "within 1e-12", mon.iteration_count());
    };
    { cd::ell_matrix<int, double, Space> E(A); check("ell", E); }
    { cd::coo_matrix<int, double, Space> E(A); check("coo", E); }
    { cd::dia_matrix<int, double, Space> E(A); check("dia", E); }
    { cd::hyb_matrix<int, double, Space> E(A); check("hyb", E); }
}

// float operators and vectors: the f32 instances of the sharded multiply and of the (fused, on device) CG -- scalars stay doubles
template <typename Space> static void run_float(cd::communicator &comm, const char *space_name)
{
    const size_t m = 53, nn = 41;
    cd::csr_matrix<int, float, Space> A(comm);
    cd::poisson5pt(A, m, nn);
    cusp::csr_matrix<int, float, cusp::host_memory> G;
    cusp::gallery::poisson5pt(G, m, nn);
    cusp::array1d<float, cusp::host_memory> xg(m * nn), yg(m * nn);
    for (size_t i = 0; i < xg.size(); i++) xg[i] = float((unsigned(i) * 2654435761u) % 1000u) / 997.0f - 0.5f;
    cusp::multiply(G, xg, yg);
    auto x = A.make_vector(), y = A.make_vector(7.0f);
    { cusp::array1d<float, cusp::host_memory> t(xg.begin() + A.row_begin(), xg.begin() + A.row_end()); auto xv = x.local(); cusp::copy_array(t, xv); }
    cusp::multiply(A, x, y);
    cusp::array1d<float, cusp::host_memory> got(y.local());
    bool same = true;
    for (size_t i = 0; same && i < got.size(); i++) same = got[i] == yg[A.row_begin() + i];
    CHECK(same, "float sharded multiply differs [%s]", space_name);
    auto b = A.make_vector(1.0f), sol = A.make_vector(0.0f);
    cusp::monitor<float> mon(b, 300, 1e-4f);
    cusp::krylov::cg(A, sol, b, mon);
    cusp::array1d<float, cusp::host_memory> bg(m * nn, 1.0f), sg(m * nn, 0.0f);
    cusp::monitor<float> mon_ref(bg, 300, 1e-4f);
    cusp::krylov::cg(G, sg, bg, mon_ref);
    const long di = (long)mon.iteration_count() - (long)mon_ref.iteration_count();
    CHECK(mon.converged() && di >= -2 && di <= 2, "float CG: %zu iterations sharded, %zu single-process", mon.iteration_count(), mon_ref.iteration_count());
    if (comm.rank() == 0) std::printf("ok  float: sharded poisson5pt(%zu,%zu) multiply + cg [%s, world %d]  mode %s, %zu iterations\n", m, nn, space_name, comm.size(), A.mode_name(), mon.iteration_count());
}

int main(int argc, char **argv)
{
    const bool device = argc > 1 && !std::strcmp(argv[1], "device");
    try {
        auto comm = cd::communicator::from_environment(device);
        g_rank = comm->rank();
        if (device) {
            run<cusp::device_memory>(*comm, "device_memory");
            run_formats<cusp::device_memory>(*comm, "device_memory");
            run_float<cusp::device_memory>(*comm, "device_memory");
            int v = 0;
            cusp::detail::check(cmi_comm_library_version(&v));
            if (g_rank == 0) std::printf("RCCL version code %d, world %d\n", v, comm->size());
        } else {
            run<cusp::host_memory>(*comm, "host_memory");
            run_formats<cusp::host_memory>(*comm, "host_memory");
            run_float<cusp::host_memory>(*comm, "host_memory");
        }
        comm->barrier(cusp::host_memory());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[rank %d] EXCEPTION: %s\n", g_rank, e.what());
        return 1;
    }
    if (g_fail) { std::fprintf(stderr, "[rank %d] %d check(s) failed\n", g_rank, g_fail); return 1; }
    return 0;
}
