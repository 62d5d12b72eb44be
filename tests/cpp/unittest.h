// Minimal registry + assertion macros in the spirit of the reference's home-grown framework
// (testing/unittest/testframework.h:107-149, testing/unittest/matrix.h:6-61): tests register
// themselves, the runner executes all of them and reports failures; fan-out macros instantiate a
// test template for every sparse format x value type in one memory space.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace unittest {
struct failure : std::runtime_error { using std::runtime_error::runtime_error; };
struct test { std::string name; std::function<void()> fn; };
inline std::vector<test> &registry() { static std::vector<test> r; return r; }
struct registrar { registrar(const std::string &n, std::function<void()> f) { registry().push_back({n, f}); } };

template <typename A, typename B> void assert_equal(const A &a, const B &b, const char *ea, const char *eb, const char *file, int line)
{
    if (!(a == b)) {
        std::ostringstream os;
        os << file << ":" << line << ": ASSERT_EQUAL(" << ea << ", " << eb << ") failed: " << a << " != " << b;
        throw failure(os.str());
    }
}
inline void assert_true(bool v, const char *e, const char *file, int line)
{
    if (!v) { std::ostringstream os; os << file << ":" << line << ": ASSERT(" << e << ") failed"; throw failure(os.str()); }
}

inline int run_all(int argc, char **argv)
{
    std::string filter = argc > 1 ? argv[1] : "";
    int failed = 0, ran = 0;
    for (auto &t : registry()) {
        if (!filter.empty() && t.name.find(filter) == std::string::npos) continue;
        ran++;
        try { t.fn(); }
        catch (const std::exception &e) { failed++; std::printf("[FAIL] %s\n        %s\n", t.name.c_str(), e.what()); }
    }
    std::printf("%d tests, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
} // namespace unittest

#define ASSERT_EQUAL(a, b) unittest::assert_equal((a), (b), #a, #b, __FILE__, __LINE__)
#define ASSERT_TRUE(e) unittest::assert_true((e), #e, __FILE__, __LINE__)
#define ASSERT_THROWS(expr, ex)                                                                  \
    do { bool thrown__ = false; try { expr; } catch (const ex &) { thrown__ = true; }           \
         unittest::assert_true(thrown__, #expr " throws " #ex, __FILE__, __LINE__); } while (0)
#define UT_CAT2(a, b) a##b
#define UT_CAT(a, b) UT_CAT2(a, b)
#define DECLARE_UNITTEST(fn) static unittest::registrar UT_CAT(reg_, __LINE__)(#fn, fn)
// a test template over the memory space, instantiated for TEST_SPACE
#define DECLARE_SPACE_UNITTEST(fn) static unittest::registrar UT_CAT(reg_, __LINE__)(#fn "<" TEST_SPACE_NAME ">", fn<TEST_SPACE>)
// a test template over the matrix type: {coo,csr,dia,ell,hyb} x {float,double} in TEST_SPACE
// (reference DECLARE_SPARSE_MATRIX_UNITTEST is float only; double added because the benchmark is fp64)
#define UT_ONE(fn, M, V) static unittest::registrar UT_CAT(UT_CAT(reg_##M##V##_, __LINE__), _r)(#fn "<" #M "<int," #V "," TEST_SPACE_NAME ">>", fn<cusp::M<int, V, TEST_SPACE>>)
#define DECLARE_SPARSE_MATRIX_UNITTEST(fn) \
    UT_ONE(fn, coo_matrix, float); UT_ONE(fn, csr_matrix, float); UT_ONE(fn, dia_matrix, float); UT_ONE(fn, ell_matrix, float); UT_ONE(fn, hyb_matrix, float); \
    UT_ONE(fn, coo_matrix, double); UT_ONE(fn, csr_matrix, double); UT_ONE(fn, dia_matrix, double); UT_ONE(fn, ell_matrix, double); UT_ONE(fn, hyb_matrix, double)
