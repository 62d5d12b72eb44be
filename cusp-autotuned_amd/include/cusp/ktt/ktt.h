// cusp/ktt/ktt.h -- the fork's tuning entry points, re-hosted on the OFFLINE tuning table.
//
// Reference cusp/ktt/ktt.h:35-101 + cusp/ktt/detail/ktt.inl: a process-wide KTT tuner that compiles
// kernels with NVRTC and tries one configuration per multiply call.  Here nothing is compiled at run
// time: tools/autotune.py sweeps the kernel-variant space once on an MI355X and persists the winners
// (cusp-autotuned_amd/tuned/gfx950.json); these functions only steer which table the library uses.
//   enable() / disable()   use the persisted table  /  use the built-in heuristics only
//   reset_tuning()         forget every loaded entry            (ktt.inl:130-142)
//   load(path) / save(path) read / write a table
//   multiply(A, x, y)      same as cusp::multiply (every call already runs the tuned kernel)
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../blas/blas.h"
#include "../multiply.h"
#include "ellr_matrix.h"

namespace cusp {
namespace ktt {

inline void load(const std::string &path) { cusp::detail::check(cmi_tuning_load(path.c_str())); }
inline void save(const std::string &path) { cusp::detail::check(cmi_tuning_save(path.c_str())); }
inline void reset_tuning() { cusp::detail::check(cmi_tuning_clear()); }
inline void disable() { cusp::detail::check(cmi_tuning_clear()); }
inline void enable(const std::string &table = std::string())
{
    if (!table.empty()) load(table);
    else if (std::getenv("CMI_TUNING_TABLE")) cusp::detail::check(cmi_tuning_load(nullptr));
}

template <typename Matrix, typename X, typename Y> void multiply(const Matrix &A, const X &x, Y &y) { cusp::multiply(A, x, y); }

// the configuration a multiply on this matrix runs (what KTT's best-configuration query returned)
template <typename Matrix> cmi_config selected_configuration(const Matrix &A, int format)
{
    cmi_config c;
    const int dtype = std::is_same<typename Matrix::value_type, float>::value ? CMI_F32 : CMI_F64;
    cusp::detail::check(cmi_tuning_select(format, dtype, A.num_rows, A.num_cols, A.num_entries, &c));
    return c;
}

// ---- cusp::ktt::tune --------------------------------------------------------------------------------
// The fork's tune(A, x, y, ...) (cusp/ktt/ktt.h:35-101 -> cuda/ktt/multiply.h:107-154) ran KTT's searcher
// over NVRTC-compiled variants and returned a std::vector<KernelResult>.  Here the variants are the
// library's pre-compiled kernels: tune() walks a compact configuration space for the matrix's format,
// validates every configuration against the reference result (the simplest kernel of the format, as
// testing/ktt.cu:176-196 validates against the stock multiply), times it with device events, installs the
// fastest valid one in the tuning table for this matrix's (format, dtype, row-length bucket) and returns all
// results, fastest first.  The offline tools/autotune.py does the same over many matrices and persists it.
struct tuning_result {
    cmi_config config;
    double milliseconds;
    bool valid;
};

namespace detail {
template <typename F> struct format_id;
template <> struct format_id<csr_format> { static const int value = CMI_FORMAT_CSR; };
template <> struct format_id<ell_format> { static const int value = CMI_FORMAT_ELL; };
template <> struct format_id<dia_format> { static const int value = CMI_FORMAT_DIA; };
template <> struct format_id<coo_format> { static const int value = CMI_FORMAT_COO; };

inline cmi_config make(int kernel, int block, int tpr, int rpb, int ipt, int nt, int swz, int bpc)
{
    cmi_config c = {kernel, block, tpr, rpb, ipt, nt, swz, bpc};
    return c;
}

inline std::vector<cmi_config> configuration_space(int format, double mean)
{
    std::vector<cmi_config> out;
    const int blocks[3] = {128, 256, 512};
    if (format == CMI_FORMAT_CSR) {
        out.push_back(make(CMI_CSR_SCALAR, 256, 0, 0, 0, 0, 0, 0)); // first = the validator
        for (int b : blocks)
            for (int ipt : {1, 2, 4})
                for (int nt : {0, 2})
                    for (int tpr : {0, 4, 8, 32}) {
                        if (tpr > 8 * mean && tpr) continue;
                        out.push_back(make(CMI_CSR_STREAM, b, tpr, 0, ipt, nt, 0, 0));
                        if (!tpr) out.push_back(make(CMI_CSR_STREAM, b, 0, 0, ipt, nt, 64, 0));
                    }
        if (mean <= 16) // short rows: the entry streams requested lane-strided (policy bit 4), with and without the nt load hint
            for (int b : blocks)
                for (int nt : {6, 7}) out.push_back(make(CMI_CSR_STREAM, b, 0, 0, 1, nt, 64, 0));
        if (mean <= 10) { // (nearly) equal short rows: wave-private tiles, as many entries per lane as the mean row rounded up
            const int k = mean <= 2 ? 2 : (int)(mean + 0.999);
            for (int nt : {2, 3}) out.push_back(make(CMI_CSR_STREAM_WAVE, 256, 0, 0, k, nt, 64, 0));
        }
        for (int tpr : {4, 16, 64}) out.push_back(make(CMI_CSR_VECTOR, 256, tpr, 0, 0, 0, 0, 0));
        out.push_back(make(CMI_CSR_BALANCED, 256, 0, 0, 0, 0, 0, 0)); // merge-path split: wins on skewed row lengths
        if (mean <= 40)
            for (int bpc : {3, 8}) out.push_back(make(CMI_CSR_STREAM_PIPE, 256, 0, 0, 0, 2, 0, bpc));
    } else if (format == CMI_FORMAT_ELL || format == CMI_FORMAT_DIA) {
        const int k = format == CMI_FORMAT_ELL ? CMI_ELL_ROW : CMI_DIA_ROW;
        out.push_back(make(k, 256, 0, 0, 1, 0, 0, 0));
        for (int b : {128, 256, 512, 1024})
            for (int rpl : {1, 2})
                for (int nt : {0, 2, 3}) out.push_back(make(k, b, 0, 0, rpl, nt, 0, 0));
    } else if (format == CMI_FORMAT_COO) {
        out.push_back(make(CMI_COO_SEGMENTED, 256, 0, 0, 4, 0, 0, 0));
        for (int k : {CMI_COO_SEGMENTED, CMI_COO_LANE4})
            for (int b : blocks)
                for (int ipt : {1, 2, 4, 8}) out.push_back(make(k, b, 0, 0, ipt, 1, 0, 0));
    }
    return out;
}
} // namespace detail

template <typename Matrix, typename X, typename Y>
std::vector<tuning_result> tune(const Matrix &A, const X &x, Y &y, int iterations = 20)
{
    typedef typename Matrix::value_type V;
    static_assert(std::is_same<typename Matrix::memory_space, device_memory>::value, "cusp::ktt::tune works on device_memory matrices");
    const int format = detail::format_id<typename Matrix::format>::value;
    const int dtype = std::is_same<V, float>::value ? CMI_F32 : CMI_F64;
    const double mean = A.num_rows ? double(A.num_entries) / double(A.num_rows) : 0.0;
    const std::vector<cmi_config> space = detail::configuration_space(format, mean);
    std::vector<tuning_result> results;
    cusp::array1d<V, host_memory> want, got;
    void *e0 = nullptr, *e1 = nullptr;
    cusp::detail::check(cmi_event_create(&e0));
    cusp::detail::check(cmi_event_create(&e1));
    const double tol = std::is_same<V, float>::value ? 1e-5 : 1e-6;
    double scale = 0;
    for (size_t i = 0; i < space.size(); i++) {
        tuning_result r = {space[i], 0.0, false};
        cusp::detail::forced_config() = &space[i];
        try {
            cusp::blas::fill(y, V(10));
            cusp::multiply(A, x, y);
            got = y;
            if (i == 0) {
                want = got;
                for (size_t k = 0; k < want.size(); k++) scale = std::max(scale, std::fabs(double(want[k])));
                r.valid = true;
            } else {
                r.valid = true;
                for (size_t k = 0; k < want.size() && r.valid; k++) r.valid = std::fabs(double(got[k]) - double(want[k])) <= tol * std::max(scale, 1e-300);
            }
            if (r.valid) {
                float ms = 0;
                cusp::detail::check(cmi_event_record(e0, nullptr));
                for (int it = 0; it < iterations; it++) cusp::multiply(A, x, y);
                cusp::detail::check(cmi_event_record(e1, nullptr));
                cusp::detail::check(cmi_event_elapsed_ms(e0, e1, &ms));
                r.milliseconds = ms / iterations;
            }
        } catch (const cusp::exception &) { // a configuration the launcher rejects for this shape: not a candidate
            r.valid = false;
        }
        cusp::detail::forced_config() = nullptr;
        results.push_back(r);
    }
    cmi_event_destroy(e0);
    cmi_event_destroy(e1);
    std::stable_sort(results.begin(), results.end(), [](const tuning_result &a, const tuning_result &b) {
        if (a.valid != b.valid) return a.valid;
        return a.milliseconds < b.milliseconds;
    });
    if (!results.empty() && results[0].valid) cusp::detail::check(cmi_tuning_set(format, dtype, mean, &results[0].config));
    return results;
}

} // namespace ktt
} // namespace cusp
