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

} // namespace ktt
} // namespace cusp
