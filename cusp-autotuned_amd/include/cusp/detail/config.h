// cusp/detail/config.h -- glue between the header-only cusp:: layer and the C-ABI library
// (include/cusp_mi355x.h).  Every C-ABI status becomes the cusp exception the reference would throw
// for the same condition (cusp/exception.h); unlike the reference, kernel-launch failures are reported.
#pragma once
#include <cstdlib>
#include <new>
#include <string>

#include "../../../../include/cusp_mi355x.h"
#include "../exception.h"

#define CUSP_MI355X 1
#define CUSP_VERSION 500 /* API level of the reference this layer mirrors: CUSP v0.5.x */

namespace cusp {
namespace detail {

inline void check(int status)
{
    if (status == CMI_SUCCESS) return;
    const std::string msg = std::string(cmi_status_string(status)) + ": " + cmi_last_error();
    switch (status) {
    case CMI_ERROR_INVALID_VALUE: throw cusp::invalid_input_exception(msg);
    case CMI_ERROR_NOT_SUPPORTED: throw cusp::not_implemented_exception(msg);
    case CMI_ERROR_ALLOC: throw std::bad_alloc();
    case CMI_ERROR_IO: throw cusp::io_exception(msg);
    default: throw cusp::runtime_exception(msg);
    }
}

// cusp::detail::round_up (reference cusp/detail/utils.h:24-28)
template <typename I> inline I round_up(I n, I k) { return k * ((n + k - 1) / k); }

} // namespace detail
} // namespace cusp
