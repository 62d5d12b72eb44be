// cusp/copy.h -- cusp::copy(src, dst): same-format deep copy between memory spaces (reference cusp/copy.h,
// cusp/detail/copy.inl: arrays copy element-wise, matrices copy member arrays; formats must agree --
// cusp::convert is the call that changes format).
#pragma once
#include <type_traits>

#include "array1d.h"
#include "array2d.h"
#include "convert.h"
#include "exception.h"

namespace cusp {

namespace detail {
template <typename Src, typename Dst> void copy_impl(const Src &src, Dst &dst, array1d_format) { cusp::copy_array(src, dst); }
template <typename Src, typename Dst, typename Format> void copy_impl(const Src &src, Dst &dst, Format) { cusp::convert(src, dst); }
} // namespace detail

template <typename Src, typename Dst> void copy(const Src &src, Dst &dst)
{
    static_assert(std::is_same<typename Src::format, typename Dst::format>::value,
                  "cusp::copy needs source and destination of the same format (use cusp::convert to change format)");
    detail::copy_impl(src, dst, typename Dst::format());
}

} // namespace cusp
