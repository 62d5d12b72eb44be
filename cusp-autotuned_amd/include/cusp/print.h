// cusp/print.h -- cusp::print(object[, stream]) and cusp::print_matrix (reference cusp/print.h:63-108,
// cusp/detail/print.inl:33-133): the same text the reference writes, so its examples read the same.
//   COO (and every sparse format, through a host COO copy):
//       "sparse matrix <R, C> with N entries" then one line per entry: row, column (width 14), "(value)"
//   array2d: "array2d <R, C>" then the rows;  array1d: "array1d <N>" then one value per line.
// Device containers are copied to the host first (a debugging aid, not a hot path).
#pragma once
#include <iomanip>
#include <iostream>

#include "array1d.h"
#include "array2d.h"
#include "convert.h"
#include "detail/matrices.h"

namespace cusp {

template <typename Printable, typename Stream> void print(const Printable &p, Stream &s);

namespace detail {
namespace print_detail {
template <typename T, typename Stream> void marshall(const T &val, Stream &s, bool newline = true)
{
    s << " " << std::setprecision(4) << std::setw(8) << "(" << val << ")" << (newline ? "\n" : " ");
}
} // namespace print_detail

template <typename Printable, typename Stream> void print(const Printable &p, Stream &s, cusp::coo_format)
{
    typedef typename std::remove_const<typename Printable::index_type>::type I;
    typedef typename std::remove_const<typename Printable::value_type>::type V;
    cusp::array1d<I, cusp::host_memory> ri(p.row_indices), ci(p.column_indices);
    cusp::array1d<V, cusp::host_memory> va(p.values);
    s << "sparse matrix <" << p.num_rows << ", " << p.num_cols << "> with " << p.num_entries << " entries\n";
    for (size_t n = 0; n < p.num_entries; n++) {
        s << " " << std::setw(14) << ri[n];
        s << " " << std::setw(14) << ci[n];
        print_detail::marshall(va[n], s);
    }
}

template <typename Printable, typename Stream> void print(const Printable &p, Stream &s, cusp::sparse_format)
{
    cusp::coo_matrix<typename Printable::index_type, typename Printable::value_type, cusp::host_memory> coo(p);
    cusp::print(coo, s);
}

template <typename Printable, typename Stream> void print(const Printable &p, Stream &s, cusp::array2d_format)
{
    typedef typename Printable::value_type V;
    cusp::array2d<V, cusp::host_memory, typename Printable::orientation> h(p);
    s << "array2d <" << h.num_rows << ", " << h.num_cols << ">\n";
    for (size_t i = 0; i < h.num_rows; i++) {
        for (size_t j = 0; j < h.num_cols; j++) print_detail::marshall(h(i, j), s, false);
        s << "\n";
    }
}

template <typename Printable, typename Stream> void print(const Printable &p, Stream &s, cusp::array1d_format)
{
    typedef typename std::remove_const<typename Printable::value_type>::type V;
    cusp::array1d<V, cusp::host_memory> h(p);
    s << "array1d <" << h.size() << ">\n";
    for (size_t i = 0; i < h.size(); i++) print_detail::marshall(h[i], s);
}
} // namespace detail

template <typename Printable> void print(const Printable &p) { cusp::print(p, std::cout); }
template <typename Printable, typename Stream> void print(const Printable &p, Stream &s) { cusp::detail::print(p, s, typename Printable::format()); }
template <typename Matrix> void print_matrix(const Matrix &A) { cusp::print(A); }

} // namespace cusp
