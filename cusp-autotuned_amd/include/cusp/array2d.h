// cusp/array2d.h -- dense 2-D array with a pitch (reference cusp/array2d.h:93-255,
// cusp/detail/array2d_format_utils.h:93-104).  ELL and DIA keep their column indices / values in
// column_major array2d's whose pitch is round_up(num_rows, alignment): element (i,j) lives at
// j*pitch + i, so a wave reading 64 consecutive rows of one slot reads 64 consecutive addresses.
#pragma once
#include "array1d.h"

namespace cusp {

namespace detail {
template <typename Orientation> struct index_of_impl;
template <> struct index_of_impl<row_major> {
    static size_t at(size_t i, size_t j, size_t pitch) { return i * pitch + j; }
    static size_t minor(size_t rows, size_t cols) { (void)rows; return cols; }
    static size_t major(size_t rows, size_t cols) { (void)cols; return rows; }
};
template <> struct index_of_impl<column_major> {
    static size_t at(size_t i, size_t j, size_t pitch) { return j * pitch + i; }
    static size_t minor(size_t rows, size_t cols) { (void)cols; return rows; }
    static size_t major(size_t rows, size_t cols) { (void)rows; return cols; }
};
} // namespace detail

template <typename T, typename MemorySpace, typename Orientation = row_major> class array2d {
public:
    typedef T value_type;
    typedef int index_type;
    typedef MemorySpace memory_space;
    typedef array2d_format format;
    typedef Orientation orientation;
    typedef array1d<T, MemorySpace> values_array_type;
    template <typename Space> struct rebind { typedef array2d<T, Space, Orientation> type; };

    size_t num_rows, num_cols, num_entries, pitch;
    values_array_type values;

    array2d() : num_rows(0), num_cols(0), num_entries(0), pitch(0) {}
    array2d(size_t rows, size_t cols) : array2d() { resize(rows, cols); }
    array2d(size_t rows, size_t cols, const T &v) : array2d() { resize(rows, cols); fill(v); }
    array2d(size_t rows, size_t cols, const T &v, size_t pitch_) : array2d() { resize(rows, cols, pitch_); fill(v); }
    array2d(const array2d &) = default;
    array2d(array2d &&) = default;
    array2d &operator=(const array2d &) = default;
    array2d &operator=(array2d &&) = default;
    // same layout, other memory space (or element type)
    template <typename U, typename Space2>
    array2d(const array2d<U, Space2, Orientation> &o)
        : num_rows(o.num_rows), num_cols(o.num_cols), num_entries(o.num_entries), pitch(o.pitch), values(o.values) {}
    // the OTHER orientation (any memory space / element type): transposed element by element through the host
    // (reference testing/array2d.cu:229-252); set-up convenience
    template <typename U, typename Space2, typename Orientation2, typename = typename std::enable_if<!std::is_same<Orientation2, Orientation>::value>::type>
    array2d(const array2d<U, Space2, Orientation2> &o) : array2d() { assign_transposed(o); }
    template <typename U, typename Space2, typename Orientation2, typename = typename std::enable_if<!std::is_same<Orientation2, Orientation>::value>::type>
    array2d &operator=(const array2d<U, Space2, Orientation2> &o) { assign_transposed(o); return *this; }
    template <typename U, typename Space2, typename = typename std::enable_if<!std::is_same<Space2, MemorySpace>::value || !std::is_same<U, T>::value>::type>
    array2d &operator=(const array2d<U, Space2, Orientation> &o)
    {
        num_rows = o.num_rows; num_cols = o.num_cols; num_entries = o.num_entries; pitch = o.pitch;
        values = o.values;
        return *this;
    }
    // from a sparse matrix: see cusp/convert.h (array2d(const Matrix&) is defined there)
    template <typename Matrix, typename = typename Matrix::format, typename = typename std::enable_if<!std::is_same<typename Matrix::format, array2d_format>::value>::type>
    array2d(const Matrix &m);

    void resize(size_t rows, size_t cols) { resize(rows, cols, detail::index_of_impl<Orientation>::minor(rows, cols)); }
    void resize(size_t rows, size_t cols, size_t pitch_)
    {
        // reference cusp/detail/array2d.inl:43-44: pitch smaller than the minor dimension is an error
        if (pitch_ < detail::index_of_impl<Orientation>::minor(rows, cols))
            throw cusp::invalid_input_exception("array2d pitch cannot be less than minor dimension");
        num_rows = rows;
        num_cols = cols;
        num_entries = rows * cols;
        pitch = pitch_;
        values.resize(pitch_ * detail::index_of_impl<Orientation>::major(rows, cols));
    }
    void fill(const T &v)
    {
        array1d<T, host_memory> h(values.size(), v);
        values = h;
    }
    void swap(array2d &o)
    {
        std::swap(num_rows, o.num_rows); std::swap(num_cols, o.num_cols);
        std::swap(num_entries, o.num_entries); std::swap(pitch, o.pitch);
        values.swap(o.values);
    }

    size_t index_of(size_t i, size_t j) const { return detail::index_of_impl<Orientation>::at(i, j, pitch); }

private:
    template <typename Other> void assign_transposed(const Other &o)
    {
        array1d<typename Other::value_type, host_memory> src(o.values);
        resize(o.num_rows, o.num_cols);
        array1d<T, host_memory> dst(values.size(), T(0));
        for (size_t i = 0; i < num_rows; i++)
            for (size_t j = 0; j < num_cols; j++) dst[index_of(i, j)] = static_cast<T>(src[o.index_of(i, j)]);
        values = dst;
    }
public:

    // host: direct reference; device: proxy (set-up only)
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, host_memory>::value, T &>::type operator()(size_t i, size_t j) { return values[index_of(i, j)]; }
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, host_memory>::value, const T &>::type operator()(size_t i, size_t j) const { return values[index_of(i, j)]; }
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, device_memory>::value, detail::device_reference<T>>::type operator()(size_t i, size_t j)
    {
        return values[index_of(i, j)];
    }
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, device_memory>::value, T>::type operator()(size_t i, size_t j) const { return values[index_of(i, j)]; }
};

} // namespace cusp
