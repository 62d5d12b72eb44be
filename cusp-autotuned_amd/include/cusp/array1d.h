// cusp/array1d.h -- cusp::array1d<T, MemorySpace> and cusp::array1d_view (reference
// cusp/array1d.h:98-242,361).  The reference derives from thrust::detail::vector_base; here
//   host_memory   -> std::vector<T>
//   device_memory -> an RAII buffer in HBM (cmi_malloc / cmi_free) with H<->D copies on converting
//                    construction / assignment, exactly where the reference's containers copy.
// Element access on a device array goes through a proxy (one 4/8-byte copy per access), like
// thrust::device_reference: fine for tests and set-up code, never used on the hot path.
#pragma once
#include <algorithm>
#include <cstddef>
#include <utility>
#include <iterator>
#include <vector>

#include "format.h"
#include "memory.h"

namespace cusp {

template <typename T, typename MemorySpace> class array1d;
template <typename T, typename MemorySpace> class array1d_view;

namespace detail {

// read / write one element of device memory (set-up convenience, never on the hot path)
template <typename T> class device_reference {
public:
    explicit device_reference(T *p) : ptr(p) {}
    operator T() const
    {
        T v;
        check(cmi_memcpy_d2h(&v, ptr, sizeof(T), nullptr));
        return v;
    }
    device_reference &operator=(const T &v)
    {
        check(cmi_memcpy_h2d(ptr, &v, sizeof(T), nullptr));
        return *this;
    }
    device_reference &operator=(const device_reference &o) { return *this = static_cast<T>(o); }
private:
    T *ptr;
};

template <typename Space> struct copier;
template <> struct copier<host_memory> {
    template <typename T> static void from_host(T *dst, const T *src, size_t n) { std::copy(src, src + n, dst); }
    template <typename T> static void from_device(T *dst, const T *src, size_t n) { check(cmi_memcpy_d2h(dst, src, n * sizeof(T), nullptr)); }
};
template <> struct copier<device_memory> {
    template <typename T> static void from_host(T *dst, const T *src, size_t n) { check(cmi_memcpy_h2d(dst, src, n * sizeof(T), nullptr)); }
    template <typename T> static void from_device(T *dst, const T *src, size_t n)
    {
        check(cmi_memcpy_d2d(dst, src, n * sizeof(T), nullptr));
        check(cmi_stream_synchronize(nullptr));
    }
};

// dst[0..n) = src[0..n) across any pair of memory spaces
template <typename T, typename DstSpace, typename SrcSpace> struct raw_copy;
template <typename T, typename DstSpace> struct raw_copy<T, DstSpace, host_memory> {
    static void run(T *dst, const T *src, size_t n) { if (n) copier<DstSpace>::from_host(dst, src, n); }
};
template <typename T, typename DstSpace> struct raw_copy<T, DstSpace, device_memory> {
    static void run(T *dst, const T *src, size_t n) { if (n) copier<DstSpace>::from_device(dst, src, n); }
};

} // namespace detail

// ---------------------------------------------------------------------------------------------
// host array
// ---------------------------------------------------------------------------------------------
template <typename T> class array1d<T, host_memory> : public std::vector<T> {
    typedef std::vector<T> Parent;
public:
    typedef T value_type;
    typedef host_memory memory_space;
    typedef array1d_format format;
    typedef array1d_view<T, host_memory> view;
    typedef array1d_view<const T, host_memory> const_view;
    template <typename Space> struct rebind { typedef array1d<T, Space> type; };

    array1d() {}
    explicit array1d(size_t n) : Parent(n) {}
    array1d(size_t n, const T &v) : Parent(n, v) {}
    array1d(const array1d &o) : Parent(o) {}
    array1d(array1d &&o) noexcept : Parent(std::move(o)) {}
    array1d(const std::vector<T> &v) : Parent(v) {}
    template <typename It> array1d(It first, It last) : Parent(first, last) {}
    // converting construction from any array-like with data()/size()/memory_space
    template <typename Other, typename = typename Other::memory_space, typename = typename Other::value_type>
    array1d(const Other &o) : Parent(o.size()) { assign_from(o); }

    array1d &operator=(const array1d &o) { Parent::operator=(o); return *this; }
    array1d &operator=(array1d &&o) noexcept { Parent::operator=(std::move(o)); return *this; }
    template <typename Other, typename = typename Other::memory_space, typename = typename Other::value_type>
    array1d &operator=(const Other &o) { this->resize(o.size()); assign_from(o); return *this; }

    T *data() { return Parent::data(); }
    const T *data() const { return Parent::data(); }
    view subarray(size_t start, size_t n) { return view(data() + start, n); }
    const_view subarray(size_t start, size_t n) const { return const_view(data() + start, n); }

private:
    template <typename Other> void assign_from(const Other &o)
    {
        typedef typename Other::value_type U;
        if (std::is_same<typename std::remove_const<U>::type, T>::value) {
            detail::raw_copy<T, host_memory, typename Other::memory_space>::run(this->data(), reinterpret_cast<const T *>(o.data()), o.size());
        } else { // element type conversion goes through a host copy of the source
            array1d<typename std::remove_const<U>::type, host_memory> tmp(o);
            for (size_t i = 0; i < tmp.size(); i++) (*this)[i] = static_cast<T>(tmp[i]);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// device array
// ---------------------------------------------------------------------------------------------
template <typename T> class array1d<T, device_memory> {
public:
    typedef T value_type;
    typedef device_memory memory_space;
    typedef array1d_format format;
    typedef T *iterator;
    typedef const T *const_iterator;
    typedef array1d_view<T, device_memory> view;
    typedef array1d_view<const T, device_memory> const_view;
    template <typename Space> struct rebind { typedef array1d<T, Space> type; };

    array1d() : ptr_(nullptr), size_(0), capacity_(0) {}
    explicit array1d(size_t n) : array1d() { resize(n); }
    array1d(size_t n, const T &v) : array1d() { resize(n, v); }
    array1d(const array1d &o) : array1d() { assign_from(o); }
    array1d(array1d &&o) noexcept : ptr_(o.ptr_), size_(o.size_), capacity_(o.capacity_) { o.ptr_ = nullptr; o.size_ = o.capacity_ = 0; }
    array1d(const std::vector<T> &v) : array1d() { resize(v.size()); detail::raw_copy<T, device_memory, host_memory>::run(ptr_, v.data(), v.size()); }
    // a range of HOST iterators (device iterators are plain pointers here and cannot be told apart: construct from
    // the array or a view instead)
    template <typename It, typename = typename std::iterator_traits<It>::iterator_category,
              typename = typename std::enable_if<!std::is_pointer<It>::value>::type>
    array1d(It first, It last) : array1d(std::vector<T>(first, last)) {}
    template <typename Other, typename = typename Other::memory_space, typename = typename Other::value_type>
    array1d(const Other &o) : array1d() { assign_from(o); }
    ~array1d() { release(); }

    array1d &operator=(const array1d &o) { if (this != &o) assign_from(o); return *this; }
    array1d &operator=(array1d &&o) noexcept { swap(o); return *this; }
    template <typename Other, typename = typename Other::memory_space, typename = typename Other::value_type>
    array1d &operator=(const Other &o) { assign_from(o); return *this; }

    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    T *data() { return ptr_; }
    const T *data() const { return ptr_; }
    iterator begin() { return ptr_; }
    iterator end() { return ptr_ + size_; }
    const_iterator begin() const { return ptr_; }
    const_iterator end() const { return ptr_ + size_; }

    detail::device_reference<T> operator[](size_t i) { return detail::device_reference<T>(ptr_ + i); }
    T operator[](size_t i) const { return static_cast<T>(detail::device_reference<T>(ptr_ + i)); }

    void resize(size_t n)
    {
        if (n > capacity_) {
            T *p = nullptr;
            detail::check(cmi_malloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
            if (size_) detail::raw_copy<T, device_memory, device_memory>::run(p, ptr_, size_);
            release();
            ptr_ = p;
            capacity_ = n;
        }
        size_ = n;
    }
    void resize(size_t n, const T &v)
    {
        const size_t old = size_;
        resize(n);
        if (n > old) { // new elements take the fill value (set-up path: staged through the host)
            std::vector<T> fill(n - old, v);
            detail::raw_copy<T, device_memory, host_memory>::run(ptr_ + old, fill.data(), n - old);
        }
    }
    void clear() { size_ = 0; }
    void swap(array1d &o) { std::swap(ptr_, o.ptr_); std::swap(size_, o.size_); std::swap(capacity_, o.capacity_); }
    void reserve(size_t n)
    {
        if (n <= capacity_) return;
        const size_t keep = size_;
        resize(n);
        size_ = keep;
    }
    void push_back(const T &v) // set-up convenience (reference testing/array1d.cu:7-27): amortised doubling, one 1-element copy
    {
        if (size_ == capacity_) reserve(capacity_ ? 2 * capacity_ : 4);
        detail::raw_copy<T, device_memory, host_memory>::run(ptr_ + size_, &v, 1);
        size_++;
    }
    view subarray(size_t start, size_t n) { return view(ptr_ + start, n); }
    const_view subarray(size_t start, size_t n) const { return const_view(ptr_ + start, n); }

private:
    template <typename Other> void assign_from(const Other &o)
    {
        typedef typename std::remove_const<typename Other::value_type>::type U;
        resize(o.size());
        if (std::is_same<U, T>::value) {
            detail::raw_copy<T, device_memory, typename Other::memory_space>::run(ptr_, reinterpret_cast<const T *>(o.data()), o.size());
        } else {
            array1d<U, host_memory> src(o);
            std::vector<T> tmp(src.begin(), src.end());
            detail::raw_copy<T, device_memory, host_memory>::run(ptr_, tmp.data(), tmp.size());
        }
    }
    void release()
    {
        if (ptr_) cmi_free(ptr_); // destructor path: never throws
        ptr_ = nullptr;
        capacity_ = 0;
    }
    T *ptr_;
    size_t size_, capacity_;
};

// ---------------------------------------------------------------------------------------------
// non-owning view (reference cusp/array1d.h:361, make_array1d_view)
// ---------------------------------------------------------------------------------------------
template <typename T, typename MemorySpace> class array1d_view {
public:
    typedef typename std::remove_const<T>::type value_type;
    typedef MemorySpace memory_space;
    typedef array1d_format format;
    typedef array1d_view view;

    array1d_view() : ptr_(nullptr), size_(0) {}
    array1d_view(T *p, size_t n) : ptr_(p), size_(n) {}
    array1d_view(const array1d_view &) = default;
    template <typename U> array1d_view(array1d<U, MemorySpace> &a) : ptr_(a.data()), size_(a.size()) {}
    template <typename U> array1d_view(const array1d<U, MemorySpace> &a) : ptr_(a.data()), size_(a.size()) {}

    size_t size() const { return size_; }
    T *data() const { return ptr_; }
    T *begin() const { return ptr_; }
    T *end() const { return ptr_ + size_; }
    void resize(size_t n)
    {
        if (n > size_) throw cusp::invalid_input_exception("array1d_view cannot grow beyond the storage it views");
        size_ = n;
    }
    array1d_view subarray(size_t start, size_t n) const { return array1d_view(ptr_ + start, n); }

    // host views index directly, device views through the proxy
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, host_memory>::value, T &>::type operator[](size_t i) const { return ptr_[i]; }
    template <typename S = MemorySpace>
    typename std::enable_if<std::is_same<S, device_memory>::value, detail::device_reference<T>>::type operator[](size_t i) const
    {
        return detail::device_reference<T>(ptr_ + i);
    }

private:
    T *ptr_;
    size_t size_;
};

template <typename T, typename M> array1d_view<T, M> make_array1d_view(array1d<T, M> &a) { return array1d_view<T, M>(a); }
template <typename T, typename M> array1d_view<const T, M> make_array1d_view(const array1d<T, M> &a) { return array1d_view<const T, M>(a); }
template <typename T, typename M> array1d_view<T, M> make_array1d_view(const array1d_view<T, M> &v) { return v; }

// equality across memory spaces and with std::vector (reference testing/array1d.cu:165-193): element-wise on
// host copies -- a test / set-up convenience, not a hot path
namespace detail {
template <typename A> std::vector<typename std::remove_const<typename A::value_type>::type> host_copy(const A &a)
{
    typedef typename std::remove_const<typename A::value_type>::type U;
    std::vector<U> out(a.size());
    raw_copy<U, host_memory, typename A::memory_space>::run(out.data(), const_cast<const U *>(a.data()), a.size());
    return out;
}
template <typename T> const std::vector<T> &host_copy(const std::vector<T> &v) { return v; }
template <typename X, typename = void> struct is_array_like : std::false_type {};
template <typename X> struct is_array_like<X, typename std::enable_if<std::is_same<typename X::format, array1d_format>::value>::type> : std::true_type {};
template <typename T, typename A> struct is_array_like<std::vector<T, A>, void> : std::true_type {};
template <typename A, typename B> bool arrays_equal_any(const A &a, const B &b)
{
    if (a.size() != b.size()) return false;
    const auto ha = host_copy(a);
    const auto hb = host_copy(b);
    for (size_t i = 0; i < ha.size(); i++)
        if (!(ha[i] == hb[i])) return false;
    return true;
}
} // namespace detail
template <typename T, typename M, typename B, typename = typename std::enable_if<detail::is_array_like<B>::value>::type>
bool operator==(const array1d<T, M> &a, const B &b) { return detail::arrays_equal_any(a, b); }
template <typename T, typename M, typename B, typename = typename std::enable_if<detail::is_array_like<B>::value>::type>
bool operator!=(const array1d<T, M> &a, const B &b) { return !detail::arrays_equal_any(a, b); }

// cusp::copy between any two array-likes (reference cusp/copy.h); sizes must already match or dst resizes
template <typename Src, typename Dst> void copy_array(const Src &src, Dst &dst)
{
    typedef typename Dst::value_type T;
    dst.resize(src.size());
    detail::raw_copy<T, typename Dst::memory_space, typename Src::memory_space>::run(dst.data(), reinterpret_cast<const T *>(src.data()), src.size());
}

template <typename T> T *raw_pointer_cast(T *p) { return p; }

// element-wise equality across spaces (what the reference tests' ASSERT_EQUAL does for arrays)
template <typename A, typename B> bool equal(const A &a, const B &b)
{
    array1d<typename A::value_type, host_memory> ha(a);
    array1d<typename B::value_type, host_memory> hb(b);
    if (ha.size() != hb.size()) return false;
    for (size_t i = 0; i < ha.size(); i++)
        if (!(ha[i] == hb[i])) return false;
    return true;
}

} // namespace cusp
