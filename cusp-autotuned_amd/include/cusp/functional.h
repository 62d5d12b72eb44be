// cusp/functional.h -- the functors the 7-argument cusp::multiply is called with
// (reference cusp/functional.h; generic/multiply.inl:104-110 uses constant_functor(0), multiplies, plus).
#pragma once
namespace cusp {
template <typename T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };
template <typename T> struct multiplies { T operator()(const T &a, const T &b) const { return a * b; } };
template <typename T> struct identity_function { T operator()(const T &a) const { return a; } };
template <typename T> struct constant_functor {
    T value;
    explicit constant_functor(T v = T(0)) : value(v) {}
    T operator()(const T &) const { return value; }
};
} // namespace cusp
