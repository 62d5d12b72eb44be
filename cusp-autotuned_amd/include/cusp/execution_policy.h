// cusp/execution_policy.h -- execution policies (reference cusp/execution_policy.h; there they are
// Thrust policies and double as memory spaces).
//
//   cusp::hip::par            default policy: the HIP default stream
//   cusp::hip::par.on(s)      run the multiply on hipStream_t s (reference: cusp::cuda::par.on(stream),
//                             stream(derived_cast(exec)) at cuda/detail/multiply/csr_vector_spmv.h:198)
//   cusp::omp::par            host_memory, OpenMP: the CSR multiply runs the reference's OpenMP backend loop
//                             (cusp/system/omp/detail/multiply/csr_spmv.h:51-86: `#pragma omp parallel for` over the
//                             rows, the sequential kernel's body); every other format inherits the sequential loops,
//                             as the reference's omp system does (cusp/system/omp/detail/multiply.h:26-40).  Built
//                             without -fopenmp the pragma is inert and the loop is the sequential one.
//   cusp::execution_policy<Derived>
//                             CRTP base for USER policies.  cusp::multiply(policy, A, x, y) makes an
//                             unqualified call with the derived policy (never copied), so a user
//                             `multiply(my_policy&, ...)` overload in the policy's namespace is found by
//                             ADL and wins -- the extension point of testing/multiply.cu:792-858
//                             (testing/unittest/special_types.h:108-141).
#pragma once
#include "detail/config.h"

namespace cusp {

template <typename Derived> struct execution_policy {
    Derived &derived() { return static_cast<Derived &>(*this); }
    const Derived &derived() const { return static_cast<const Derived &>(*this); }
};

namespace hip {

class execution_policy : public cusp::execution_policy<execution_policy> {
public:
    execution_policy() : stream_(nullptr) {}
    explicit execution_policy(void *stream) : stream_(stream) {}
    // hipStream_t is a pointer type; taken as void* so this header needs no HIP headers
    execution_policy on(void *stream) const { return execution_policy(stream); }
    void *stream() const { return stream_; }
private:
    void *stream_;
};

static const execution_policy par;

} // namespace hip

namespace omp {
class execution_policy : public cusp::execution_policy<execution_policy> {};
static const execution_policy par;
} // namespace omp

} // namespace cusp
