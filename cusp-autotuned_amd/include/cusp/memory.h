// cusp/memory.h -- memory-space tags (reference cusp/memory.h; there they are Thrust execution
// policies, cusp/iterator/detail/{host,device}_system_tag.h:29-30).  device_memory = HBM of the
// current HIP device, owned through cmi_malloc / cmi_free.
#pragma once
#include "detail/config.h"

namespace cusp {

struct host_memory {};
struct device_memory {};
typedef device_memory any_device; // convenience

// A vector or operator SHARDED over the ranks of a cusp::distributed::communicator (one process per GPU), each rank holding its
// piece in `Local` memory (cusp/distributed/*.h).  cusp::blas and cusp::multiply dispatch on it like on the two spaces above:
// element-wise work runs on the local piece, reductions are all-reduced.  No reference equivalent (single device).
template <typename Local> struct distributed_memory { typedef Local local_space; };

namespace detail {
template <typename A, typename B> struct is_same_space { static const bool value = false; };
template <typename A> struct is_same_space<A, A> { static const bool value = true; };
} // namespace detail

} // namespace cusp
