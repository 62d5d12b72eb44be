// cusp/distributed/vector.h -- a vector sharded over the ranks: this rank's slice [offset, offset + size()) of `global_size`
// elements, in `Local` memory.  memory_space = cusp::distributed_memory<Local>, so cusp::blas::{axpy, axpby, copy, fill} run on the
// slice and cusp::blas::{dot, dotc, nrm2} all-reduce (cusp/blas/blas.h) -- cusp::monitor<T>(b) therefore sees ||b|| of the WHOLE
// vector, and a Krylov solver written against cusp::blas (reference cusp/krylov/detail/cg.inl:41-107) runs unchanged.
// The slice is either owned or a VIEW of memory owned elsewhere: a sharded operator keeps its x slice inside the full-length
// exchange buffer (cusp/distributed/csr_matrix.h), and CG's direction vector p is such a view -- no copy before the all-gather.
#pragma once
#include "../array1d.h"
#include "communicator.h"

namespace cusp {
namespace distributed {

template <typename T, typename Local> class vector {
public:
    typedef T value_type;
    typedef cusp::distributed_memory<Local> memory_space;
    typedef Local local_space;
    typedef cusp::array1d_format format;
    typedef cusp::array1d_view<T, Local> local_view;
    typedef cusp::array1d_view<const T, Local> const_local_view;

    vector() : comm_(nullptr), ptr_(nullptr), size_(0), global_(0), offset_(0) {}
    // owning: `local_size` elements of a vector of `global_size`, starting at global index `offset`
    vector(communicator &c, size_t local_size, size_t global_size, size_t offset)
        : comm_(&c), storage_(local_size), ptr_(storage_.data()), size_(local_size), global_(global_size), offset_(offset) {}
    vector(communicator &c, size_t local_size, size_t global_size, size_t offset, const T &fill_value)
        : comm_(&c), storage_(local_size, fill_value), ptr_(storage_.data()), size_(local_size), global_(global_size), offset_(offset) {}
    // view of memory owned elsewhere
    vector(communicator &c, T *data, size_t local_size, size_t global_size, size_t offset)
        : comm_(&c), ptr_(data), size_(local_size), global_(global_size), offset_(offset) {}
    vector(const vector &o) : comm_(o.comm_), storage_(o.local()), ptr_(storage_.data()), size_(o.size_), global_(o.global_), offset_(o.offset_) {} // (deep copy, owning)
    vector(vector &&o) noexcept : comm_(o.comm_), storage_(std::move(o.storage_)), ptr_(o.ptr_), size_(o.size_), global_(o.global_), offset_(o.offset_) { o.ptr_ = nullptr; o.size_ = 0; }
    vector &operator=(const vector &o) // element-wise into the existing slice when the shapes agree (a view stays a view)
    {
        if (this == &o) return *this;
        if (size_ == o.size_ && ptr_) { local_view dst = local(); cusp::copy_array(o.local(), dst); comm_ = o.comm_; global_ = o.global_; offset_ = o.offset_; return *this; }
        storage_ = o.local();
        comm_ = o.comm_; ptr_ = storage_.data(); size_ = o.size_; global_ = o.global_; offset_ = o.offset_;
        return *this;
    }

    // another owning vector with this one's shape (what a solver's work vectors are)
    vector like() const { return vector(*comm_, size_, global_, offset_); }
    vector like(const T &fill_value) const { return vector(*comm_, size_, global_, offset_, fill_value); }

    size_t size() const { return size_; }          // LOCAL length: what cusp::blas checks and loops over
    size_t global_size() const { return global_; }
    size_t offset() const { return offset_; }
    T *data() { return ptr_; }
    const T *data() const { return ptr_; }
    local_view local() { return local_view(ptr_, size_); }
    const_local_view local() const { return const_local_view(ptr_, size_); }
    communicator &comm() const { return *comm_; }
    bool is_view() const { return ptr_ != storage_.data() || storage_.size() != size_; }

    // set-up / test convenience: the whole vector on the host of every rank (an all-gather of unequal pieces over the star)
    cusp::array1d<T, cusp::host_memory> gather() const
    {
        cusp::array1d<T, cusp::host_memory> mine(local()), all(global_);
        std::vector<int64_t> counts(comm_->size()), displs(comm_->size());
        int64_t rec[2] = {(int64_t)size_, (int64_t)offset_};
        std::vector<int64_t> recs(2 * comm_->size());
        comm_->host().allgather(rec, recs.data(), sizeof(rec));
        for (int r = 0; r < comm_->size(); r++) { counts[r] = recs[2 * r]; displs[r] = recs[2 * r + 1]; }
        comm_->allgatherv(mine.data(), all.data(), counts.data(), displs.data(), cusp::host_memory());
        return all;
    }

private:
    communicator *comm_;
    cusp::array1d<T, Local> storage_;
    T *ptr_;
    size_t size_, global_, offset_;
};

} // namespace distributed
} // namespace cusp
