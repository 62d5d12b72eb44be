// cusp/distributed/communicator.h -- one process per GPU: who am I, who else is there, and the collectives between us.
//
// MI355X-first design of SURVEY.md 8(e) (the reference is single-process, single-device: cusp/ktt/detail/ktt.inl:34-35): a job is
// started as N processes -- tools/bin/cmi_launch, or torchrun --no-python, both BEFORE anything touches a GPU -- which find each
// other through the environment torchrun also sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT; CMI_* variants override).
// A communicator holds
//   * the TCP star of transport.h: bootstrap and the collectives of host_memory vectors (tests, set-up records), and
//   * a cmi_comm (include/cusp_mi355x.h: RCCL over xGMI) made lazily at the first device collective: rank 0 draws the unique id,
//     the star hands it round, every rank calls cmi_comm_create after cmi_set_device(LOCAL_RANK).
// Device collectives are ENQUEUED on the stream (default: the null stream the rest of the header layer uses) and never block the
// host; host collectives block.
// REHEARSAL ($CMI_COMM_STAGED=1, tests only): several ranks share ONE GPU -- RCCL refuses that ("duplicate GPU") -- so the device
// collectives are staged through the host and the star (device -> host copy, TCP, host -> device; synchronous).  Everything above
// the transport (partitions, exchange plans, the one-sided IPC pull, CG's sequence of reductions) runs as on a node of GPUs.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../detail/config.h"
#include "../memory.h"
#include "transport.h"

namespace cusp {
namespace distributed {

class communicator {
public:
    // explicit: rank / world / where rank 0 listens.  device < 0: do not select a device (host-only jobs, or the caller has done it)
    communicator(int rank, int world, const std::string &master_addr = "127.0.0.1", int port = 29511, int device = -1)
        : rank_(rank), world_(world), star_(new detail::tcp_star(rank, world, master_addr, port)), comm_(nullptr), staged_(env_int("CMI_COMM_STAGED", 0) != 0)
    {
        if (rank < 0 || world < 1 || rank >= world) throw cusp::invalid_input_exception("communicator: bad rank / world size");
        if (device >= 0) cusp::detail::check(cmi_set_device(device));
    }
    ~communicator() { if (comm_) cmi_comm_destroy(comm_); }
    communicator(const communicator &) = delete;
    communicator &operator=(const communicator &) = delete;

    // from the launcher's environment; selects device LOCAL_RANK (mod the visible devices) when `use_device`
    static std::unique_ptr<communicator> from_environment(bool use_device = true)
    {
        const int rank = env_int("CMI_RANK", env_int("RANK", 0)), world = env_int("CMI_WORLD_SIZE", env_int("WORLD_SIZE", 1));
        const int local = env_int("CMI_LOCAL_RANK", env_int("LOCAL_RANK", rank));
        const char *addr = std::getenv("CMI_MASTER_ADDR");
        if (!addr) addr = std::getenv("MASTER_ADDR");
        // torchrun's MASTER_PORT belongs to its own store when Python ranks are around: the star listens one above it
        const int port = env_int("CMI_PORT", env_int("MASTER_PORT", 29510) + 1);
        int device = -1;
        if (use_device) {
            int count = 0;
            cusp::detail::check(cmi_device_count(&count));
            if (count < 1) throw cusp::runtime_exception("communicator: no GPU visible");
            device = local % count; // (a rehearsal with more ranks than GPUs lands several ranks on one device: CMI_COMM_STAGED=1)
        }
        return std::unique_ptr<communicator>(new communicator(rank, world, addr ? addr : "127.0.0.1", port, device));
    }

    int rank() const { return rank_; }
    int size() const { return world_; }
    bool staged() const { return staged_; }
    detail::tcp_star &host() { return *star_; }

    // the RCCL communicator behind the C-ABI (collective on first use: every rank must reach its first device collective)
    cmi_comm *device()
    {
        if (!comm_) {
            unsigned char id[CMI_COMM_ID_BYTES] = {0};
            if (rank_ == 0) cusp::detail::check(cmi_comm_unique_id(id));
            star_->broadcast(id, sizeof(id), 0);
            cusp::detail::check(cmi_comm_create(id, rank_, world_, &comm_));
        }
        return comm_;
    }

    // ---- collectives by memory space ------------------------------------------------------------------------------------------
    // recv[r * count, +count) <- rank r's send (in place when send == recv + rank * count)
    template <typename T> void allgather(const T *send, T *recv, size_t count, host_memory) { star_->allgather(send, recv, count * sizeof(T)); }
    void allgather(const double *send, double *recv, size_t count, device_memory, void *stream = nullptr)
    { if (staged_) staged_allgather(send, recv, count, stream); else cusp::detail::check(cmi_allgather_f64(device(), send, recv, (int64_t)count, stream)); }
    void allgather(const float *send, float *recv, size_t count, device_memory, void *stream = nullptr)
    { if (staged_) staged_allgather(send, recv, count, stream); else cusp::detail::check(cmi_allgather_f32(device(), send, recv, (int64_t)count, stream)); }
    // unequal pieces (element counts / displacements, `size()` entries each)
    template <typename T> void allgatherv(const T *send, T *recv, const int64_t *counts, const int64_t *displs, host_memory)
    {
        std::vector<size_t> c(world_), d(world_);
        for (int r = 0; r < world_; r++) { c[r] = (size_t)counts[r] * sizeof(T); d[r] = (size_t)displs[r] * sizeof(T); }
        star_->allgatherv(send, recv, c.data(), d.data());
    }
    void allgatherv(const double *send, double *recv, const int64_t *counts, const int64_t *displs, device_memory, int algorithm = 1, void *stream = nullptr)
    { if (staged_) staged_allgatherv(send, recv, counts, displs, stream); else cusp::detail::check(cmi_allgatherv_f64(device(), send, recv, counts, displs, algorithm, stream)); }
    void allgatherv(const float *send, float *recv, const int64_t *counts, const int64_t *displs, device_memory, int algorithm = 1, void *stream = nullptr)
    { if (staged_) staged_allgatherv(send, recv, counts, displs, stream); else cusp::detail::check(cmi_allgatherv_f32(device(), send, recv, counts, displs, algorithm, stream)); }
    // sum of `n` doubles over the ranks, in place
    void allreduce_sum(double *v, size_t n, host_memory) { star_->allreduce(v, n, 0); }
    void allreduce_sum(double *v, size_t n, device_memory, void *stream = nullptr)
    { if (staged_) staged_allreduce(v, n, 0, stream); else cusp::detail::check(cmi_allreduce_f64(device(), v, v, (int64_t)n, CMI_OP_SUM, stream)); }
    void allreduce_max(double *v, size_t n, host_memory) { star_->allreduce(v, n, 1); }
    void allreduce_max(double *v, size_t n, device_memory, void *stream = nullptr)
    { if (staged_) staged_allreduce(v, n, 1, stream); else cusp::detail::check(cmi_allreduce_f64(device(), v, v, (int64_t)n, CMI_OP_MAX, stream)); }
    void barrier(host_memory) { star_->barrier(); }
    // every rank has reached this point and everything queued on `stream` before it has completed everywhere
    void barrier(device_memory, void *stream = nullptr)
    {
        if (staged_) { cusp::detail::check(cmi_stream_synchronize(stream)); star_->barrier(); }
        else cusp::detail::check(cmi_comm_barrier(device(), stream));
    }
    // ranged exchange inside one full-length DEVICE buffer (the halo exchange), staged through the host: rehearsal transport only
    template <typename T> void staged_exchange(T *buf, size_t n, const size_t *send_lo, const size_t *send_n, const size_t *recv_lo, const size_t *recv_n, void *stream)
    {
        std::vector<T> h(n);
        cusp::detail::check(cmi_memcpy_d2h(h.data(), buf, n * sizeof(T), stream));
        star_->exchange(h.data(), send_lo, send_n, recv_lo, recv_n);
        for (int p = 0; p < world_; p++)
            if (p != rank_ && recv_n[p]) cusp::detail::check(cmi_memcpy_h2d(reinterpret_cast<char *>(buf) + recv_lo[p], reinterpret_cast<char *>(h.data()) + recv_lo[p], recv_n[p], stream));
    }

private:
    template <typename T> void staged_allgather(const T *send, T *recv, size_t count, void *stream)
    {
        std::vector<int64_t> c(world_, (int64_t)count), d(world_);
        for (int r = 0; r < world_; r++) d[r] = (int64_t)r * (int64_t)count;
        staged_allgatherv(send, recv, c.data(), d.data(), stream);
    }
    template <typename T> void staged_allgatherv(const T *send, T *recv, const int64_t *counts, const int64_t *displs, void *stream)
    {
        size_t total = 0;
        for (int r = 0; r < world_; r++) total = std::max<size_t>(total, (size_t)(displs[r] + counts[r]));
        std::vector<T> h(total);
        if (counts[rank_]) cusp::detail::check(cmi_memcpy_d2h(h.data() + displs[rank_], send, (size_t)counts[rank_] * sizeof(T), stream));
        allgatherv(h.data() + displs[rank_], h.data(), counts, displs, host_memory());
        for (int r = 0; r < world_; r++) // (the rank's own piece too: `send` need not be in place)
            if (counts[r]) cusp::detail::check(cmi_memcpy_h2d(recv + displs[r], h.data() + displs[r], (size_t)counts[r] * sizeof(T), stream));
    }
    void staged_allreduce(double *v, size_t n, int op, void *stream)
    {
        std::vector<double> h(n);
        cusp::detail::check(cmi_memcpy_d2h(h.data(), v, n * sizeof(double), stream));
        star_->allreduce(h.data(), n, op);
        cusp::detail::check(cmi_memcpy_h2d(v, h.data(), n * sizeof(double), stream));
    }
    static int env_int(const char *name, int fallback)
    {
        const char *e = std::getenv(name);
        return (e && e[0]) ? std::atoi(e) : fallback;
    }
    int rank_, world_;
    std::unique_ptr<detail::tcp_star> star_;
    cmi_comm *comm_;
    bool staged_;
};

// Row partitions.  Both return world + 1 non-decreasing cuts from 0 to num_rows.
//   partition_rows        equal counts, ceil(num_rows / world) each (what the in-place ncclAllGather needs)
//   partition_by_entries  block r ends at the first row whose cumulative entry count reaches (r + 1) / world of the total
//                         ("balanced by nnz"); slices differ in length -> all-gather of unequal pieces
inline std::vector<int64_t> partition_rows(int64_t num_rows, int world)
{
    const int64_t count = world > 0 ? (num_rows + world - 1) / world : 0;
    std::vector<int64_t> cuts(world + 1);
    for (int r = 0; r <= world; r++) cuts[r] = std::min<int64_t>((int64_t)r * count, num_rows);
    return cuts;
}
template <typename Offsets> std::vector<int64_t> partition_by_entries(const Offsets &row_offsets, int world)
{
    const int64_t num_rows = (int64_t)row_offsets.size() - 1, nnz = num_rows >= 0 ? (int64_t)row_offsets[num_rows] : 0;
    std::vector<int64_t> cuts(1, 0);
    for (int r = 1; r < world; r++) {
        const int64_t target = (nnz * r + world - 1) / world;
        int64_t lo = cuts.back(), hi = num_rows; // first row offset >= target
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if ((int64_t)row_offsets[mid] < target) lo = mid + 1; else hi = mid; }
        cuts.push_back(std::min(std::max(lo, cuts.back()), num_rows));
    }
    cuts.push_back(num_rows);
    return cuts;
}

} // namespace distributed
} // namespace cusp
