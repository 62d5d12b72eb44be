// cusp/distributed/csr_matrix.h -- a CSR matrix sharded by ROW BLOCKS over the ranks of a communicator (one process per GPU).
//
// SURVEY.md 8(e) / BASELINE.json configs[4]: rank g owns the contiguous rows [cuts[g], cuts[g + 1]) of A as an ordinary
// cusp::csr_matrix with GLOBAL column indices, the matching slices of every vector, and a full-length x buffer; before each
// multiply the x entries its rows reference are brought into that buffer by ONE exchange step, then the single-GPU hot path runs
// (cusp::multiply on the local block -> cmi_spmv_csr_plan_*).  The reference has no distributed code (its caller is
// cusp/krylov/detail/cg.inl:80, `y <- A p`, on one device); cusp::multiply and cusp::krylov::cg accept this operator
// (cusp/distributed/multiply.h, cusp/distributed/cg.h).
//
// Exchange modes (the same volumes and decisions as the round-1/2 Python rehearsal, now through the C-ABI):
//   allgather  every rank contributes its slice: cmi_allgather_* IN PLACE (the slice already sits at its place in the buffer) for
//              equal row counts, cmi_allgatherv_* for blocks balanced by entries.  The north-star exchange; per rank
//              (world - 1) / world * N values, bound by the xGMI links.
//   halo       only the parts of other ranks' slices inside this rank's column window [col_min, col_max] move: one grouped
//              ncclSend / ncclRecv launch (cmi_halo_exchange_*).  5-point Poisson: 2 m values per rank instead of N.
//   peer       the halo plan, ONE-SIDED: xGMI is a load / store fabric, so every rank maps its neighbours' exchange buffers once
//              (cmi_ipc_get_handle / cmi_ipc_open_handle; handles travel over the star) and before a multiply PULLS the ranges it needs
//              with one small copy kernel on its own stream (cmi_copy_ranges) -- no collective launch at all.  Visibility is at kernel
//              boundaries: the CALLER orders the peers' producing kernels before the pull.  cusp::multiply does it with two fences per
//              call (correct, not fast); cusp::krylov::cg needs none inside its loop -- its own all-reduces are the ordering
//              (cusp/distributed/cg.h).  Verified end to end at set-up (a pull of per-rank signatures); any failure on any rank leaves
//              the two-sided halo exchange in force.
//   automatic  halo when the WORST rank's halo volume is less than half of the all-gather's, else allgather (decided alike
//              on every rank); the halo one-sided when $CMI_EXCHANGE_PEER is not 0 and every rank could map and verify its peers.
// host_memory operators run the same plans over the TCP star (tests, set-up; no one-sided mode there).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "../blas/blas.h"
#include "../csr_matrix.h"
#include "../multiply.h"
#include "communicator.h"
#include "vector.h"

namespace cusp {
namespace distributed {

enum class exchange_mode { automatic, allgather, halo, peer };

namespace detail {
inline void column_span(const cusp::csr_matrix<int, double, cusp::device_memory> &a, int &lo, int &hi) { cusp::detail::check(cmi_csr_column_span((int64_t)a.num_entries, a.column_indices.data(), &lo, &hi, nullptr)); }
inline void column_span(const cusp::csr_matrix<int, float, cusp::device_memory> &a, int &lo, int &hi) { cusp::detail::check(cmi_csr_column_span((int64_t)a.num_entries, a.column_indices.data(), &lo, &hi, nullptr)); }
template <typename I, typename V> void column_span(const cusp::csr_matrix<I, V, cusp::host_memory> &a, int &lo, int &hi)
{
    lo = 0; hi = -1;
    if (a.num_entries == 0) return;
    lo = hi = (int)a.column_indices[0];
    for (size_t k = 1; k < a.num_entries; k++) { const int c = (int)a.column_indices[k]; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
}
inline int halo_call(cmi_comm *c, double *x, int n, const int *peers, const int64_t *sl, const int64_t *sc, const int64_t *rl, const int64_t *rc, void *s)
{ return cmi_halo_exchange_f64(c, x, n, peers, sl, sc, rl, rc, s); }
inline int halo_call(cmi_comm *c, float *x, int n, const int *peers, const int64_t *sl, const int64_t *sc, const int64_t *rl, const int64_t *rc, void *s)
{ return cmi_halo_exchange_f32(c, x, n, peers, sl, sc, rl, rc, s); }
} // namespace detail

// A full-length vector buffer (indexed by global column) that peers can PULL from: the allocation (its own cmi_malloc, so that its
// IPC handle maps exactly it), the mapped peers' buffers and the copy ranges of this rank's halo.  host_memory: the plain array.
template <typename T, typename Local> class exchange_buffer {
public:
    cusp::array1d<T, Local> data;
    bool map_peers(communicator &, const std::vector<int> &, const std::vector<int64_t> &, const std::vector<int64_t> &) { return false; }
    void pull(void * = nullptr) const {}
    void close() {}
};
template <typename T> class exchange_buffer<T, cusp::device_memory> {
public:
    cusp::array1d<T, cusp::device_memory> data;
    exchange_buffer() {}
    ~exchange_buffer() { close(); }
    exchange_buffer(const exchange_buffer &) = delete;
    exchange_buffer &operator=(const exchange_buffer &) = delete;

    // COLLECTIVE.  Maps every peer this rank receives from and builds the pull; true only if EVERY rank succeeded and a test pull of
    // per-rank signatures returned the right data (then the buffer is left zeroed).  On false nothing stays mapped.
    bool map_peers(communicator &comm, const std::vector<int> &peers, const std::vector<int64_t> &recv_lo, const std::vector<int64_t> &recv_n)
    {
        const int world = comm.size(), rank = comm.rank();
        struct record { unsigned char handle[CMI_IPC_HANDLE_BYTES]; int32_t device, ok; } mine;
        std::memset(&mine, 0, sizeof(mine));
        mine.ok = cmi_ipc_get_handle(data.data(), mine.handle) == CMI_SUCCESS && cmi_get_device(&mine.device) == CMI_SUCCESS;
        std::vector<record> all(world);
        comm.host().allgather(&mine, all.data(), sizeof(record));
        double good = 1.0;
        for (int r = 0; r < world; r++) if (!all[r].ok) good = 0.0;
        src_.clear(); dst_.clear(); bytes_.clear();
        if (good > 0) {
            for (size_t i = 0; i < peers.size() && good > 0; i++) {
                if (recv_n[i] <= 0) continue;
                const int p = peers[i];
                int can = 0;
                if (cmi_device_can_access_peer(mine.device, all[p].device, &can) != CMI_SUCCESS || !can) { good = 0.0; break; }
                void *ptr = nullptr;
                if (cmi_ipc_open_handle(all[p].handle, &ptr) != CMI_SUCCESS) { good = 0.0; break; }
                mapped_.push_back(ptr);
                // every rank's buffer is indexed by GLOBAL column: the same offset on both sides
                src_.push_back(static_cast<const char *>(ptr) + (size_t)recv_lo[i] * sizeof(T));
                dst_.push_back(reinterpret_cast<char *>(data.data()) + (size_t)recv_lo[i] * sizeof(T));
                bytes_.push_back((int64_t)recv_n[i] * (int64_t)sizeof(T));
            }
        }
        // end-to-end check with a per-rank signature (every collective step runs on every rank, mapped or not)
        cusp::array1d_view<T, cusp::device_memory> whole(data.data(), data.size());
        cusp::blas::fill(whole, T(rank + 1));
        comm.barrier(cusp::device_memory());
        if (good > 0) {
            pull();
            if (cmi_stream_synchronize(nullptr) != CMI_SUCCESS) good = 0.0;
            for (size_t i = 0; i < peers.size() && good > 0; i++) {
                if (recv_n[i] <= 0) continue;
                T first, last;
                cusp::detail::check(cmi_memcpy_d2h(&first, data.data() + recv_lo[i], sizeof(T), nullptr));
                cusp::detail::check(cmi_memcpy_d2h(&last, data.data() + recv_lo[i] + recv_n[i] - 1, sizeof(T), nullptr));
                if (first != T(peers[i] + 1) || last != T(peers[i] + 1)) good = 0.0;
            }
        }
        double bad = 1.0 - good;
        comm.allreduce_max(&bad, 1, cusp::host_memory());
        comm.barrier(cusp::device_memory());
        cusp::blas::fill(whole, T(0));
        comm.barrier(cusp::device_memory());
        if (bad > 0) { close(); return false; }
        return true;
    }
    // the pull: one launch per 16 ranges on `stream`; nothing to wait for on the host
    void pull(void *stream = nullptr) const
    {
        for (size_t i = 0; i < src_.size(); i += CMI_MAX_COPY_RANGES) {
            const int n = (int)std::min<size_t>(CMI_MAX_COPY_RANGES, src_.size() - i);
            cusp::detail::check(cmi_copy_ranges(n, reinterpret_cast<const void *const *>(src_.data() + i), reinterpret_cast<void *const *>(dst_.data() + i), bytes_.data() + i, stream));
        }
    }
    void close()
    {
        for (void *p : mapped_) cmi_ipc_close_handle(p);
        mapped_.clear(); src_.clear(); dst_.clear(); bytes_.clear();
    }
    bool mapped() const { return !src_.empty() || !mapped_.empty(); }

private:
    std::vector<void *> mapped_;
    std::vector<const char *> src_;
    std::vector<char *> dst_;
    std::vector<int64_t> bytes_;
};

template <typename IndexType, typename ValueType, typename Local> class csr_matrix {
public:
    typedef IndexType index_type;
    typedef ValueType value_type;
    typedef cusp::distributed_memory<Local> memory_space;
    typedef Local local_space;
    typedef cusp::csr_matrix<IndexType, ValueType, Local> local_matrix_type;
    typedef vector<ValueType, Local> vector_type;

    size_t num_rows, num_cols, num_entries; // of the WHOLE matrix
    local_matrix_type local;                // this rank's rows x num_cols, global column indices
    std::vector<int64_t> cuts;              // world + 1 row cuts; x and y are sharded by the same cuts

    explicit csr_matrix(communicator &c) : num_rows(0), num_cols(0), num_entries(0), comm_(&c), mode_(exchange_mode::allgather), count_(0), uniform_(true) {}

    communicator &comm() const { return *comm_; }
    size_t row_begin() const { return (size_t)cuts[comm_->rank()]; }
    size_t row_end() const { return (size_t)cuts[comm_->rank() + 1]; }
    size_t local_rows() const { return row_end() - row_begin(); }
    exchange_mode mode() const { return mode_; }
    // values this rank receives per exchange, and what the all-gather would receive
    int64_t exchange_values() const { return mode_ == exchange_mode::allgather ? allgather_values() : halo_recv_values_; }
    const char *mode_name() const { return mode_ == exchange_mode::peer ? "peer (one-sided pull)" : mode_ == exchange_mode::halo ? "halo (grouped send/recv)" : "allgather"; }
    // the halo this rank keeps of the other ranks' slices: (first global index, length) per peer it receives from
    std::vector<std::pair<int64_t, int64_t>> halo_ranges() const
    {
        std::vector<std::pair<int64_t, int64_t>> out;
        for (size_t i = 0; i < peers_.size(); i++) if (recv_n_[i] > 0) out.push_back({recv_lo_[i], recv_n_[i]});
        return out;
    }
    // Another full-length buffer with this operator's halo plan and -- in peer mode -- its own mappings of the neighbours' copies
    // (COLLECTIVE).  cusp::krylov::cg keeps its residual in one so that peers can pull its boundary values.
    void make_exchange_buffer(exchange_buffer<ValueType, Local> &out) const
    {
        out.data.resize(xbuf_.data.size());
        cusp::array1d_view<ValueType, Local> whole(out.data.data(), out.data.size());
        cusp::blas::fill(whole, ValueType(0));
        if (mode_ == exchange_mode::peer && !out.map_peers(*comm_, peers_, recv_lo_, recv_n_))
            throw cusp::runtime_exception("distributed::csr_matrix: a second exchange buffer could not be mapped by every rank");
    }
    // everything queued on this rank's stream has completed AND every rank has reached this point: the epoch boundary the one-sided
    // pull needs between the peers' producers and itself when the caller's algorithm does not already provide it
    void fence() const { comm_->barrier(Local()); }
    // the pull alone (peer mode), no ordering: for callers whose algorithm orders it (cg)
    void pull(void *stream = nullptr) const { xbuf_.pull(stream); }
    int64_t allgather_values() const { return (int64_t)(comm_->size() - 1) * count_; }
    int64_t halo_values() const { return halo_recv_values_; }

    // Adopt a row block (COLLECTIVE: every rank calls it with its own block).  `block`: rows [row_cuts[rank], row_cuts[rank + 1]) of a
    // square global matrix of `global_rows` rows, global column indices; moved in.
    void assemble(local_matrix_type &&block, size_t global_rows, const std::vector<int64_t> &row_cuts, exchange_mode want = exchange_mode::automatic)
    {
        const int world = comm_->size(), rank = comm_->rank();
        if ((int)row_cuts.size() != world + 1 || row_cuts.front() != 0 || row_cuts.back() != (int64_t)global_rows)
            throw cusp::invalid_input_exception("distributed::csr_matrix: row cuts must be world + 1 values from 0 to the global row count");
        for (int r = 0; r < world; r++) if (row_cuts[r + 1] < row_cuts[r]) throw cusp::invalid_input_exception("distributed::csr_matrix: row cuts must not decrease");
        if (block.num_rows != (size_t)(row_cuts[rank + 1] - row_cuts[rank]) || block.num_cols != global_rows)
            throw cusp::invalid_input_exception("distributed::csr_matrix: the local block must hold this rank's rows and all (global) columns of a square matrix");
        local = std::move(block);
        cuts = row_cuts;
        num_rows = num_cols = global_rows;
        count_ = 0;
        for (int r = 0; r < world; r++) count_ = std::max<int64_t>(count_, cuts[r + 1] - cuts[r]);
        uniform_ = true;
        for (int r = 0; r <= world; r++) uniform_ = uniform_ && cuts[r] == std::min<int64_t>((int64_t)r * count_, (int64_t)global_rows);
        // the buffer is padded to world * count so that the in-place all-gather of a uniform partition can write straight into it
        xbuf_.data.resize(std::max<size_t>((size_t)world * (size_t)count_, std::max<size_t>(global_rows, 1)));
        cusp::array1d_view<ValueType, Local> whole(xbuf_.data.data(), xbuf_.data.size());
        cusp::blas::fill(whole, ValueType(0));

        // every rank learns every rank's column window and entry count (set-up, over the star)
        int lo = 0, hi = -1;
        detail::column_span(local, lo, hi);
        int64_t rec[3] = {lo, hi, (int64_t)local.num_entries};
        std::vector<int64_t> recs(3 * world);
        comm_->host().allgather(rec, recs.data(), sizeof(rec));
        num_entries = 0;
        for (int r = 0; r < world; r++) num_entries += (size_t)recs[3 * r + 2];
        auto overlap = [&](int64_t slo, int64_t shi, int owner, int64_t &l, int64_t &n) { // [slo, shi] with owner's slice
            const int64_t a = std::max(slo, cuts[owner]), b = std::min(shi + 1, cuts[owner + 1]);
            l = b > a ? a : 0;
            n = b > a ? b - a : 0;
        };
        peers_.clear(); send_lo_.clear(); send_n_.clear(); recv_lo_.clear(); recv_n_.clear();
        halo_recv_values_ = 0;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            int64_t rl, rn, sl, sn;
            overlap(recs[3 * rank], recs[3 * rank + 1], p, rl, rn);  // what I need of p's slice
            overlap(recs[3 * p], recs[3 * p + 1], rank, sl, sn);     // what p needs of mine
            if (rn == 0 && sn == 0) continue;
            peers_.push_back(p); recv_lo_.push_back(rl); recv_n_.push_back(rn); send_lo_.push_back(sl); send_n_.push_back(sn);
            halo_recv_values_ += rn;
        }
        double worst = (double)halo_recv_values_;
        comm_->allreduce_max(&worst, 1, cusp::host_memory());
        worst_halo_values_ = (int64_t)worst;
        mode_ = want;
        if (want == exchange_mode::automatic) mode_ = 2 * worst_halo_values_ < allgather_values() ? exchange_mode::halo : exchange_mode::allgather;
        counts_.resize(world); displs_.resize(world);
        for (int r = 0; r < world; r++) { counts_[r] = cuts[r + 1] - cuts[r]; displs_[r] = cuts[r]; }
        // the halo one-sided where the buffers are in HBM and every rank can map and verify its neighbours' (collective; any failure
        // anywhere leaves the two-sided exchange in force -- asked for explicitly, it is an error instead)
        const char *pe = std::getenv("CMI_EXCHANGE_PEER");
        const bool try_peer = world > 1 && std::is_same<Local, cusp::device_memory>::value &&
                              (want == exchange_mode::peer || (want == exchange_mode::automatic && mode_ == exchange_mode::halo && !(pe && pe[0] == '0')));
        if (mode_ == exchange_mode::peer) mode_ = exchange_mode::halo;
        if (try_peer) {
            if (xbuf_.map_peers(*comm_, peers_, recv_lo_, recv_n_)) mode_ = exchange_mode::peer;
            else if (want == exchange_mode::peer) throw cusp::runtime_exception("distributed::csr_matrix: exchange_mode::peer asked for, but a rank could not map or verify its neighbours' buffers");
        }
        setup_overlap(Local());
    }

    // in the two-sided halo mode the rows [interior_first, interior_last) of the block -- those whose columns all lie in this
    // rank's own slice -- are multiplied on a side stream WHILE the halo is in flight; the boundary rows at the block's two ends follow
    // on the caller's stream once it has landed.  A row range of a CSR block is itself a CSR matrix (the offsets are absolute positions
    // in the shared column / value arrays): no copy, same kernels, same bits.  Off: $CMI_EXCHANGE_OVERLAP=0, host_memory, one rank,
    // the all-gather and the one-sided modes (the pull is one small copy kernel: nothing to hide).
    bool overlapped() const { return overlap_; }
    size_t interior_first() const { return (size_t)interior_a_; }
    size_t interior_last() const { return (size_t)interior_b_; }

    // Every rank holds the WHOLE matrix on the host (tests, MatrixMarket input): take this rank's rows (collective).
    template <typename HostCsr> void scatter(const HostCsr &A, const std::vector<int64_t> &row_cuts, exchange_mode want = exchange_mode::automatic)
    {
        const int rank = comm_->rank();
        if (A.num_rows != A.num_cols) throw cusp::invalid_input_exception("distributed::csr_matrix: square matrices only (x is sharded like the rows)");
        const size_t r0 = (size_t)row_cuts[rank], r1 = (size_t)row_cuts[rank + 1];
        const size_t e0 = (size_t)A.row_offsets[r0], e1 = (size_t)A.row_offsets[r1];
        cusp::csr_matrix<IndexType, ValueType, cusp::host_memory> h(r1 - r0, A.num_cols, e1 - e0);
        for (size_t i = r0; i <= r1; i++) h.row_offsets[i - r0] = (IndexType)((size_t)A.row_offsets[i] - e0);
        for (size_t k = e0; k < e1; k++) { h.column_indices[k - e0] = (IndexType)A.column_indices[k]; h.values[k - e0] = (ValueType)A.values[k]; }
        local_matrix_type block(h);
        assemble(std::move(block), A.num_rows, row_cuts, want);
    }

    // a work vector sharded like the rows (owning), and THE x slice: a view of this rank's place in the exchange buffer -- fill it,
    // call exchange(), and local rows can be multiplied; CG keeps its direction vector p there
    vector_type make_vector() const { return vector_type(*comm_, local_rows(), num_rows, row_begin()); }
    vector_type make_vector(const ValueType &v) const { return vector_type(*comm_, local_rows(), num_rows, row_begin(), v); }
    vector_type exchange_slice() const { return vector_type(*comm_, const_cast<ValueType *>(xbuf_.data.data()) + row_begin(), local_rows(), num_rows, row_begin()); }
    const ValueType *x_full() const { return xbuf_.data.data(); }
    ValueType *x_full() { return xbuf_.data.data(); }
    cusp::array1d_view<const ValueType, Local> x_view() const { return cusp::array1d_view<const ValueType, Local>(xbuf_.data.data(), num_cols); }

    // fill the exchange buffer from every rank's slice (COLLECTIVE; device: enqueued on `stream`, returns at once)
    void exchange(void *stream = nullptr) const { exchange_impl(stream, Local()); }

    // y_local <- A[rows of this rank, :] * (buffer)   -- the buffer must hold what exchange() brings
    template <typename Y> void multiply_local(Y &y_local) const
    {
        auto xv = x_view();
        cusp::multiply(local, xv, y_local);
    }

    // exchange() + multiply_local() -- with the interior rows overlapped with the exchange where that is set up (COLLECTIVE)
    template <typename Y> void exchange_and_multiply(Y &y_local, void *stream = nullptr) const
    {
        if (!overlap_) {
            exchange(stream);
            multiply_local(y_local);
            return;
        }
        multiply_overlapped(y_local.data(), stream, Local());
    }

private:
    communicator *comm_;
    exchange_mode mode_;
    int64_t count_;      // longest slice = the all-gather's padded piece
    bool uniform_;       // cuts[r] == r * count_: the in-place ncclAllGather applies
    mutable exchange_buffer<ValueType, Local> xbuf_;
    std::vector<int> peers_;
    std::vector<int64_t> send_lo_, send_n_, recv_lo_, recv_n_, counts_, displs_;
    int64_t halo_recv_values_ = 0, worst_halo_values_ = 0;
    bool overlap_ = false;
    int64_t interior_a_ = 0, interior_b_ = 0;
    cmi_config row_cfg_ = cmi_config();
    bool have_row_cfg_ = false;
    struct side_resources { // a stream and two events, released with the operator
        void *stream = nullptr, *ready = nullptr, *done = nullptr;
        ~side_resources()
        {
            if (ready) (void)cmi_event_destroy(ready);
            if (done) (void)cmi_event_destroy(done);
            if (stream) (void)cmi_stream_destroy(stream);
        }
    };
    std::shared_ptr<side_resources> side_;

    void setup_overlap(cusp::host_memory) { overlap_ = false; }
    void setup_overlap(cusp::device_memory)
    {
        overlap_ = false;
        const char *e = std::getenv("CMI_EXCHANGE_OVERLAP");
        if (comm_->size() == 1 || mode_ != exchange_mode::halo || (e && e[0] == '0') || local.num_rows == 0 || local.num_entries == 0) return;
        if (!std::is_same<IndexType, int>::value) return;
        int64_t a = 0, b = 0;
        cusp::detail::check(cmi_csr_interior_rows((int64_t)local.num_rows, reinterpret_cast<const int *>(local.row_offsets.data()),
                                                  reinterpret_cast<const int *>(local.column_indices.data()), cuts[comm_->rank()], cuts[comm_->rank() + 1], &a, &b, nullptr));
        // every rank must take the same path through the collective below (it does not: the schedule is local -- the exchange is the
        // same call either way), so the decision is this rank's alone
        if (b - a <= (int64_t)local.num_rows / 2) return;
        // the launch shape of the WHOLE block (a row range passes the full column / value arrays: its own entry count is not what
        // the table should see), or what the block's plan made of it when that is a row-tile kernel that runs without a plan
        const cmi_plan *plan = local.plan();
        have_row_cfg_ = false;
        if (plan && cmi_plan_config(plan, &row_cfg_) == CMI_SUCCESS && (row_cfg_.kernel == CMI_CSR_STREAM || (row_cfg_.kernel == CMI_CSR_STREAM_WAVE && row_cfg_.rows_per_block > 0))) have_row_cfg_ = true; // (a WAVE plan on a row partition reads rows_per_block 0: its shape is the plan's, not a plan-less call's)
        if (!have_row_cfg_) {
            cusp::detail::check(cmi_tuning_select(CMI_FORMAT_CSR, cusp::detail::dtype_code<ValueType>::value, (int64_t)local.num_rows, (int64_t)local.num_cols,
                                                  (int64_t)local.num_entries, &row_cfg_));
            have_row_cfg_ = true;
        }
        side_ = std::make_shared<side_resources>();
        cusp::detail::check(cmi_stream_create(&side_->stream));
        cusp::detail::check(cmi_event_create(&side_->ready));
        cusp::detail::check(cmi_event_create(&side_->done));
        interior_a_ = a;
        interior_b_ = b;
        overlap_ = true;
    }
    static int rows_call(int64_t rows, int64_t cols, int64_t nnz, const int *Ap, const int *Aj, const double *Ax, const double *x, double *y, const cmi_config *c, void *s)
    { return cmi_spmv_csr_f64(rows, cols, nnz, Ap, Aj, Ax, x, y, 0, c, s); }
    static int rows_call(int64_t rows, int64_t cols, int64_t nnz, const int *Ap, const int *Aj, const float *Ax, const float *x, float *y, const cmi_config *c, void *s)
    { return cmi_spmv_csr_f32(rows, cols, nnz, Ap, Aj, Ax, x, y, 0, c, s); }
    void rows(int64_t a, int64_t b, ValueType *y, void *stream) const // y[a:b] = A[a:b, :] * buffer
    {
        if (b <= a) return;
        cusp::detail::check(rows_call(b - a, (int64_t)local.num_cols, (int64_t)local.num_entries, reinterpret_cast<const int *>(local.row_offsets.data()) + a,
                                      reinterpret_cast<const int *>(local.column_indices.data()), local.values.data(), xbuf_.data.data(), y + a, &row_cfg_, stream));
    }
    void multiply_overlapped(ValueType *, void *, cusp::host_memory) const {}
    void multiply_overlapped(ValueType *y, void *stream, cusp::device_memory) const
    {
        cusp::detail::check(cmi_event_record(side_->ready, stream));           // this rank's slice of x is in the buffer (stream order)
        cusp::detail::check(cmi_stream_wait_event(side_->stream, side_->ready));
        rows(interior_a_, interior_b_, y, side_->stream);                        // ... interior rows while ...
        cusp::detail::check(cmi_event_record(side_->done, side_->stream));
        exchange(stream);                                                        // ... the halo is in flight
        rows(0, interior_a_, y, stream);
        rows(interior_b_, (int64_t)local.num_rows, y, stream);
        cusp::detail::check(cmi_stream_wait_event(stream, side_->done));         // join: later work on `stream` sees all of y
    }

    void exchange_impl(void *stream, cusp::device_memory) const
    {
        if (comm_->size() == 1 && mode_ != exchange_mode::allgather) return;
        ValueType *buf = xbuf_.data.data();
        if (mode_ == exchange_mode::peer) { // correct for ANY caller, not fast: the peers' slices are complete before anyone pulls, and nobody
            fence();                        // overwrites a slice a peer is still pulling from (cg orders its pulls itself and never comes here)
            xbuf_.pull(stream);
            fence();
        } else if (mode_ == exchange_mode::halo) {
            if (comm_->staged()) { // rehearsal transport (ranks sharing one GPU): through the host
                const int world = comm_->size();
                std::vector<size_t> sl(world, 0), sn(world, 0), rl(world, 0), rn(world, 0);
                for (size_t i = 0; i < peers_.size(); i++) {
                    const int p = peers_[i];
                    sl[p] = (size_t)send_lo_[i] * sizeof(ValueType); sn[p] = (size_t)send_n_[i] * sizeof(ValueType);
                    rl[p] = (size_t)recv_lo_[i] * sizeof(ValueType); rn[p] = (size_t)recv_n_[i] * sizeof(ValueType);
                }
                comm_->staged_exchange(buf, xbuf_.data.size(), sl.data(), sn.data(), rl.data(), rn.data(), stream);
            } else
                cusp::detail::check(detail::halo_call(comm_->device(), buf, (int)peers_.size(), peers_.data(), send_lo_.data(), send_n_.data(), recv_lo_.data(), recv_n_.data(), stream));
        } else if (uniform_) {
            comm_->allgather(buf + row_begin(), buf, (size_t)count_, cusp::device_memory(), stream); // (a 1-rank world still goes through RCCL)
        } else {
            comm_->allgatherv(buf + row_begin(), buf, counts_.data(), displs_.data(), cusp::device_memory(), allgatherv_algorithm(), stream);
        }
    }
    void exchange_impl(void *, cusp::host_memory) const
    {
        if (comm_->size() == 1) return;
        ValueType *buf = xbuf_.data.data();
        if (mode_ == exchange_mode::halo) {
            const int world = comm_->size();
            std::vector<size_t> sl(world, 0), sn(world, 0), rl(world, 0), rn(world, 0);
            for (size_t i = 0; i < peers_.size(); i++) {
                const int p = peers_[i];
                sl[p] = (size_t)send_lo_[i] * sizeof(ValueType); sn[p] = (size_t)send_n_[i] * sizeof(ValueType);
                rl[p] = (size_t)recv_lo_[i] * sizeof(ValueType); rn[p] = (size_t)recv_n_[i] * sizeof(ValueType);
            }
            comm_->host().exchange(buf, sl.data(), sn.data(), rl.data(), rn.data());
        } else {
            comm_->allgatherv(buf + row_begin(), buf, counts_.data(), displs_.data(), cusp::host_memory());
        }
    }
    static int allgatherv_algorithm()
    {
        static const int a = [] { const char *e = std::getenv("CMI_ALLGATHERV"); return e ? std::atoi(e) : 1; }();
        return a;
    }
};

// ---- builders -----------------------------------------------------------------------------------------------------------------
// cusp::gallery::poisson5pt(m, n) (reference cusp/gallery/detail/poisson.inl:29-47), row-block sharded: every rank generates ITS
// rows only -- on the device straight into HBM (cmi_poisson5pt_csr_*: 1e8-row matrices never exist in one piece anywhere), on the
// host with the same stencil loop.
namespace detail {
inline int poisson_block(int64_t m, int64_t n, int64_t r0, int64_t r1, int *Ap, int *Aj, double *Ax) { return cmi_poisson5pt_csr_f64(m, n, r0, r1, Ap, Aj, Ax, nullptr); }
inline int poisson_block(int64_t m, int64_t n, int64_t r0, int64_t r1, int *Ap, int *Aj, float *Ax) { return cmi_poisson5pt_csr_f32(m, n, r0, r1, Ap, Aj, Ax, nullptr); }
template <typename V> void poisson_rows(cusp::csr_matrix<int, V, cusp::device_memory> &b, size_t m, size_t n, size_t r0, size_t r1)
{
    const size_t nnz = (size_t)cmi_poisson5pt_shard_entries((int64_t)m, (int64_t)n, (int64_t)r0, (int64_t)r1);
    b.resize(r1 - r0, m * n, nnz);
    cusp::detail::check(poisson_block((int64_t)m, (int64_t)n, (int64_t)r0, (int64_t)r1, b.row_offsets.data(), b.column_indices.data(), b.values.data()));
}
template <typename I, typename V> void poisson_rows(cusp::csr_matrix<I, V, cusp::host_memory> &b, size_t m, size_t n, size_t r0, size_t r1)
{
    // grid point (ix, iy), row = iy * m + ix; entries in ascending column order: -m, -1, 0, +1, +m (stencil.inl:143-206)
    std::vector<I> Ap(1, 0), Aj;
    std::vector<V> Ax;
    for (size_t r = r0; r < r1; r++) {
        const size_t ix = r % m, iy = r / m;
        if (iy > 0) { Aj.push_back((I)(r - m)); Ax.push_back(V(-1)); }
        if (ix > 0) { Aj.push_back((I)(r - 1)); Ax.push_back(V(-1)); }
        Aj.push_back((I)r); Ax.push_back(V(4));
        if (ix + 1 < m) { Aj.push_back((I)(r + 1)); Ax.push_back(V(-1)); }
        if (iy + 1 < n) { Aj.push_back((I)(r + m)); Ax.push_back(V(-1)); }
        Ap.push_back((I)Aj.size());
    }
    b.resize(r1 - r0, m * n, Aj.size());
    for (size_t i = 0; i < Ap.size(); i++) b.row_offsets[i] = Ap[i];
    for (size_t k = 0; k < Aj.size(); k++) { b.column_indices[k] = Aj[k]; b.values[k] = Ax[k]; }
}
} // namespace detail

template <typename I, typename V, typename L>
void poisson5pt(csr_matrix<I, V, L> &A, size_t m, size_t n, exchange_mode want = exchange_mode::automatic)
{
    communicator &c = A.comm();
    const std::vector<int64_t> cuts = partition_rows((int64_t)(m * n), c.size()); // (5-point rows: balanced by rows = balanced by entries)
    typename csr_matrix<I, V, L>::local_matrix_type block;
    detail::poisson_rows(block, m, n, (size_t)cuts[c.rank()], (size_t)cuts[c.rank() + 1]);
    A.assemble(std::move(block), m * n, cuts, want);
}

} // namespace distributed
} // namespace cusp
