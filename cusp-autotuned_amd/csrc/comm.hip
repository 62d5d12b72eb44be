// comm.hip -- the communicator and collectives of the row-block sharded SpMV / CG behind the C-ABI (cmi_allgather_f64,
// cmi_allreduce_f64; 8(e): one process per GPU, RCCL over xGMI).  The reference has no distributed code at all (its KTT layer is
// pinned to device 0, cusp/ktt/detail/ktt.inl:34-35): this is new design, and it lives here -- not in a Python file -- so that the
// header-only C++ layer (cusp/distributed/*.h) and any other host language reach it through the same thin boundary as the kernels.
//
// RCCL is bound at RUN time (dlopen "librccl.so.1" at the first cmi_comm_* call): a single-GPU user never loads it, the library
// keeps loading on a box without it, and a process that already carries an RCCL (PyTorch bundles one) shares that copy instead of
// getting a second set of the same symbols.  Every collective is enqueued on the CALLER's stream -- RCCL orders it behind the
// kernels already queued there and the next SpMV behind it; nothing here synchronises except cmi_comm_create / destroy / barrier
// and the host-buffer convenience cmi_comm_allgather_host.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <rccl/rccl.h>

namespace cmi {

struct rccl_api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

static rccl_api g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_error[256] = "";

static void load_rccl()
{
    const char *names[] = {std::getenv("CMI_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !n[0]) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl.so.1 could not be loaded: %s", dlerror()); return; }
#define CMI_RCCL_SYM(field, name)                                                                      \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                           \
    if (!g_rccl.field) { snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl: no symbol %s", name); return; }
    CMI_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    CMI_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    CMI_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    CMI_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    CMI_RCCL_SYM(AllGather, "ncclAllGather")
    CMI_RCCL_SYM(AllReduce, "ncclAllReduce")
    CMI_RCCL_SYM(Broadcast, "ncclBroadcast")
    CMI_RCCL_SYM(Send, "ncclSend")
    CMI_RCCL_SYM(Recv, "ncclRecv")
    CMI_RCCL_SYM(GroupStart, "ncclGroupStart")
    CMI_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    CMI_RCCL_SYM(GetVersion, "ncclGetVersion")
#undef CMI_RCCL_SYM
    g_rccl.handle = h;
}

static int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) { set_error("%s", g_rccl_error); return CMI_ERROR_COMM; }
    return CMI_SUCCESS;
}

static int rccl_fail(ncclResult_t r, const char *what)
{
    set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return CMI_ERROR_COMM;
}
#define CMI_RCCL(call, what)                                     \
    do {                                                         \
        ncclResult_t r__ = (call);                               \
        if (r__ != ncclSuccess) return ::cmi::rccl_fail(r__, what); \
    } while (0)

} // namespace cmi

struct cmi_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    double *scratch = nullptr; // 256 device bytes: the barrier's token and the staging of cmi_comm_allgather_host's small records
    size_t scratch_bytes = 0;
};

using namespace cmi;

CMI_API int cmi_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_unique_id: null buffer");
    if (int st = need_rccl()) return st;
    static_assert(sizeof(ncclUniqueId) == CMI_COMM_ID_BYTES, "CMI_COMM_ID_BYTES must be RCCL's unique-id size");
    ncclUniqueId id;
    CMI_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return CMI_SUCCESS;
}

CMI_API int cmi_comm_create(const void *unique_id, int rank, int world, cmi_comm **comm_out)
{
    if (!comm_out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_create: null result pointer");
    *comm_out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_create: bad id, rank or world size");
    if (int st = need_rccl()) return st;
    cmi_comm *c = new (std::nothrow) cmi_comm;
    if (!c) return fail(CMI_ERROR_ALLOC, "cmi_comm_create: out of host memory");
    c->rank = rank;
    c->world = world;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { delete c; return hip_fail(e, "cmi_comm_create: hipGetDevice"); }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank); // collective: every rank of the world calls it
    if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    c->scratch_bytes = 4096;
    e = hipMalloc((void **)&c->scratch, c->scratch_bytes);
    if (e == hipSuccess) e = hipMemset(c->scratch, 0, c->scratch_bytes);
    if (e != hipSuccess) { (void)g_rccl.CommDestroy(c->nccl); delete c; return hip_fail(e, "cmi_comm_create: scratch"); }
    *comm_out = c;
    return CMI_SUCCESS;
}

CMI_API int cmi_comm_destroy(cmi_comm *comm)
{
    if (!comm) return CMI_SUCCESS;
    int st = CMI_SUCCESS;
    if (comm->scratch) (void)hipFree(comm->scratch);
    if (comm->nccl && g_rccl.CommDestroy) {
        const ncclResult_t r = g_rccl.CommDestroy(comm->nccl);
        if (r != ncclSuccess) st = rccl_fail(r, "ncclCommDestroy");
    }
    delete comm;
    return st;
}

CMI_API int cmi_comm_rank(const cmi_comm *comm, int *rank, int *world)
{
    if (!comm) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_rank: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return CMI_SUCCESS;
}

CMI_API int cmi_comm_library_version(int *version)
{
    if (!version) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_library_version: null result");
    if (int st = need_rccl()) return st;
    CMI_RCCL(g_rccl.GetVersion(version), "ncclGetVersion");
    return CMI_SUCCESS;
}

namespace cmi {
static int check_comm(const cmi_comm *c, const char *who)
{
    if (!c || !c->nccl) { set_error("%s: null communicator", who); return CMI_ERROR_INVALID_VALUE; }
    return CMI_SUCCESS;
}

// recv[r * count, (r + 1) * count) <- rank r's send[0, count).  In place when send == recv + rank * count (RCCL's convention).
static int allgather(cmi_comm *c, const void *send, void *recv, int64_t count, ncclDataType_t t, hipStream_t s)
{
    if (int st = check_comm(c, "cmi_allgather")) return st;
    if (count < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allgather: negative count");
    if (count == 0) return CMI_SUCCESS;
    if (!send || !recv) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allgather: null buffer");
    CMI_RCCL(g_rccl.AllGather(send, recv, (size_t)count, t, c->nccl, s), "ncclAllGather");
    return CMI_SUCCESS;
}

// Unequal pieces: recv[displs[r], displs[r] + counts[r]) <- rank r's send[0, counts[r]).  Two transports, same result:
//   algo 0  one ncclBroadcast per rank inside ONE group (RCCL fuses them into a single launch)
//   algo 1  direct exchange: a grouped ncclSend to / ncclRecv from every peer -- on the point-to-point xGMI mesh each of the 7 links
//           of a GPU then carries exactly one peer's piece in each direction, which is the all-gather's lower bound on that fabric
// The rank's own piece is a device copy when send is not already in place.
static int allgatherv(cmi_comm *c, const void *send, void *recv, const int64_t *counts, const int64_t *displs, size_t esize, ncclDataType_t t,
                      int algo, hipStream_t s)
{
    if (int st = check_comm(c, "cmi_allgatherv")) return st;
    if (!counts || !displs) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allgatherv: null counts / displacements");
    for (int r = 0; r < c->world; r++)
        if (counts[r] < 0 || displs[r] < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allgatherv: negative count or displacement");
    if (!recv || (counts[c->rank] > 0 && !send)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allgatherv: null buffer");
    char *rbase = static_cast<char *>(recv);
    char *mine = rbase + (size_t)displs[c->rank] * esize;
    if (counts[c->rank] > 0 && mine != send)
        CMI_HIP(hipMemcpyAsync(mine, send, (size_t)counts[c->rank] * esize, hipMemcpyDeviceToDevice, s));
    if (c->world == 1) return CMI_SUCCESS;
    CMI_RCCL(g_rccl.GroupStart(), "ncclGroupStart");
    ncclResult_t r = ncclSuccess;
    if (algo == 0) {
        for (int p = 0; p < c->world && r == ncclSuccess; p++)
            if (counts[p] > 0) r = g_rccl.Broadcast(rbase + (size_t)displs[p] * esize, rbase + (size_t)displs[p] * esize, (size_t)counts[p], t, p, c->nccl, s);
    } else {
        for (int p = 0; p < c->world && r == ncclSuccess; p++) {
            if (p == c->rank) continue;
            if (counts[c->rank] > 0) r = g_rccl.Send(mine, (size_t)counts[c->rank], t, p, c->nccl, s);
            if (r == ncclSuccess && counts[p] > 0) r = g_rccl.Recv(rbase + (size_t)displs[p] * esize, (size_t)counts[p], t, p, c->nccl, s);
        }
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "cmi_allgatherv: enqueue");
    if (r2 != ncclSuccess) return rccl_fail(r2, "ncclGroupEnd");
    return CMI_SUCCESS;
}

// Two-sided halo exchange inside ONE full-length buffer indexed by global column: for every listed peer, send
// x_full[send_lo, send_lo + send_count) (part of this rank's own slice) and receive x_full[recv_lo, recv_lo + recv_count) (part of
// the peer's slice), all in one group = one launch.
static int halo(cmi_comm *c, void *x_full, int npeers, const int *peers, const int64_t *send_lo, const int64_t *send_count, const int64_t *recv_lo,
                const int64_t *recv_count, size_t esize, ncclDataType_t t, hipStream_t s)
{
    if (int st = check_comm(c, "cmi_halo_exchange")) return st;
    if (npeers < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_halo_exchange: negative peer count");
    if (npeers == 0) return CMI_SUCCESS;
    if (!x_full || !peers || !send_lo || !send_count || !recv_lo || !recv_count) return fail(CMI_ERROR_INVALID_VALUE, "cmi_halo_exchange: null array");
    for (int i = 0; i < npeers; i++) {
        if (peers[i] < 0 || peers[i] >= c->world || peers[i] == c->rank) return fail(CMI_ERROR_INVALID_VALUE, "cmi_halo_exchange: a peer is out of range or this rank itself");
        if (send_lo[i] < 0 || send_count[i] < 0 || recv_lo[i] < 0 || recv_count[i] < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_halo_exchange: negative range");
    }
    char *base = static_cast<char *>(x_full);
    CMI_RCCL(g_rccl.GroupStart(), "ncclGroupStart");
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < npeers && r == ncclSuccess; i++) {
        if (send_count[i] > 0) r = g_rccl.Send(base + (size_t)send_lo[i] * esize, (size_t)send_count[i], t, peers[i], c->nccl, s);
        if (r == ncclSuccess && recv_count[i] > 0) r = g_rccl.Recv(base + (size_t)recv_lo[i] * esize, (size_t)recv_count[i], t, peers[i], c->nccl, s);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "cmi_halo_exchange: enqueue");
    if (r2 != ncclSuccess) return rccl_fail(r2, "ncclGroupEnd");
    return CMI_SUCCESS;
}
} // namespace cmi

CMI_API int cmi_allgather_f64(cmi_comm *comm, const double *send, double *recv, int64_t count, void *stream)
{
    return cmi::allgather(comm, send, recv, count, ncclDouble, as_stream(stream));
}
CMI_API int cmi_allgather_f32(cmi_comm *comm, const float *send, float *recv, int64_t count, void *stream)
{
    return cmi::allgather(comm, send, recv, count, ncclFloat, as_stream(stream));
}
CMI_API int cmi_allgatherv_f64(cmi_comm *comm, const double *send, double *recv, const int64_t *counts, const int64_t *displs, int algorithm, void *stream)
{
    return cmi::allgatherv(comm, send, recv, counts, displs, sizeof(double), ncclDouble, algorithm, as_stream(stream));
}
CMI_API int cmi_allgatherv_f32(cmi_comm *comm, const float *send, float *recv, const int64_t *counts, const int64_t *displs, int algorithm, void *stream)
{
    return cmi::allgatherv(comm, send, recv, counts, displs, sizeof(float), ncclFloat, algorithm, as_stream(stream));
}
CMI_API int cmi_halo_exchange_f64(cmi_comm *comm, double *x_full, int npeers, const int *peers, const int64_t *send_lo, const int64_t *send_count,
                                  const int64_t *recv_lo, const int64_t *recv_count, void *stream)
{
    return cmi::halo(comm, x_full, npeers, peers, send_lo, send_count, recv_lo, recv_count, sizeof(double), ncclDouble, as_stream(stream));
}
CMI_API int cmi_halo_exchange_f32(cmi_comm *comm, float *x_full, int npeers, const int *peers, const int64_t *send_lo, const int64_t *send_count,
                                  const int64_t *recv_lo, const int64_t *recv_count, void *stream)
{
    return cmi::halo(comm, x_full, npeers, peers, send_lo, send_count, recv_lo, recv_count, sizeof(float), ncclFloat, as_stream(stream));
}

// recv[i] <- op over the ranks of send[i] (i < count; device buffers; in place when send == recv).  CG's <.,.> scalars: count = 1 or 2,
// the values never leave device memory.  RCCL reduces in a fixed order for a fixed world size and algorithm: the same inputs give the
// same bits on every rank and every run.
CMI_API int cmi_allreduce_f64(cmi_comm *comm, const double *send, double *recv, int64_t count, int op, void *stream)
{
    if (int st = check_comm(comm, "cmi_allreduce_f64")) return st;
    if (count < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allreduce_f64: negative count");
    if (count == 0) return CMI_SUCCESS;
    if (!send || !recv) return fail(CMI_ERROR_INVALID_VALUE, "cmi_allreduce_f64: null buffer");
    ncclRedOp_t o;
    switch (op) {
    case CMI_OP_SUM: o = ncclSum; break;
    case CMI_OP_MAX: o = ncclMax; break;
    case CMI_OP_MIN: o = ncclMin; break;
    default: return fail(CMI_ERROR_INVALID_VALUE, "cmi_allreduce_f64: op must be CMI_OP_SUM, CMI_OP_MAX or CMI_OP_MIN");
    }
    CMI_RCCL(g_rccl.AllReduce(send, recv, (size_t)count, ncclDouble, o, comm->nccl, as_stream(stream)), "ncclAllReduce");
    return CMI_SUCCESS;
}

// every rank has reached this point AND everything queued on `stream` before it has completed on every rank (an all-reduce of one
// token, then a stream synchronisation): the ordering the one-sided pull needs between the peers' producers and itself
CMI_API int cmi_comm_barrier(cmi_comm *comm, void *stream)
{
    if (int st = check_comm(comm, "cmi_comm_barrier")) return st;
    CMI_RCCL(g_rccl.AllReduce(comm->scratch, comm->scratch, 1, ncclDouble, ncclSum, comm->nccl, as_stream(stream)), "ncclAllReduce (barrier)");
    CMI_HIP(hipStreamSynchronize(as_stream(stream)));
    return CMI_SUCCESS;
}

// Set-up convenience for small HOST records (IPC handles, column spans, row counts): recv_host[r * bytes, (r + 1) * bytes) <- rank r's
// send_host[0, bytes).  Staged through device memory and ncclAllGather on `stream`; synchronises.  bytes * world <= 2048.
CMI_API int cmi_comm_allgather_host(cmi_comm *comm, const void *send_host, void *recv_host, size_t bytes, void *stream)
{
    if (int st = check_comm(comm, "cmi_comm_allgather_host")) return st;
    if (bytes == 0) return CMI_SUCCESS;
    if (!send_host || !recv_host) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_allgather_host: null buffer");
    const size_t pad = (bytes + 7) & ~(size_t)7; // whole 8-byte words per rank
    if (pad * (size_t)comm->world + 8 > comm->scratch_bytes - 64) return fail(CMI_ERROR_INVALID_VALUE, "cmi_comm_allgather_host: record too large (bytes x world <= ~4000)");
    hipStream_t s = as_stream(stream);
    char *stage = reinterpret_cast<char *>(comm->scratch) + 64; // (the first line holds the barrier token)
    CMI_HIP(hipMemcpyAsync(stage + pad * comm->rank, send_host, bytes, hipMemcpyHostToDevice, s));
    CMI_RCCL(g_rccl.AllGather(stage + pad * comm->rank, stage, pad, ncclChar, comm->nccl, s), "ncclAllGather (host records)");
    CMI_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < comm->world; r++)
        CMI_HIP(hipMemcpy(static_cast<char *>(recv_host) + bytes * r, stage + pad * r, bytes, hipMemcpyDeviceToHost));
    return CMI_SUCCESS;
}
