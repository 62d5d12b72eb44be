// cusp/distributed/transport.h -- the out-of-band channel of one-process-per-GPU jobs: a TCP star through rank 0.
//
// Two jobs.  (1) BOOTSTRAP: rank 0 makes the RCCL unique id (cmi_comm_unique_id) and every other rank needs it before
// cmi_comm_create -- the reference has nothing of the kind (single process, device 0: cusp/ktt/detail/ktt.inl:34-35), MPI is not in
// the image, so the header layer carries its own forty lines of sockets, addressed like torchrun addresses its store
// (MASTER_ADDR / MASTER_PORT).  (2) HOST TRANSPORT: the collectives of a sharded operator whose vectors live in host_memory
// (all-gather, all-gather of unequal pieces, ranged exchange, all-reduce) -- the world-size-2 CPU tests run the same sharding,
// exchange-plan and CG logic through it that device_memory runs through RCCL.  Set-up and test infrastructure: a star through
// rank 0, blocking sockets, reductions summed in rank order (deterministic).  Never on the device path.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <poll.h>
#include <cerrno>
#include <string>
#include <thread>
#include <vector>

#include "../exception.h"

namespace cusp {
namespace distributed {
namespace detail {

class tcp_star {
public:
    tcp_star(int rank, int world, const std::string &addr, int port, double timeout_s = 120.0) : rank_(rank), world_(world)
    {
        if (world <= 1) return;
        if (rank == 0) {
            const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
            if (ls < 0) fail("socket");
            int one = 1;
            ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
            sockaddr_in sa{};
            sa.sin_family = AF_INET;
            sa.sin_addr.s_addr = htonl(INADDR_ANY);
            sa.sin_port = htons((uint16_t)port);
            if (::bind(ls, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) != 0) { ::close(ls); fail("bind (is the port in use? set CMI_PORT)"); }
            if (::listen(ls, world) != 0) { ::close(ls); fail("listen"); }
            peers_.assign(world, -1);
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 1; k < world; k++) {
                // a peer that died before connecting must not leave rank 0 in accept() for ever: wait for the listening socket with the
                // same budget the connecting side has
                for (;;) {
                    const double left = timeout_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (left <= 0.0) { ::close(ls); fail("accept: a peer did not connect in time"); }
                    pollfd pf{};
                    pf.fd = ls;
                    pf.events = POLLIN;
                    const int pr = ::poll(&pf, 1, (int)((left < 1.0 ? left : 1.0) * 1000.0) + 1);
                    if (pr > 0) break;
                    if (pr < 0 && errno != EINTR) { ::close(ls); fail("poll"); }
                }
                const int fd = ::accept(ls, nullptr, nullptr);
                if (fd < 0) { ::close(ls); fail("accept"); }
                nodelay(fd);
                int32_t who = -1;
                recv_all(fd, &who, sizeof(who));
                if (who < 1 || who >= world || peers_[who] >= 0) { ::close(ls); fail("a peer announced a bad rank"); }
                peers_[who] = fd;
            }
            ::close(ls);
        } else {
            addrinfo hints{}, *res = nullptr;
            hints.ai_family = AF_INET;
            hints.ai_socktype = SOCK_STREAM;
            const std::string p = std::to_string(port);
            if (::getaddrinfo(addr.c_str(), p.c_str(), &hints, &res) != 0 || !res) fail("getaddrinfo(MASTER_ADDR)");
            const auto t0 = std::chrono::steady_clock::now();
            int fd = -1;
            for (;;) { // rank 0 may not be listening yet
                fd = ::socket(AF_INET, SOCK_STREAM, 0);
                if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
                if (fd >= 0) ::close(fd);
                fd = -1;
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
            ::freeaddrinfo(res);
            if (fd < 0) fail("connect to rank 0 timed out");
            nodelay(fd);
            const int32_t who = rank;
            send_all(fd, &who, sizeof(who));
            root_ = fd;
        }
    }
    ~tcp_star()
    {
        for (int fd : peers_) if (fd >= 0) ::close(fd);
        if (root_ >= 0) ::close(root_);
    }
    tcp_star(const tcp_star &) = delete;
    tcp_star &operator=(const tcp_star &) = delete;

    int rank() const { return rank_; }
    int size() const { return world_; }

    void broadcast(void *buf, size_t bytes, int root = 0)
    {
        if (world_ <= 1 || bytes == 0) return;
        if (root != 0) { // through rank 0
            if (rank_ == root) send_all(root_, buf, bytes);
            if (rank_ == 0) recv_all(peers_[root], buf, bytes);
        }
        if (rank_ == 0) { for (int r = 1; r < world_; r++) send_all(peers_[r], buf, bytes); }
        else recv_all(root_, buf, bytes);
    }
    void barrier() { double token = 0.0; allreduce(&token, 1, 0); } // every rank -> rank 0 -> every rank: nobody leaves before all have arrived

    // recv[r * bytes, +bytes) <- rank r's send
    void allgather(const void *send, void *recv, size_t bytes)
    {
        std::vector<size_t> counts(world_, bytes), displs(world_);
        for (int r = 0; r < world_; r++) displs[r] = (size_t)r * bytes;
        allgatherv(send, recv, counts.data(), displs.data());
    }
    // recv[displs[r], +counts[r]) <- rank r's send[0, counts[r])   (bytes; send may alias its place in recv)
    void allgatherv(const void *send, void *recv, const size_t *counts, const size_t *displs)
    {
        char *out = static_cast<char *>(recv);
        if (counts[rank_] && out + displs[rank_] != send) std::memmove(out + displs[rank_], send, counts[rank_]);
        if (world_ <= 1) { sync_token(); return; }
        if (rank_ == 0) {
            for (int r = 1; r < world_; r++) if (counts[r]) recv_all(peers_[r], out + displs[r], counts[r]);
            for (int r = 1; r < world_; r++)
                for (int q = 0; q < world_; q++) if (q != r && counts[q]) send_all(peers_[r], out + displs[q], counts[q]);
        } else {
            if (counts[rank_]) send_all(root_, out + displs[rank_], counts[rank_]);
            for (int q = 0; q < world_; q++) if (q != rank_ && counts[q]) recv_all(root_, out + displs[q], counts[q]);
        }
        sync_token();
    }
    // ranged exchange inside one buffer indexed alike on every rank (the halo exchange): this rank SENDS buf[send_lo[p], +send_n[p]) to
    // peer p and RECEIVES buf[recv_lo[p], +recv_n[p]) from it (bytes; arrays of `world` entries, own entry ignored).  Routed by rank 0.
    void exchange(void *buf, const size_t *send_lo, const size_t *send_n, const size_t *recv_lo, const size_t *recv_n)
    {
        if (world_ <= 1) return;
        char *b = static_cast<char *>(buf);
        if (rank_ == 0) {
            // pieces addressed to rank 0 land directly; pieces between other ranks are relayed
            std::vector<std::vector<std::vector<char>>> relay(world_, std::vector<std::vector<char>>(world_));
            for (int src = 1; src < world_; src++) {
                std::vector<uint64_t> n(world_);
                recv_all(peers_[src], n.data(), n.size() * sizeof(uint64_t));
                for (int dst = 0; dst < world_; dst++) {
                    if (dst == src || n[dst] == 0) continue;
                    if (dst == 0) {
                        if (n[dst] != recv_n[src]) fail("halo exchange: rank 0 expected a different length");
                        recv_all(peers_[src], b + recv_lo[src], n[dst]);
                    } else {
                        relay[dst][src].resize(n[dst]);
                        recv_all(peers_[src], relay[dst][src].data(), n[dst]);
                    }
                }
            }
            for (int dst = 1; dst < world_; dst++) {
                std::vector<uint64_t> n(world_, 0);
                for (int src = 0; src < world_; src++) n[src] = src == 0 ? send_n[dst] : relay[dst][src].size();
                n[dst] = 0;
                send_all(peers_[dst], n.data(), n.size() * sizeof(uint64_t));
                for (int src = 0; src < world_; src++) {
                    if (src == dst || n[src] == 0) continue;
                    if (src == 0) send_all(peers_[dst], b + send_lo[dst], n[src]);
                    else send_all(peers_[dst], relay[dst][src].data(), n[src]);
                }
            }
        } else {
            std::vector<uint64_t> n(world_, 0);
            for (int dst = 0; dst < world_; dst++) if (dst != rank_) n[dst] = send_n[dst];
            send_all(root_, n.data(), n.size() * sizeof(uint64_t));
            for (int dst = 0; dst < world_; dst++) if (dst != rank_ && n[dst]) send_all(root_, b + send_lo[dst], n[dst]);
            std::vector<uint64_t> m(world_, 0);
            recv_all(root_, m.data(), m.size() * sizeof(uint64_t));
            for (int src = 0; src < world_; src++) {
                if (src == rank_ || m[src] == 0) continue;
                if (m[src] != recv_n[src]) fail("halo exchange: a peer sent a different length than the plan expects");
                recv_all(root_, b + recv_lo[src], m[src]);
            }
        }
    }
    // v[i] <- sum (op 0) / max (1) / min (2) over the ranks, combined in RANK ORDER on rank 0: the same bits on every rank and run
    void allreduce(double *v, size_t n, int op = 0)
    {
        if (world_ <= 1 || n == 0) return;
        if (rank_ == 0) {
            std::vector<double> in(n);
            for (int r = 1; r < world_; r++) {
                recv_all(peers_[r], in.data(), n * sizeof(double));
                for (size_t i = 0; i < n; i++) v[i] = op == 0 ? v[i] + in[i] : op == 1 ? (in[i] > v[i] ? in[i] : v[i]) : (in[i] < v[i] ? in[i] : v[i]);
            }
            for (int r = 1; r < world_; r++) send_all(peers_[r], v, n * sizeof(double));
        } else {
            send_all(root_, v, n * sizeof(double));
            recv_all(root_, v, n * sizeof(double));
        }
    }

private:
    int rank_, world_, root_ = -1;
    std::vector<int> peers_; // rank 0: socket of every other rank

    void sync_token() {}
    static void nodelay(int fd) { int one = 1; ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }
    [[noreturn]] static void fail(const char *what) { throw cusp::runtime_exception(std::string("cusp::distributed transport: ") + what + ": " + std::strerror(errno)); }
    static void send_all(int fd, const void *p, size_t n)
    {
        const char *c = static_cast<const char *>(p);
        while (n) {
            const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
            if (k <= 0) { if (k < 0 && errno == EINTR) continue; fail("send (a peer has gone)"); }
            c += k; n -= (size_t)k;
        }
    }
    static void recv_all(int fd, void *p, size_t n)
    {
        char *c = static_cast<char *>(p);
        while (n) {
            const ssize_t k = ::recv(fd, c, n, 0);
            if (k <= 0) { if (k < 0 && errno == EINTR) continue; fail("recv (a peer has gone)"); }
            c += k; n -= (size_t)k;
        }
    }
};

} // namespace detail
} // namespace distributed
} // namespace cusp
