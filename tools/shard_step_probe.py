"""How much does the host side of ONE sharded SpMV step cost?  (tools, not the product)

On the one-GPU box a 1-rank RCCL communicator stands in for the neighbours: the rank sends its two
halo slices to itself (ncclSend/ncclRecv to self inside a group -- the same host path and the same
device-side launches as the N>1 step, only the wire is missing), multiplies the interior rows while
they are in flight and the boundary rows afterwards, exactly as ShardedCsr.multiply does.

Prints us/step for: plain SpMV, the eager sharded step, and the sharded step replayed from a HIP graph
(torch.cuda.CUDAGraph capture of the RCCL ops + the three cmi launches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    import cusp_autotuned_amd as cmi
    from cusp_autotuned_amd import binding as B
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    m = 3162
    A = cmi.poisson5pt(m, m, "csr")
    n = A.num_rows
    x = cmi.fill_x(n).cuda()
    y = torch.empty(n, dtype=torch.float64, device=dev)
    halo_lo = torch.zeros(m, dtype=torch.float64, device=dev)
    halo_hi = torch.zeros(m, dtype=torch.float64, device=dev)
    cfg = B.tuning_select(B.FORMAT_CSR, B.F64, n, n, A.num_entries)
    a, b = m, n - m

    def rows(lo, hi):
        B.spmv_csr(hi - lo, n, A.row_offsets[lo:hi + 1], A.column_indices, A.values, x, y[lo:hi], cfg=cfg)

    ops = [dist.P2POp(dist.isend, x[:m], 0), dist.P2POp(dist.isend, x[n - m:], 0),
           dist.P2POp(dist.irecv, halo_lo, 0), dist.P2POp(dist.irecv, halo_hi, 0)]

    def plain():
        B.spmv_csr(n, n, A.row_offsets, A.column_indices, A.values, x, y, cfg=cfg)

    def sharded():
        works = dist.batch_isend_irecv(ops)
        rows(a, b)
        for w in works:
            w.wait()
        rows(0, a)
        rows(b, n)

    def timeit(f, steps=300):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6, t_host / steps * 1e6

    def three_launches():
        rows(a, b)
        rows(0, a)
        rows(b, n)

    def comm_only():
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def comm_then_whole():
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        plain()

    gather_in = torch.zeros(2 * m, dtype=torch.float64, device=dev)
    gather_out = torch.zeros(2 * m, dtype=torch.float64, device=dev)

    def small_allgather_then_whole():
        gather_in[:m].copy_(x[:m])
        gather_in[m:].copy_(x[n - m:])
        dist.all_gather_into_tensor(gather_out, gather_in)
        plain()

    comm_stream = torch.cuda.Stream()
    ev_ready, ev_done = torch.cuda.Event(), torch.cuda.Event()

    def copy_kernel_overlap():
        # what a one-sided halo pull would cost on the device: two 25 KB copies on a side stream
        # ordered by events, interior rows meanwhile, boundary rows after
        ev_ready.record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev_ready)
            halo_lo.copy_(x[:m])
            halo_hi.copy_(x[n - m:])
            ev_done.record()
        rows(a, b)
        torch.cuda.current_stream().wait_event(ev_done)
        rows(0, a)
        rows(b, n)

    def copy_kernel_inline():
        halo_lo.copy_(x[:m])
        halo_hi.copy_(x[n - m:])
        plain()

    pull = B.CopyRanges([(x.data_ptr(), halo_lo.data_ptr(), m * 8), (x.data_ptr() + (n - m) * 8, halo_hi.data_ptr(), m * 8)])

    def one_sided_pull_then_whole():
        # the device-side work of the "peer" exchange: ONE cmi_copy_ranges launch for both halos on the
        # compute stream, then the whole-block SpMV (here the sources are local; over xGMI they are the
        # neighbours' mapped buffers)
        pull.launch()
        plain()

    for name, f in (("plain", plain), ("sharded (p2p, overlapped)", sharded), ("three launches, no comm", three_launches),
                    ("p2p only", comm_only), ("p2p then whole SpMV", comm_then_whole),
                    ("small all-gather then whole", small_allgather_then_whole),
                    ("copy kernels on side stream, overlapped", copy_kernel_overlap),
                    ("copy kernels inline then whole", copy_kernel_inline),
                    ("one-sided pull (cmi_copy_ranges) then whole", one_sided_pull_then_whole)):
        only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
        if only and not any(o in name for o in only):
            continue
        print("%-46s: %7.1f us/step (host enqueue %6.1f us)" % ((name,) + timeit(f)), flush=True)
    if "--graph" not in sys.argv:
        dist.destroy_process_group()
        return
    assert torch.equal(halo_lo, x[:m]) and torch.equal(halo_hi, x[n - m:])
    y_ref = y.clone()

    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                sharded()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sharded()
        torch.cuda.synchronize()
        y.zero_()
        halo_lo.zero_()
        g.replay()
        torch.cuda.synchronize()
        ok = torch.equal(y, y_ref) and torch.equal(halo_lo, x[:m])
        print("graph   : %7.1f us/step (host enqueue %6.1f us)  replay correct: %s" % (*timeit(g.replay), ok), flush=True)
    except Exception as e:  # noqa: BLE001
        print("graph capture failed:", type(e).__name__, str(e)[:300], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
