"""Do RCCL collectives / point-to-point accept tensors that view a cmi_malloc'ed DeviceBuffer (memory that
torch's caching allocator does not own)?  1-rank communicator on the one GPU of the box (tools, not product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from cusp_autotuned_amd import binding as B
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n = 1 << 20
    buf = B.DeviceBuffer(n * 8, dev)
    t = buf.tensor(torch.float64)
    src = torch.arange(n, dtype=torch.float64, device=dev)
    w = dist.all_gather_into_tensor(t, src, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    print("all_gather_into_tensor into a DeviceBuffer view:", bool(torch.equal(t, src)), flush=True)
    t[:1].fill_(3.0)
    dist.all_reduce(t[:1])
    print("all_reduce on a DeviceBuffer view:", float(t[0]), flush=True)
    halo = buf.tensor(torch.float64)[n // 2:n // 2 + 1000]
    ops = [dist.P2POp(dist.isend, t[:1000], 0), dist.P2POp(dist.irecv, halo, 0)]
    for wk in dist.batch_isend_irecv(ops):
        wk.wait()
    torch.cuda.synchronize()
    print("isend/irecv between DeviceBuffer views:", bool(torch.equal(halo, t[:1000])), flush=True)
    objs = [None]
    dist.all_gather_object(objs, (buf.ipc_handle(), 0))
    print("all_gather_object of an IPC handle:", len(objs[0][0]), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("ok", flush=True)


if __name__ == "__main__":
    main()
