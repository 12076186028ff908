"""Worker for the transport tests: every collective of the Comm interface on odd-sized
payloads (larger than one staging chunk), results checked against closed forms; exits
non-zero on the first mismatch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

from distributed_machine_learning_amd.parallel.comm import RcclCommunicator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--master-ip", default="127.0.0.1:29451")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--num-nodes", type=int, default=1)
    ap.add_argument("--transport", default="loopback")
    ns = ap.parse_args()
    r, w = ns.rank, ns.num_nodes
    index = int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()
    dev = torch.device("cuda", index)
    torch.cuda.set_device(dev)
    comm = RcclCommunicator(ns.master_ip, r, w, index, 60.0, transport=ns.transport)
    assert comm.peer_count == w
    side = torch.cuda.Stream(device=dev)
    n = (4 << 20) // 4 + 1237          # one chunk and a bit, in floats

    base = torch.arange(n, device=dev, dtype=torch.float32) % 1000
    # all_reduce fp32 (sum and mean), issued on two different streams back to back
    a = base + r
    b = (base * 2 + r).contiguous()
    comm.all_reduce(a, "sum")
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce(b, "mean", stream=side)
    torch.cuda.current_stream().wait_stream(side)
    comm.guard_sync(60)
    assert torch.equal(a, base * w + sum(range(w))), "all_reduce sum"
    assert torch.allclose(b, base * 2 + sum(range(w)) / w), "all_reduce mean"
    # bf16 and int64
    h = torch.full((5001,), 1.5 + r, device=dev, dtype=torch.bfloat16)
    comm.all_reduce(h)
    i64 = torch.arange(77, device=dev, dtype=torch.int64) * (r + 1)
    comm.all_reduce(i64)
    comm.guard_sync(60)
    assert torch.equal(h.float(), torch.full((5001,), 1.5 * w + sum(range(w)), device=dev)), "bf16"
    assert torch.equal(i64, torch.arange(77, device=dev) * sum(k + 1 for k in range(w))), "i64"
    # broadcast from the last rank
    t = torch.full((n,), float(r), device=dev)
    comm.broadcast(t, w - 1)
    comm.guard_sync(60)
    assert bool((t == w - 1).all()), "broadcast"
    # gather on rank 0, scatter back distinct chunks
    mine = torch.full((n,), 10.0 + r, device=dev)
    slabs = [torch.zeros(n, device=dev) for _ in range(w)] if r == 0 else None
    comm.group_start()
    comm.gather(mine, slabs, 0)
    comm.group_end()
    comm.guard_sync(60)
    if r == 0:
        for k in range(w):
            assert bool((slabs[k] == 10.0 + k).all()), f"gather slab {k}"
    out = torch.zeros(n, device=dev)
    chunks = [torch.full((n,), 100.0 + k, device=dev) for k in range(w)] if r == 0 else None
    comm.scatter(out, chunks, 0)
    comm.guard_sync(60)
    assert bool((out == 100.0 + r).all()), "scatter"
    comm.barrier(60)
    comm.destroy()
    print(f"rank {r}: collectives ok")


if __name__ == "__main__":
    main()
