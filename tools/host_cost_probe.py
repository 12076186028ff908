"""Host-side cost of the HIP calls a multi-stream schedule is made of (per call, microseconds):
kernel launches on one stream, launches alternating between two streams, event record + wait
pairs.  Run on the GPU box: python tools/host_cost_probe.py"""
import time

import torch


def main():
    dev = torch.device("cuda", 0)
    x = torch.zeros(1024, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(64)]
    n = 2000

    def timed(label, fn, per):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        host = (time.perf_counter() - t) / (n * per) * 1e6
        torch.cuda.synchronize()
        total = (time.perf_counter() - t) / (n * per) * 1e6
        print(f"{label:48s} host {host:6.2f} us/call   with drain {total:6.2f} us/call")

    def launch_one(i):
        with torch.cuda.stream(s1):
            x.add_(1.0)

    def launch_alt(i):
        with torch.cuda.stream(s1 if i & 1 else s2):
            x.add_(1.0)

    def record_wait(i):
        e = evs[i & 63]
        e.record(s1)
        s2.wait_event(e)

    def launch_record_wait_launch(i):
        e = evs[i & 63]
        with torch.cuda.stream(s1):
            x.add_(1.0)
        e.record(s1)
        s2.wait_event(e)
        with torch.cuda.stream(s2):
            x.add_(1.0)

    for _ in range(2):
        timed("kernel launch, one stream", launch_one, 1)
        timed("kernel launch, alternating two streams", launch_alt, 1)
        timed("event record + wait (no kernels between)", record_wait, 2)
        timed("launch, record, wait, launch (4 calls)", launch_record_wait_launch, 4)


if __name__ == "__main__":
    main()
