"""Bitwise repeatability of one BatchNorm backward launch under contention.
usage: python tools/probes/stress_bn_slab.py <launches> [procs] [pool 0/1] [rows] [channels]

Every process runs the same launch on the same inputs over and over and counts the launches
whose outputs differ from its first one; with procs > 1 the processes share the GPU."""
import os
import subprocess
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from distributed_machine_learning_amd.ops import native  # noqa: E402


def worker(launches, pool, batch, hw, c):
    C = native.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    rows = batch * hw * hw
    out_rows = rows // 4 if pool else rows
    x = torch.randn(rows, c, device=dev).bfloat16()
    dy = (torch.randn(out_rows, c, device=dev) * 1e-3).bfloat16()
    gamma = 1 + 0.1 * torch.randn(c, device=dev)
    beta = 0.1 * torch.randn(c, device=dev)
    mean = 0.05 * torch.randn(c, device=dev)
    rstd = 1 + 0.05 * torch.rand(c, device=dev)
    ws = torch.zeros(1 << 20, device=dev)
    ring = 32
    dx = torch.zeros(ring, rows, c, device=dev, dtype=torch.bfloat16)
    dg = torch.zeros(ring, c, device=dev)
    db = torch.zeros(ring, c, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def launch(k):
        if pool:
            C.bn_relu_pool_bwd(x.data_ptr(), dy.data_ptr(), dx[k].data_ptr(), gamma.data_ptr(),
                               beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dg[k].data_ptr(),
                               db[k].data_ptr(), ws.data_ptr(), batch, hw, c, s)
        else:
            C.bn_relu_bwd(x.data_ptr(), dy.data_ptr(), dx[k].data_ptr(), gamma.data_ptr(),
                          beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dg[k].data_ptr(),
                          db[k].data_ptr(), ws.data_ptr(), rows, c, s)

    # what a communication stream does next to the compute stream: device <-> pinned host copies
    stop = []
    if os.environ.get("STRESS_COPIES", "0") != "0":
        import threading
        side = torch.cuda.Stream()
        buf = torch.zeros(int(os.environ.get("STRESS_COPY_ELEMS", 600000)), device=dev)
        pinned = torch.zeros(buf.numel()).pin_memory()

        def copier():
            with torch.cuda.stream(side):
                while not stop:
                    pinned.copy_(buf, non_blocking=True)
                    side.synchronize()
                    buf.copy_(pinned, non_blocking=True)
                    side.synchronize()

        th = threading.Thread(target=copier, daemon=True)
        th.start()
    npoll = int(os.environ.get("STRESS_POLLERS", "0"))
    if npoll:
        # waves that poll a flag word next to the kernel under test, as a waiting stream's do
        pstream = torch.cuda.Stream()
        pflag = torch.zeros(1, dtype=torch.int32, device=dev)
        C.launch_pollers(npoll, int(float(os.environ.get("STRESS_POLL_S", "20")) * 1e8),
                         pflag.data_ptr(), pstream.cuda_stream)
    flush = None
    if os.environ.get("STRESS_FLUSH", "0") != "0":
        flush = torch.zeros(128 << 20, device=dev, dtype=torch.uint8)
    launch(0)
    torch.cuda.synchronize()
    ref_dx, ref_dg, ref_db = dx[0].clone(), dg[0].clone(), db[0].clone()
    bad, cols = 0, {}
    import time
    gap_us = float(os.environ.get("STRESS_GAP_US", "0"))
    for n in range(0, launches, ring):
        for k in range(ring):
            if flush is not None:
                flush.add_(1)
            if gap_us:
                # a mostly idle GPU: the launch finds its compute units cold (clocks, caches)
                torch.cuda.synchronize()
                time.sleep(gap_us * 1e-6)
            launch(k)
        wrong_g = (dg.view(torch.int32) != ref_dg.view(torch.int32)) | \
                  (db.view(torch.int32) != ref_db.view(torch.int32))
        wrong_x = (dx.view(torch.int16) != ref_dx.view(torch.int16)).any(dim=1)
        wrong = wrong_g | wrong_x
        if bool(wrong.any()):
            for k, col in wrong.nonzero().tolist():
                bad += 1
                cols[col] = cols.get(col, 0) + 1
    stop.append(1)
    if npoll:
        pflag.fill_(1)
        torch.cuda.synchronize()
    if os.environ.get("STRESS_COPIES", "0") != "0":
        th.join()
    print(f"pid {os.getpid()}: {launches} launches, {bad} (launch, channel) pairs differ; "
          f"channels mod 8: {sorted({k % 8 for k in cols})} channels {sorted(cols)[:16]}", flush=True)


def churn(seconds):
    """Start and end GPU processes for a while: every start and exit changes the set of hardware
    queues, for which the driver takes the running queues off the GPU and puts them back."""
    import time
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        subprocess.run([sys.executable, "-c",
                        "import torch; s = torch.cuda.Stream(); x = torch.zeros(1024, device='cuda'); "
                        "torch.cuda.synchronize()"], check=False)
        n += 1
    print(f"churn: {n} short GPU processes", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "churn":
        churn(float(sys.argv[2]))
    elif sys.argv[1] == "worker":
        worker(*(int(v) for v in sys.argv[2:7]))
    else:
        launches = int(sys.argv[1])
        procs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        pool = int(sys.argv[3]) if len(sys.argv) > 3 else 1
        batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
        hw = int(sys.argv[5]) if len(sys.argv) > 5 else 8
        c = int(sys.argv[6]) if len(sys.argv) > 6 else 256
        ps = [subprocess.Popen([sys.executable, __file__, "worker", str(launches), str(pool),
                                str(batch), str(hw), str(c)]) for _ in range(procs)]
        nchurn = int(os.environ.get("STRESS_CHURN", "0"))
        cs = [subprocess.Popen([sys.executable, __file__, "churn", os.environ.get("STRESS_CHURN_S", "40")])
              for _ in range(nchurn)]
        rc = [p.wait(timeout=600) for p in ps]
        for p in cs:
            p.wait(timeout=600)
        print("exit codes", rc, flush=True)
