"""Single-node launcher: one process per GPU.

The reference is started by hand, one ``python main.py --master-ip .. --rank r
--num-nodes N`` per machine (/root/reference/part2/2a/main.py:210-218).  On one
8-GPU node that becomes N local processes; this helper starts them with the same
three flags, pins each to its GPU through ``LOCAL_RANK`` and propagates the first
non-zero exit code, terminating the remaining ranks so a crashed rank cannot leave
the others blocked in a collective.

    python -m distributed_machine_learning_amd.launch --nproc 8 main.py --sync ddp
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from typing import List


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(script: str, script_args: List[str], nproc: int, master_addr: str = "127.0.0.1",
           master_port: int = 0, timeout_s: float = 0.0) -> int:
    port = master_port or free_port()
    procs = []
    # one id per launch: the loopback transport names its shared-memory segment after it, so a
    # segment a crashed earlier run left behind on the same port can never be mistaken for ours
    launch_id = f"{os.getpid():x}{int(time.time() * 1e3) & 0xffffffff:x}"
    for rank in range(nproc):
        env = dict(os.environ)
        env["LOCAL_RANK"] = str(rank)
        env["DML_LAUNCH_ID"] = launch_id
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-u", script, *script_args,
               "--master-ip", f"{master_addr}:{port}", "--rank", str(rank),
               "--num-nodes", str(nproc)]
        procs.append(subprocess.Popen(cmd, env=env))
    deadline = time.time() + timeout_s if timeout_s > 0 else None
    code = 0
    try:
        alive = set(range(nproc))
        while alive:
            for r in list(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and code == 0:
                    code = rc
            if code != 0 or (deadline and time.time() > deadline):
                if code == 0:
                    code = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return code


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--nproc", type=int, required=True)
    ap.add_argument("--master-addr", default="127.0.0.1")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--timeout", type=float, default=0.0)
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    return launch(ns.script, ns.script_args, ns.nproc, ns.master_addr, ns.master_port,
                  ns.timeout)


if __name__ == "__main__":
    sys.exit(main())
