"""CPU sanitizer builds of the host runtime (no GPU needed, none used).

GPU AddressSanitizer is not available on this pool; what CAN be checked is the host side,
which is where the runtime's threads and state machines live:

  asan   every csrc source compiled by hipcc with AddressSanitizer + UBSan on the HOST pass
         only (-Xarch_host -fsanitize=...), linked with tests/native/host_harness.cpp into an
         executable that runs the TCP store (server thread, concurrent clients, hostile
         requests, time-outs), the bucket planner and the engine's layout planner;
  tsan   the TCP store alone (it is the threaded component) under ThreadSanitizer.

    python tools/sanitize_host.py [asan] [tsan]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import build as B  # noqa: E402

HARNESS = os.path.join(ROOT, "tests", "native", "host_harness.cpp")
OUT = os.path.join(ROOT, "build", "sanitize")
CLANGXX = os.path.join(B.ROCM, "lib", "llvm", "bin", "clang++")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r


def asan() -> None:
    import pybind11
    os.makedirs(OUT, exist_ok=True)
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
           "-Xarch_host", "-fno-sanitize-recover=undefined", "-Xarch_host", "-fno-omit-frame-pointer",
           "-g1"]
    inc = ["-I" + os.path.join(B.CSRC, "include"), "-I" + pybind11.get_include(),
           "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(B.ROCM, "include")]
    srcs = [s for s in B.sources() if not s.endswith("bindings.cpp")] + [HARNESS]
    objs = []

    def one(src):
        obj = os.path.join(OUT, os.path.basename(src) + ".asan.o")
        flags = B.INFO.flags_for(src) if src != HARNESS else B.INFO.COMMON_FLAGS
        flags = [f for f in flags if f != "-O3"] + ["-O1"]
        _run([B.HIPCC, "-x", "hip", "-c", src, "-o", obj, *flags, *san, *inc])
        return obj

    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, srcs))
    exe = os.path.join(OUT, "host_harness_asan")
    pylib = "python" + sysconfig.get_config_var("LDVERSION")
    _run([B.HIPCC, "-o", exe, *objs, "-fsanitize=address", "-fsanitize=undefined",
          f"--offload-arch={B.ARCH}", "-fno-gpu-rdc", "-L" + os.path.join(B.ROCM, "lib"), "-lrccl",
          "-lamdhip64", "-Wl,-rpath," + os.path.join(B.ROCM, "lib"),
          "-L" + (sysconfig.get_config_var("LIBDIR") or "/usr/lib"), "-l" + pylib,
          "-lpthread", "-ldl", "-lrt"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = _run([exe, "all"], env=env, timeout=600)
    sys.stdout.write(r.stdout)
    print("[sanitize] asan+ubsan: clean")


def tsan() -> None:
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "host_harness_tsan")
    _run([CLANGXX, "-std=c++17", "-O1", "-g1", "-fsanitize=thread", "-DHARNESS_STORE_ONLY",
          "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(B.CSRC, "include"),
          "-I" + os.path.join(B.ROCM, "include"), HARNESS,
          os.path.join(B.CSRC, "comm", "tcp_store.cpp"), "-o", exe, "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = _run([exe, "store"], env=env, timeout=600)
    sys.stdout.write(r.stdout)
    print("[sanitize] tsan (tcp store): clean")


if __name__ == "__main__":
    which = sys.argv[1:] or ["asan", "tsan"]
    for w in which:
        {"asan": asan, "tsan": tsan}[w]()
