"""Builds the native runtime ``distributed_machine_learning_amd/_C*.so`` in-tree.

Every ``.hip`` / ``.cpp`` under ``csrc/`` is compiled by ``hipcc`` for gfx950 only and
linked with RCCL and the HIP runtime into one pybind11 extension (no torch headers, no
hipify pass).  Incremental: an object is rebuilt when its source or any header is newer.

    python build.py            # build
    python build.py --force    # rebuild everything
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "distributed_machine_learning_amd")
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
ARCH = "gfx950"
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")


def target_path() -> str:
    return os.path.join(PKG, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def sources():
    out = []
    for d, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(d, f))
    return sorted(out)


def headers_mtime() -> float:
    m = 0.0
    for d, _, files in os.walk(os.path.join(CSRC, "include")):
        for f in files:
            m = max(m, os.path.getmtime(os.path.join(d, f)))
    return max(m, os.path.getmtime(__file__))


def compile_one(src: str, obj: str) -> None:
    import pybind11
    cmd = [HIPCC, "-c", src, "-o", obj, "-O3", "-std=c++17", "-fPIC",
           f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           "-I" + os.path.join(CSRC, "include"), "-I" + pybind11.get_include(),
           "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROCM, "include"),
           "-D__HIP_PLATFORM_AMD__"]
    if src.endswith(".hip"):
        cmd += ["-ffast-math", "-munsafe-fp-atomics"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.relpath(src, ROOT)}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = headers_mtime()
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            todo.append((src, obj))
    if verbose:
        print(f"[build] {len(todo)}/{len(objs)} objects to compile for {ARCH}")
    with cf.ThreadPoolExecutor(max_workers=max(1, min(jobs, 8))) as ex:
        for fut in [ex.submit(compile_one, s, o) for s, o in todo]:
            fut.result()
    out = target_path()
    if todo or not os.path.exists(out):
        link = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", out,
                *objs, "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64",
                "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"]
        r = subprocess.run(link, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {os.path.relpath(out, ROOT)}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    ns = ap.parse_args()
    build(ns.force, ns.jobs)
