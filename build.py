"""Builds the native runtime ``distributed_machine_learning_amd/_C*.so`` in-tree.

Every ``.hip`` / ``.cpp`` under ``csrc/`` is compiled by ``hipcc`` for gfx950 only and
linked with RCCL and the HIP runtime into one pybind11 extension (no torch headers, no
hipify pass).

Rebuilds are decided by CONTENT, never by file times: an object is rebuilt when the hash of
its source, every header under ``csrc/include`` and its compile command differs from the
hash stored next to it, and the library carries the hash of the whole source tree
(``_C.build_id``), which ``ops.native.lib()`` compares with the sources it finds at import.
A checkout with fresh mtimes, or an edited source with a stale ``.so``, is therefore caught.

    python build.py            # build what changed
    python build.py --force    # rebuild everything
    python build.py --check    # exit 1 if the library is missing or stale
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import importlib.util
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "distributed_machine_learning_amd")
OBJ = os.path.join(ROOT, "build", "obj")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")


def _buildinfo():
    # loaded by path: importing the package would pull in torch
    spec = importlib.util.spec_from_file_location("dml_buildinfo", os.path.join(PKG, "buildinfo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


INFO = _buildinfo()
CSRC, ARCH = INFO.CSRC, INFO.ARCH
sources, headers, tree_hash = INFO.sources, INFO.headers, INFO.tree_hash


def target_path() -> str:
    return os.path.join(PKG, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def compile_cmd(src: str, obj: str):
    import pybind11
    return [HIPCC, "-c", src, "-o", obj, *INFO.flags_for(src),
            "-I" + os.path.join(CSRC, "include"), "-I" + pybind11.get_include(),
            "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROCM, "include")]


def compile_one(src: str, obj: str, key: str) -> None:
    r = subprocess.run(compile_cmd(src, obj), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.relpath(src, ROOT)}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(obj + ".hash", "w") as f:
        f.write(key)


def _stored(path: str) -> str:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def library_build_id() -> str:
    """The source-tree hash the built library was stamped with ('' if there is none)."""
    return _stored(target_path() + ".build_id")


def is_current() -> bool:
    return os.path.exists(target_path()) and library_build_id() == tree_hash()


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = headers()
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        objs.append(obj)
        key = INFO.digest([src] + hdrs, " ".join(INFO.flags_for(src)))
        if force or not os.path.exists(obj) or _stored(obj + ".hash") != key:
            todo.append((src, obj, key))
    if verbose:
        print(f"[build] {len(todo)}/{len(objs)} objects to compile for {ARCH}")
    with cf.ThreadPoolExecutor(max_workers=max(1, min(jobs, 8))) as ex:
        for fut in [ex.submit(compile_one, s, o, k) for s, o, k in todo]:
            fut.result()
    out = target_path()
    tid = tree_hash()
    if todo or not os.path.exists(out) or library_build_id() != tid:
        # the id is compiled into the library (exported as _C.build_id) and written next to it
        stamp_src = os.path.join(OBJ, "build_id.cpp")
        stamp_obj = os.path.join(OBJ, "build_id.o")
        with open(stamp_src, "w") as f:
            f.write(f'extern "C" const char dml_build_id[] = "{tid}";\n')
        r = subprocess.run([HIPCC, "-c", stamp_src, "-o", stamp_obj, "-O1", "-fPIC"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for the build stamp:\n{r.stdout}\n{r.stderr}")
        link = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", out,
                *objs, stamp_obj, "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64",
                "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-ldl", "-lrt"]
        r = subprocess.run(link, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(out + ".build_id", "w") as f:
            f.write(tid)
        if verbose:
            print(f"[build] linked {os.path.relpath(out, ROOT)} (build id {tid})")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--check", action="store_true")
    ns = ap.parse_args()
    if ns.check:
        ok = is_current()
        print(f"[build] library {'is current' if ok else 'is MISSING or STALE'} "
              f"(sources {tree_hash()}, library {library_build_id() or 'none'})")
        sys.exit(0 if ok else 1)
    build(ns.force, ns.jobs)
