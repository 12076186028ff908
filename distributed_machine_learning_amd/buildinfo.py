"""What the native library is built from, and the hash that identifies it.

``build.py`` stamps the library with ``tree_hash()`` (compiled in as ``_C.build_id``);
``ops.native.lib()`` recomputes the hash from the sources it finds next to the library and
refuses a library built from different sources.  Pure standard library: usable before
torch is imported.
"""
from __future__ import annotations

import hashlib
import os

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# xnack-: the MI355X boxes run with XNACK off; code generated for that setting alone (instead of
# "either") is 0.5 % faster in the step (0.5542 -> 0.5516 ms, tools/probes/ab_flags.sh)
ARCH = "gfx950:xnack-"

# Flags every translation unit gets (machine-specific include paths are added by build.py and
# are not part of the hash).
# -fno-slp-vectorize: no packed fp32 VALU pairs (v_pk_fma_f32 and friends) formed from scalar
# code.  With them the one-launch BatchNorm backward kernel (bn_bwd_slab_kernel) took wrong
# compare masks in lanes 48-63 of a wave about once in a hundred launches while a second process
# shared the GPU (tools/probes/flake_ddp_bn.py: 7 expected in 660 two-rank runs, none without the
# packed ops; no documented wait-state rule is violated in the generated code).  The step is also
# 1.2 % faster without them (0.592 -> 0.585 ms on one box, tools/probes/ab_noslp.sh).
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
                "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__", "-fno-slp-vectorize"]

# Per-file device-code flags.  No -ffast-math anywhere: the kernels call the fast intrinsics
# they want by name (rsqrtf, __expf) and the reductions rely on a fixed summation order for
# bit-reproducible training, which reassociation would be free to break.
# -munsafe-fp-atomics (hardware float atomic add instead of a compare-and-swap loop) only
# where float atomics exist: the slab-less mode of the weight-gradient kernel, the stand-alone
# linear / loss kernels (neither runs in the training step), and the fp32 convolution's
# two-way tap split (two addends onto zero: order-independent).
FILE_FLAGS = {
    "conv_wgrad.hip": ["-munsafe-fp-atomics"],
    "optim_loss.hip": ["-munsafe-fp-atomics"],
    "conv_f32.hip": ["-munsafe-fp-atomics"],
}


def sources():
    out = []
    for d, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(d, f))
    return sorted(out)


def headers():
    out = []
    for d, _, files in os.walk(os.path.join(CSRC, "include")):
        for f in sorted(files):
            out.append(os.path.join(d, f))
    return sorted(out)


def flags_for(src: str):
    return COMMON_FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def digest(paths, extra: str = "") -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.relpath(p, CSRC).encode())
        h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(extra.encode())
    return h.hexdigest()


def tree_hash() -> str:
    """Hash of everything the library is built from: sources, headers, flags."""
    return digest(sources() + headers(), repr((COMMON_FLAGS, sorted(FILE_FLAGS.items()))))[:32]
