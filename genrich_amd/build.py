"""Builds the HIP library in-tree for gfx950: genrich_amd/libgenrich_amd.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off / no fast-math: the reference is
built without FMA contraction and the float/double results must match it bit for bit.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gx_api.hip")
EMIT = os.path.join(HERE, "csrc", "gx_emit.cpp")
DEPS = [SRC, EMIT] + [os.path.join(HERE, "csrc", f) for f in ("gx_kernels.h", "gx_stats.h", "gx_merge.h", "gx_math.h")] + [
    os.path.join(os.path.dirname(HERE), "include", "genrich_amd.h")]
LIB = os.path.join(HERE, "libgenrich_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False) -> str:
    if force or needs_build():
        subprocess.check_call([HIPCC] + FLAGS + [SRC, EMIT, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
