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
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".hip", ".cpp"))) + [
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


HOST_SRC = os.path.join(HERE, "host", "genrich_amd.cpp")
HOST_BIN = os.path.join(HERE, "genrich-amd")


def build_host(force: bool = False) -> str:
    """The command-line host program (C++, g++): SAM/BAM ingest + options over the C ABI."""
    hd = os.path.dirname(HOST_SRC)
    deps = [HOST_SRC, LIB] + [os.path.join(hd, f) for f in ("bgzf_reader.h", "gx_inflate.h", "gx_crc32.h")]
    if force or not os.path.exists(HOST_BIN) or os.path.getmtime(HOST_BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", HOST_SRC, "-o", HOST_BIN, "-L" + HERE,
                               "-lgenrich_amd", "-lz", "-Wl,-rpath,$ORIGIN"])
    return HOST_BIN


INFLATE_TEST_LIB = os.path.join(HERE, "libgx_inflate_test.so")


def build_inflate_test() -> str:
    """gx_inflate.h / gx_crc32.h behind a C entry point each, for tests/test_inflate.py (host code only: g++)."""
    hd = os.path.dirname(HOST_SRC)
    srcs = [os.path.join(hd, f) for f in ("inflate_test.cpp", "gx_inflate.h", "gx_crc32.h")]
    if not os.path.exists(INFLATE_TEST_LIB) or os.path.getmtime(INFLATE_TEST_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared", srcs[0], "-o", INFLATE_TEST_LIB, "-lz"])
    return INFLATE_TEST_LIB


def build(force: bool = False) -> str:
    if force or needs_build():
        subprocess.check_call([HIPCC] + FLAGS + [SRC, EMIT, "-o", LIB])
    build_host(force)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
