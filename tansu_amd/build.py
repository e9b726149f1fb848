"""Build the HIP library in-tree: tansu_amd/lib/libnisshi_hip.so (gfx950 only)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libnisshi_hip.so")
SOURCES = ["nisshi_hip.hip"]
DEPS = ["nisshi_hip.hip", "encode.inc", "storage.inc", "fetch.inc", "compact.inc", "inflate.inc", "zstd.inc", "deflate.inc", "api.inc", "pool.inc", "wire.h", "gf2_crc32c.h"]


def source_hash() -> str:
    """A hash of the kernel sources and the C header: what profiles are taken of.
    bench.py reports HBM traffic from a committed profile only while this matches."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, d) for d in sorted(DEPS)] + [os.path.join(ROOT, "include", "nisshi_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(ROOT, "include", "nisshi_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    # extra compiler flags for diagnostic builds, e.g. NISSHI_CXXFLAGS=-DNISSHI_PHASES
    cmd[1:1] = os.environ.get("NISSHI_CXXFLAGS", "").split()
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
