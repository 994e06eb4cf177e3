"""In-tree build of libggnn_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m ggnn_amd.build            # or __graft_entry__.build()

The shared object is written next to the package sources (git-ignored, but it travels to the GPU box
with the gpurun snapshot).  Rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libggnn_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    root = os.path.dirname(HERE)
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(root, "include", "*.h"))


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libggnn_hip.so")
    tmp = OUT + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-I", os.path.join(os.path.dirname(HERE), "include")] + sources() + ["-o", tmp]
    if verbose:
        print("[ggnn build] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
