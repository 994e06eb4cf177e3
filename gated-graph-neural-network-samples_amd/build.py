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
    if any(os.path.getmtime(s) > t for s in _deps()):
        return True
    # ... or an object older than a source it #includes (a library linked from such an object is newer than every source)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
        if os.path.exists(obj) and any(os.path.getmtime(i) > os.path.getmtime(obj) for i in _included_sources(src)):
            return True
    return False


def _headers():
    root = os.path.dirname(HERE)
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(root, "include", "*.h"))


def _included_sources(src: str) -> list:
    """The csrc/*.hip files `src` pulls in with #include "x.hip" (the *_split.hip translation units are their base file compiled
    a second time): an object is stale when one of THESE is newer too -- headers are covered by newest_header."""
    import re
    out = []
    with open(src, "r", encoding="utf-8", errors="replace") as f:
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+\.hip)"', f.read(), re.M):
            inc = os.path.join(CSRC, m.group(1))
            if os.path.exists(inc):
                out += [inc] + _included_sources(inc)
    return out


# Kernels on the bf16 matrix pipe are compiled WITHOUT packed-f32 vector instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32):
# next to bf16 MFMAs of the partner wave a packed-f32 instruction stalls the SIMD for the length of the MFMA stream, a plain
# v_fma_f32 does not (tools/mfma_overlap.hip -DBF16: 6400 FMAs beside 1600 MFMAs take 59.6k clocks unpacked, 71.7k = the sum packed).
# ... and with the AMDGPU register-pressure trackers in the machine scheduler: at the 256-register limit these kernels sit at, the
# default scheduler's pressure estimate costs 8-70 B of scratch per lane more (split fused GRU R = 1: 8 -> 0 B; panel GRU 228 -> 156 B).
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"]
PER_SOURCE_FLAGS = {"ggnn_gru_fused_split.hip": NO_PACKED_F32, "ggnn_gru_wide.hip": NO_PACKED_F32, "ggnn_dense_graph_split.hip": NO_PACKED_F32, "ggnn_gru_bwd_fused_split.hip": NO_PACKED_F32, "ggnn_msg_compact.hip": NO_PACKED_F32, "ggnn_panel.hip": NO_PACKED_F32, "ggnn_bwd_gemm.hip": NO_PACKED_F32}


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every csrc/*.hip to an object (in parallel, only those older than their source or any header) and link them."""
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libggnn_hip.so")
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I", os.path.join(os.path.dirname(HERE), "include")]
    flags += os.environ.get("GGNN_EXTRA_HIPCC_FLAGS", "").split()      # e.g. -DGGNN_XTY_TIMELINE=1 for tools/xty_timeline.py (then --force)
    newest_header = max([os.path.getmtime(h) for h in _headers()] + [0.0])
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
            [os.path.getmtime(src), newest_header] + [os.path.getmtime(i) for i in _included_sources(src)])
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if stale:
            cmd = [hipcc] + flags + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print("[ggnn build] " + " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, jobs))
    tmp = OUT + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print("[ggnn build] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
