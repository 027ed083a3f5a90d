"""Build libeavqa_hip.so (gfx950) in-tree with hipcc.

``python -m eavqa_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU.  Objects are cached under ``csrc/_obj`` keyed by source mtime so a rebuild after a one-file
edit takes seconds.  The resulting ``.so`` lives next to the sources (git-ignored, but it travels
to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libeavqa_hip.so")

HIP_SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "seq.hip", "loss.hip", "optim.hip", "decode.hip", "retrieval.hip"]
CPP_SOURCES = ["api.cpp", "lm_block.cpp"]
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libeavqa_hip.so cannot be built")
    return exe


def _newer(dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    deps = [path, os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "eavqa.h")]
    if _newer(obj, deps):
        return obj
    cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", "-c", path, "-o", obj]
    if src.endswith(".hip"):
        cmd.insert(1, f"--offload-arch={ARCH}")
    else:
        cmd[1:1] = ["-x", "hip", f"--offload-arch={ARCH}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = HIP_SOURCES + CPP_SOURCES
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    if not _newer(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
