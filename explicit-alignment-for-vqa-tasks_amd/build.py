"""Build libeavqa_hip.so (gfx950) in-tree with hipcc.

``python -m eavqa_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU.  Objects are cached under ``csrc/_obj`` keyed by a content hash of the source, every in-repo
file it includes (from hipcc -MD) and the flags, so a rebuild after a one-file edit takes seconds
and an edit to an included file is never missed.  The resulting ``.so`` lives next to the sources (git-ignored, but it travels
to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import hashlib
import json
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
EXPORTS = os.path.join(CSRC, "exports.map")

HIP_SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "seq.hip", "loss.hip", "optim.hip", "decode.hip", "decode_direct.hip", "retrieval.hip"]
CPP_SOURCES = ["api.cpp", "lm_block.cpp", "t5_block.cpp"]
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


# host side hidden: the .so exports exactly what include/eavqa.h and include/eavqa_test.h declare (their declarations carry default
# visibility); device code keeps the toolchain default
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-fvisibility=hidden"]


def _digest(rel_paths, extra: str) -> str:
    """Content hash of the given files (paths relative to the repo root, in order) plus ``extra`` (the compile flags)."""
    h = hashlib.sha256(extra.encode())
    for p in rel_paths:
        h.update(p.encode())
        with open(os.path.join(ROOT, p), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _read_deps(dfile: str):
    """In-repo prerequisites of a ``-MD`` make fragment, relative to the repo root (system headers are skipped), so the
    cache key stays valid when the tree is copied elsewhere (the GPU box)."""
    with open(dfile) as f:
        text = f.read().replace("\\\n", " ")
    deps = [os.path.abspath(t) for t in text.split(":", 1)[1].split()]
    return sorted({os.path.relpath(d, ROOT) for d in deps if d.startswith(ROOT + os.sep)})


def _compile(src: str, verbose: bool) -> str:
    """One object, cached by CONTENT: the key hashes the flags, the source and every in-repo file the previous compile
    reported as included (hipcc -MD), so an edit to e.g. attention_mfma.hip (included by attention.hip) rebuilds
    attention.o even though attention.hip itself is untouched."""
    stem = os.path.splitext(src)[0]
    obj, dfile, kfile = (os.path.join(OBJ, stem + ext) for ext in (".o", ".d", ".key.json"))
    path = os.path.join(CSRC, src)
    flags = FLAGS + [f"--offload-arch={ARCH}"] + ([] if src.endswith(".hip") else ["-x", "hip"])
    cmd = [_hipcc()] + flags + [f"-I{INCLUDE}", f"-I{CSRC}", "-MD", "-MF", dfile, "-c", path, "-o", obj]
    if os.path.exists(obj) and os.path.exists(kfile):
        try:
            with open(kfile) as f:
                key = json.load(f)
            if key["digest"] == _digest(key["deps"], " ".join(flags)):
                return obj
        except (OSError, KeyError, ValueError):
            pass
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    deps = _read_deps(dfile)
    with open(kfile, "w") as f:
        json.dump({"deps": deps, "digest": _digest(deps, " ".join(flags))}, f)
    return obj


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = HIP_SOURCES + CPP_SOURCES
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    if not _newer(LIB, objs + [EXPORTS]):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
