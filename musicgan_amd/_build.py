"""Build libmusicgan_hip.so (gfx950) in-tree with hipcc.  `python -m musicgan_amd._build [--force]`.

The shared library is git-ignored but travels to the GPU box with the repo snapshot; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmusicgan_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    extra = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    extra.append(os.path.join(os.path.dirname(HERE), "include", "musicgan_hip.h"))
    return max(os.path.getmtime(p) for p in extra)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 6) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
