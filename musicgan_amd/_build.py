"""Build libmusicgan_hip.so (gfx950) in-tree with hipcc.  `python -m musicgan_amd._build [--force]`.

The shared library is git-ignored but travels to the GPU box with the repo snapshot; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import concurrent.futures as cf
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmusicgan_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"]
# The resource remarks of every kernel are kept next to its object file (<name>.usage.json): a kernel whose accumulators end up
# in scratch memory still reports "0 spills" and passes every parity test -- at a quarter of the speed.  tests/test_build.py reads them.


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
    usage, cur, rest = {}, None, []
    after_remark = False
    for line in r.stderr.splitlines():
        if "-Rpass-analysis=kernel-resource-usage" not in line:
            if after_remark and re.match(r"^\s*\d*\s*\|", line):  # the source excerpt clang prints under a remark
                continue
            after_remark = False
            rest.append(line)
            continue
        after_remark = True
        text = line.split("remark:", 1)[1].rsplit("[-Rpass-analysis", 1)[0]
        text = re.sub(r"^\s*\S+?:\d+:\d+:", "", text).strip() if re.match(r"^\s*\S+?:\d+:\d+:", text) else text.strip()
        if text.startswith("Function Name:"):
            cur = usage.setdefault(text.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in text:
            k, v = text.rsplit(":", 1)
            try:
                cur[k.strip()] = int(v)
            except ValueError:
                pass
    with open(obj[:-2] + ".usage.json", "w") as f:
        json.dump(usage, f, indent=1, sort_keys=True)
    if any(ln.strip() for ln in rest):
        sys.stderr.write("\n".join(rest) + "\n")
    return obj


def resource_usage() -> dict:
    """{mangled kernel name: {"VGPRs": .., "ScratchSize [bytes/lane]": .., ...}} of the objects of the last build."""
    out = {}
    for f in sorted(os.listdir(LIB_DIR)):
        if f.endswith(".usage.json"):
            with open(os.path.join(LIB_DIR, f)) as fh:
                out.update(json.load(fh))
    return out


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
