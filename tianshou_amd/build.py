"""Builds libtsengine.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m tianshou_amd.build [--force]

The shared object lands in tianshou_amd/lib/ so that it travels with the source tree
(a JIT cache under ~/.cache would not).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libtsengine.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# per-file extra flags (experiments: TS_EXTRA_FLAGS="ts_ppo.hip:-mllvm -amdgpu-sched-strategy=max-ilp")
EXTRA_FLAGS: dict[str, list[str]] = {}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libtsengine.so for gfx950)")


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "tsengine.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build_library(force: bool = False, verbose: bool = False, out: str | None = None) -> str:
    global LIB, OBJDIR
    extra = dict(EXTRA_FLAGS)
    env = os.environ.get("TS_EXTRA_FLAGS")
    if env:
        name, _, fl = env.partition(":")
        extra[name] = fl.split()
    lib_path = out or LIB
    objdir = OBJDIR if out is None else OBJDIR + "_" + os.path.basename(out).replace(".", "_")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = (force or not os.path.exists(obj)
                 or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m))
        if stale:
            jobs.append([hipcc, *FLAGS, *extra.get(os.path.basename(src), []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib_path):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib_path])
    return lib_path


if __name__ == "__main__":
    out = None
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            out = a[6:]
    print(build_library(force="--force" in sys.argv, verbose=True, out=out))
