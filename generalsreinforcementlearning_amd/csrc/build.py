#!/usr/bin/env python3
"""Builds libgvec_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python generalsreinforcementlearning_amd/csrc/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so lands in
generalsreinforcementlearning_amd/ (git-ignored; it travels to the GPU box with the tree).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libgvec_hip.so")
SRCS = ["gvec_kernels.hip", "gvec_api.hip"]
DEPS = ["gvec_device.hpp", "gvec_packed.hpp", "gvec_launch.hpp", os.path.join(ROOT, "include", "generals_vec.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-save-temps=obj"]  # keeps build/*-gfx950.s: tests/test_kernel_asm.py scans the generated ISA


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    deps = [os.path.join(HERE, d) if not os.path.isabs(d) else d for d in DEPS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SRCS:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + deps):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    if jobs:
        if verbose:
            print(f"[gvec build] compiling {len(jobs)} HIP translation unit(s) for gfx950 ...", flush=True)
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed: " + " ".join(r.args))
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
        # -save-temps leaves ~60 MB of preprocessed sources and bitcode beside the objects: only the gfx950 ISA is kept
        for f in os.listdir(os.path.join(HERE, "build")):
            if f.endswith((".hipi", ".bc", ".out", ".resolution.txt", ".hipfb")) or f.endswith("x86_64-unknown-linux-gnu.s"):
                os.unlink(os.path.join(HERE, "build", f))
    if jobs or force or _newer(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[gvec build] wrote {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
