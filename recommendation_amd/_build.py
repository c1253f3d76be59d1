"""Builds recommendation_amd/libgcr.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is kept
in-tree (git-ignored) and travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgcr.so")
STAMP = LIB + ".stamp"
ARCH = "gfx950"

FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "gcr.h")]:
        h.update(path.encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def up_to_date():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build(force=False, verbose=False, keep_temps=None):
    """Compile every HIP/C++ source into libgcr.so; returns the library path."""
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libgcr.so cannot be built on this machine")
    tmp = keep_temps or tempfile.mkdtemp(prefix="gcrbuild_")
    os.makedirs(tmp, exist_ok=True)
    objs = []
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    procs = []
    for src in sources():
        obj = os.path.join(tmp, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, *inc, "-c", src, "-o", obj]
        if keep_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB + ".tmp"]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}")
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_digest())
    if not keep_temps:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True,
                keep_temps="/tmp/gcrbuild" if "--keep-temps" in sys.argv else None))
