"""Build librsaf.so (hipcc, gfx950) in-tree.

``python -m robust_speech_analysis_framework_amd.build`` or ``build_library()``.
hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD_DIR = os.path.join(PKG_DIR, "_build")
LIB_PATH = os.path.join(PKG_DIR, "librsaf.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "rsaf.h"))
    return sorted(hs)


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
                "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
                "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# RSAF_BUILD_TEST_KERNELS=1: also compile the kernels that newer ones superseded in the product (the workgroup-FFT pitch
# correlation kernels for transforms of up to 2 048 points), which some tests run as an independent A/B check
if os.environ.get("RSAF_BUILD_TEST_KERNELS", "0") == "1":
    COMMON_FLAGS.append("-DRSAF_TEST_KERNELS")


def _compile_one(src: str, hdr_digest: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(BUILD_DIR, os.path.basename(src) + ".o")
    stamp = obj + ".sha"
    key = _digest([src]) + hdr_digest + " ".join(COMMON_FLAGS)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj
    cmd = [_hipcc(), *COMMON_FLAGS, "-c", src, "-o", obj]
    if verbose:
        print("[rsaf build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(key)
    return obj


def build_library(force: bool = False, verbose: bool = True, jobs: int | None = None) -> str:
    os.makedirs(BUILD_DIR, exist_ok=True)
    srcs = _sources()
    hdr_digest = _digest(_headers())
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, hdr_digest, force, verbose), srcs))
    link_key = _digest(objs)
    stamp = LIB_PATH + ".sha"
    if (not force and os.path.exists(LIB_PATH) and os.path.exists(stamp)
            and open(stamp).read() == link_key):
        return LIB_PATH
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", LIB_PATH]
    if verbose:
        print("[rsaf build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(link_key)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
