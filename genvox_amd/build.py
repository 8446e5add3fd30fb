"""Build recipe for the gfx950 shared library (hipcc, in-tree, no JIT cache).

    python -m genvox_amd.build          # or __graft_entry__.build()

Produces genvox_amd/libgenvox_amd.so next to this file; the .so is git-ignored but travels
with the tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgenvox_amd.so")
SOURCES = ["gemm_f32.hip", "skinny.hip", "dec_resident.hip", "attention.hip", "attn_persist.hip", "misc.hip", "griffinlim.hip", "train.hip", "gvx_api.hip"]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=off"]  # keep a*b+c as written: fused/unfused choices are explicit (fmaf) in the kernels


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True, stamps: bool = False) -> str:
    """stamps=True builds the diagnostic library libgenvox_amd_stamps.so (phase timestamps, tools/stamps.py)."""
    objs, jobs = [], []
    flags = CXXFLAGS + (["-DGVX_STAMPS"] if stamps else []) + os.environ.get("GVX_EXTRA_FLAGS", "").split()
    lib = LIB.replace(".so", "_stamps.so") if stamps else LIB
    if os.environ.get("GVX_LIB_NAME"):
        lib = os.path.join(HERE, os.environ["GVX_LIB_NAME"])
    sfx = ".stamps.o" if stamps else ".o"
    if os.environ.get("GVX_LIB_NAME"):
        sfx = "." + os.environ["GVX_LIB_NAME"] + ".o"
    headers = [os.path.join(CSRC, "gvx_kernels.h"), os.path.join(CSRC, "attn_step_body.h"), os.path.join(os.path.dirname(HERE), "include", "genvox_amd.h")]
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", sfx))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            jobs.append([HIPCC, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib", "-o", lib])
    if not stamps and not os.environ.get("GVX_LIB_NAME"):
        # a diagnostic variant older than the C ABI's header no longer exports what _lib.py binds (tools/stamps*.py would fail with an
        # AttributeError at load): better gone than stale - `python -m genvox_amd.build --stamps` makes it again
        variant = LIB.replace(".so", "_stamps.so")
        if os.path.exists(variant) and os.path.getmtime(variant) < os.path.getmtime(headers[-1]):
            os.remove(variant)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
