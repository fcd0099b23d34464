"""Build the gfx950 C-ABI library in-tree with hipcc (no torch extension machinery).

    python -m openstereo_amd.build          # incremental
    python -m openstereo_amd.build --force  # rebuild everything

Produces openstereo_amd/lib/libopenstereo_amd.so.  hipcc cross-compiles for gfx950 without a
GPU, so this runs in the CPU-only dev container; the .so is git-ignored but travels with gpurun.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBNAME = "libopenstereo_amd.so"
SOURCES = ["api.hip", "volume.hip", "conv3d.hip", "conv_inst_f32.hip", "conv_inst_f16x3.hip", "conv_inst_f16.hip", "conv_march.hip", "softargmin.hip", "layout.hip", "refine.hip", "backward.hip", "wgrad.hip", "geometry.hip", "dwconv.hip", "norm.hip", "gru_train.hip"]
ARCH = "gfx950"
HIPCC_FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built")
    return exe


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "osa_common.h"), os.path.join(CSRC, "conv_kernel.h"), os.path.join(CSRC, "conv_march.h"), os.path.join(CSRC, "conv_inst.h"), os.path.join(CSRC, "conv_inst_impl.h"), os.path.join(CSRC, "conv_cfgs.def"),
               os.path.join(HERE, "..", "include", "openstereo_amd.h")]

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, *HIPCC_FLAGS, "-c", s, "-o", o]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    out = lib_path()
    if force or _stale(out, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    from openstereo_amd import _ext          # the PyTorch-ROCm C++ extension over the C ABI (csrc/torch_ext.cpp)
    print(_ext.build(force="--force" in sys.argv))
