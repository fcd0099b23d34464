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
# Per-file flags.  softargmin.hip is built WITHOUT the SLP vectoriser, i.e. without packed-fp32 VALU instructions (v_pk_mul / add / fma_f32):
# r5 found the fused head returning wrong disparities in isolated quarter waves (16 pixels of one row, errors up to tens of pixels)
# whenever the d-marching conv kernel (250-256 VGPRs, 16-pass f16 MFMAs) runs on another stream of the same GPU -- the timed
# configuration's sub-batch streams.  Its loads are right (checksum of everything it loads: clean), its arithmetic is not (checksum of its
# exponentials: dirty; polynomial exp2 instead of v_exp_f32: still dirty), and the same source compiled without packed math is clean:
# 0 differing elements in 40 launches under that load against 40 of 40 launches with ~600 wrong pixels each
# (profiles/round5/head_packed_math_under_march_load.txt, tools/diag_head_under_load.py).  Same arithmetic, same
# order (-ffp-contract=off): results are bit-identical to the packed build on an otherwise idle GPU.
EXTRA_FLAGS = {"softargmin.hip": ["-fno-slp-vectorize"]}


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
        if force or _stale(o, [s, os.path.abspath(__file__)] + headers):
            cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o]
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
