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
# NO packed-fp32 VALU instructions (v_pk_mul / add / fma_f32) in ANY kernel of the library: the backend feature is switched off for every
# translation unit, and tests/test_isa_lint_cpu.py disassembles the shipped objects and fails on the first one it finds.
# Why: on gfx950 a VOP3P fp32 instruction whose LO result reads the HI half of its src1 pair (op_sel:[0,1] -- hipcc's SLP
# vectoriser emits it for horizontal sums) returns wrong values in isolated 16-lane passes while an MFMA-dense kernel of another stream is
# resident on the same SIMD: r5 found the x4 fused head wrong next to the d-marching convolution, r6 reduced it to one instruction in a
# ten-line kernel on three GPUs (profiles/round6/pk_micro_matrix2.txt, head_variants_single.txt).  Every kernel of a forward can be
# co-resident with another sub-batch stream's marching kernel, so the property has to hold for all of them; it costs nothing (packed fp32
# beside MFMAs is an anti-lever on gfx950: +1 % on the headline without it, profiles/round6/build_flags_ab.txt) and takes the marching
# kernels' spills down.  Same operations in the same order (-ffp-contract=off): bit-identical to the packed build on an idle GPU.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
HIPCC_FLAGS += NO_PACKED_F32
EXTRA_FLAGS: dict = {}


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
