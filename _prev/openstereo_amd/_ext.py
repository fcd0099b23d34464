"""The PyTorch-ROCm C++ extension over the C ABI (csrc/torch_ext.cpp): `torch.ops.osa_native.*` -- TORCH_LIBRARY ops with at::Tensor
arguments, the current HIP stream and TORCH_CHECK errors (north_star: "exposed to Python through a PyTorch-ROCm C++/HIP extension";
SURVEY 8b).  Built in-tree next to the C-ABI library (`python -m openstereo_amd.build`: one g++ invocation against torch's headers, ~12 s)
as openstereo_amd/lib/libosa_torch_ext.so, which links libopenstereo_amd.so through $ORIGIN.

`ops` is the loaded namespace or None.  The Python layer (ops.py, engine.PackedConv3d) routes its hot launches through it when present;
without it the same entry points are reached through ctypes (`_lib.py`) -- the extension adds no kernels and no fallbacks, it replaces
argument marshalling.  OSA_TORCH_EXT=0 keeps it unloaded (A/B of the two dispatch paths)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "lib", "libosa_torch_ext.so")
SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
ops = None


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile csrc/torch_ext.cpp (host code only) against this interpreter's torch and link it to the in-tree C-ABI library."""
    import torch
    from .build import lib_path, _stale
    hdr = os.path.join(_HERE, "..", "include", "openstereo_amd.h")
    if not force and not _stale(EXT_PATH, [SRC, hdr, lib_path()]):
        return EXT_PATH
    ti = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: the torch extension cannot be built")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{ti}/include", f"-I{ti}/include/torch/csrc/api/include", f"-I{rocm}/include", SRC, "-o", EXT_PATH,
           f"-L{ti}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           f"-L{os.path.dirname(lib_path())}", "-lopenstereo_amd", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ti}/lib"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return EXT_PATH


def load():
    """torch.ops.load_library (once).  Returns the `torch.ops.osa_native` namespace, or None when the extension is not built / switched off."""
    global ops
    if ops is not None:
        return ops
    if os.environ.get("OSA_TORCH_EXT", "1") == "0" or os.environ.get("OSA_LIB_PATH") or not os.path.exists(EXT_PATH):
        return None          # (OSA_LIB_PATH: an A/B build of the C-ABI library is in use -- the extension links the shipped one)
    import torch
    from . import _lib
    _lib.load()              # the C-ABI library first: a missing / stale one is reported by its own loader
    torch.ops.load_library(EXT_PATH)
    ns = torch.ops.osa_native
    if int(ns.abi_version()) != _lib.abi_version():
        raise _lib.EngineError(f"libosa_torch_ext.so was built against ABI {int(ns.abi_version())}: rebuild with `python -m openstereo_amd.build`")
    ops = ns
    return ops
