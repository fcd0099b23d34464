"""ISA lint of the shipped gfx950 code: which kernels of a library contain packed-fp32 VALU instructions (v_pk_mul / add / fma_f32)?

    python tools/isa_lint.py [path/to/lib.so]        # default: the shipped openstereo_amd/lib/libopenstereo_amd.so

The library is built with the backend's packed-fp32 feature switched off (openstereo_amd/build.py NO_PACKED_F32; DESIGN.md 3.9: a kernel
whose loop carries these instructions returned wrong 16-lane passes next to the d-marching convolution of another stream), and
tests/test_isa_lint_cpu.py fails when any kernel of any shipped code object carries one.  Works without a GPU: the .hip_fatbin section of
the .so is split into its offload bundles, every gfx950 code object is disassembled with llvm-objdump."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PACKED_F32 = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
# r6 narrowed the failing form to: a VOP3P instruction whose LO result reads the HI half of src1 (op_sel:[.,1,..]); the library carries no
# packed-fp32 arithmetic at all, and any OTHER VOP3P instruction (v_pk_mov_b32, packed f16) must not use that operand form either
OPSEL_SRC1 = re.compile(r"\bv_pk_\w+\b[^\n]*\bop_sel:\[[01],1")


def code_objects(so_path, workdir):
    """gfx950 code objects (paths) of every offload bundle in the library's .hip_fatbin section"""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        b = os.path.join(workdir, f"bundle{i}.bin")
        open(b, "wb").write(blob[s:e])
        co = os.path.join(workdir, f"bundle{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={b}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
            out.append(co)
    return out


def scan(so_path):
    """{kernel symbol: (packed-fp32 instruction count, total instruction count)} over every gfx950 kernel of the library"""
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(so_path, wd):
            dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
            name = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    name = m.group(1)
                    res.setdefault(name, [0, 0])
                    continue
                if name is None or not line.startswith("\t") and not line.startswith(" "):
                    continue
                res[name][1] += 1
                if PACKED_F32.search(line) or OPSEL_SRC1.search(line):
                    res[name][0] += 1
    return {k: tuple(v) for k, v in res.items()}


if __name__ == "__main__":
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "openstereo_amd", "lib", "libopenstereo_amd.so")
    r = scan(path)
    bad = {k: v for k, v in r.items() if v[0]}
    names = subprocess.run(["c++filt"], input="\n".join(bad), capture_output=True, text=True).stdout.split("\n")
    for n, (k, v) in zip(names, bad.items()):
        print(f"{v[0]:6d} packed-fp32 of {v[1]:7d} instructions  {n[:150]}")
    print(f"{len(r)} kernels / device functions in {path}: {len(bad)} carry packed-fp32 instructions ({sum(v[0] for v in bad.values())} in total)")
    sys.exit(1 if bad else 0)
