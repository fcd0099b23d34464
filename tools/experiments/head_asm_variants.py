"""r6: delta-debugging the VICTIM side of the co-residency finding on the real x4 head kernel.

The fused head built WITH packed-fp32 math returns wrong 16-lane passes next to the d-marching convolution; none of the single-instruction
probes of tools/experiments/pk_probe.hip does.  So the real kernel's ISA is edited instead: csrc/softargmin.hip is compiled to assembly
with the packed feature ON, and for every variant a chosen CLASS of its v_pk_{mul,add,fma}_f32 instructions inside upsample4_softargmin_kernel
is expanded into the two scalar VOP3 instructions it stands for (same operands, same modifiers, same rounding; lo result through a spare
VGPR so that overlapping source / destination pairs stay correct).  Every variant is assembled into its own code object
(tools/experiments/head_variants/<name>.co), which tools/diag_head_variants.py loads with hipModuleLoad and runs next to the loads.

    python tools/experiments/head_asm_variants.py          # writes head_variants/*.co (no GPU needed)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
OUT = os.path.join(ROOT, "tools", "experiments", "head_variants")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN3osa27upsample4_softargmin_kernelENS_6UpArgsE"
T_LO, T_HI = "v38", "v39"          # spare registers (the kernel uses v0..v37); next_free_vgpr is raised to 40

PK = re.compile(r"^\s*v_pk_(mul|add|fma)_f32\s+(.*)$")


def halves(op):
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", op)
    if m:
        return f"{m.group(1)}{m.group(2)}", f"{m.group(1)}{m.group(3)}"
    return op, op                   # inline constant: the same value for both halves


def mods(text, name, n, default):
    m = re.search(name + r":\[([01,]+)\]", text)
    if not m:
        return [default] * n
    v = [int(x) for x in m.group(1).split(",")]
    return v + [default] * (n - len(v))


def expand(line):
    """the two scalar instructions (+ the move of the lo result) a packed-fp32 instruction stands for"""
    m = PK.match(line)
    kind, rest = m.group(1), m.group(2)
    rest = rest.split(";")[0].strip()
    parts = rest.split(" ")
    ops_txt = " ".join(p for p in parts if ":" not in p or p.startswith("v[") or p.startswith("s["))
    mod_txt = " ".join(p for p in parts if ":[" in p)
    ops = [o.strip() for o in ops_txt.split(",") if o.strip()]
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    assert n == (3 if kind == "fma" else 2), line
    sel_lo, sel_hi = mods(mod_txt, "op_sel", n, 0), mods(mod_txt, "op_sel_hi", n, 1)
    neg_lo, neg_hi = mods(mod_txt, "neg_lo", n, 0), mods(mod_txt, "neg_hi", n, 0)
    d_lo, d_hi = halves(dst)
    mn = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[kind]

    def src(i, sel, neg):
        lo, hi = halves(srcs[i])
        s = hi if sel[i] else lo
        if neg[i]:
            s = s[1:] if s.startswith("-") else "-" + s
        return s
    lo_ops = ", ".join(src(i, sel_lo, neg_lo) for i in range(n))
    hi_ops = ", ".join(src(i, sel_hi, neg_hi) for i in range(n))
    return [f"\t{mn} {T_LO}, {lo_ops}", f"\t{mn} {d_hi}, {hi_ops}", f"\tv_mov_b32_e32 {d_lo}, {T_LO}"]


def xhalf(text):
    """does this packed instruction read a VGPR pair ACROSS halves (lo result from a hi half or hi result from a lo half)?"""
    m = PK.match(text)
    rest = m.group(2).split(";")[0].strip()
    parts = rest.split(" ")
    ops = [o.strip() for o in " ".join(q for q in parts if ":[" not in q).split(",") if o.strip()][1:]
    mod_txt = " ".join(q for q in parts if ":[" in q)
    sel_lo, sel_hi = mods(mod_txt, "op_sel", len(ops), 0), mods(mod_txt, "op_sel_hi", len(ops), 1)
    return any(o.startswith("v[") and (sel_lo[i] == 1 or sel_hi[i] == 0) for i, o in enumerate(ops))


def rewrite(asm_lines, pick, pre=None, nop_scale=None):
    """expand the packed instructions of the x4 head for which pick(kind, text, in_loop) is true; pre(kind, text) -> lines inserted before a
    packed instruction that stays; nop_scale: every `s_nop N` of the kernel becomes `s_nop nop_scale`"""
    out, inside, in_loop, n_exp, n_keep = [], False, False, 0, 0
    for ln in asm_lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        elif inside and ln.startswith(".Lfunc_end"):
            inside = False
        if inside:
            if ln.startswith(".LBB3_12:"):
                in_loop = True
            elif ln.startswith(".LBB3_16:"):
                in_loop = False
            m = PK.match(ln)
            if m:
                ordinal = n_exp + n_keep
                if (pick(m.group(1), ln, in_loop, ordinal) if pick.__code__.co_argcount == 4 else pick(m.group(1), ln, in_loop)):
                    out.extend(expand(ln.rstrip("\n")))
                    n_exp += 1
                    continue
                n_keep += 1
                if pre is not None:
                    out.extend(pre(m.group(1), ln))
            elif nop_scale is not None and re.match(r"^\s*s_nop \d+", ln):
                ln = f"\ts_nop {nop_scale}"
        out.append(ln.rstrip("\n"))
    txt = "\n".join(out) + "\n"
    # the descriptor and the metadata of this kernel: two more VGPRs
    txt = re.sub(r"(\.amdhsa_kernel " + KERNEL + r".*?\.amdhsa_next_free_vgpr )38", r"\g<1>40", txt, flags=re.S)
    txt = re.sub(r"(\.name:\s+" + KERNEL + r"\n(?:.*\n)*?\s+\.vgpr_count:\s+)38", r"\g<1>40", txt)
    return txt, n_exp, n_keep


VARIANTS = {
    "packed":     lambda k, t, l: False,                        # the r4 object as compiled: must fail next to the load
    "scalar":     lambda k, t, l: True,                         # every packed instruction expanded: must be clean
    "x_fma":      lambda k, t, l: k == "fma",                   # x_<class>: ONLY this class expanded, the rest stays packed
    "x_mul":      lambda k, t, l: k == "mul",
    "x_add":      lambda k, t, l: k == "add",
    "x_sgpr":     lambda k, t, l: "s[" in t,
    "x_vgpr":     lambda k, t, l: "s[" not in t,
    "x_loop":     lambda k, t, l: l,
    "x_preloop":  lambda k, t, l: not l,
    "k_fma":      lambda k, t, l: k != "fma",                   # k_<class>: ONLY this class stays packed
    "k_mul":      lambda k, t, l: k != "mul",
    "k_add":      lambda k, t, l: k != "add",
    "k_sgpr":     lambda k, t, l: "s[" not in t,
    "k_loop_sgpr": lambda k, t, l: not (l and "s[" in t),
    "k_loop_vgpr": lambda k, t, l: not (l and "s[" not in t),
    "k_modifiers": lambda k, t, l: not ("op_sel" in t or "neg_" in t),       # only instructions that carry op_sel / neg modifiers stay packed
    "k_plain":    lambda k, t, l: ("op_sel" in t or "neg_" in t),            # only modifier-free instructions stay packed
    "k_xhalf":    lambda k, t, l: not xhalf(t),                              # only instructions that read a VGPR pair across halves stay packed
    "x_xhalf":    lambda k, t, l: xhalf(t),                                  # only those are expanded
}
# all 62 instructions stay packed, wait states added: is the compiler's hazard padding (s_nop 0 between a VALU write and a cross-half read) too short?
PADDED = {
    "pad_nops4":   dict(nop_scale=4),                                                        # every compiler s_nop becomes s_nop 4
    "pad_xhalf":   dict(pre=lambda k, t: ["\ts_nop 3"] if xhalf(t) else []),                  # 4 wait states before every cross-half reader
    "pad_all":     dict(pre=lambda k, t: ["\ts_nop 3"]),                                      # 4 wait states before EVERY packed instruction
    "pad_all1":    dict(pre=lambda k, t: ["\ts_nop 0"]),                                      # 1 wait state before every packed instruction
}


def main():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "openstereo_amd", "csrc", "softargmin.hip")
    base = os.path.join(OUT, "head_packed.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", base, src], stderr=subprocess.DEVNULL)
    lines = open(base).read().split("\n")
    # third round: exactly ONE of the pre-loop cross-half v_pk_add_f32 instructions stays packed (k_one<ordinal>), everything else scalar
    ordinal = 0
    inside = False
    for ln in lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        elif inside and ln.startswith(".LBB3_12:"):
            break
        if inside and PK.match(ln):
            if PK.match(ln).group(1) == "add" and xhalf(ln):
                VARIANTS[f"k_one{ordinal:02d}"] = (lambda o: (lambda k, t, l, i: i != o))(ordinal)
                print(f"k_one{ordinal:02d}: {ln.strip()}")
            ordinal += 1
    jobs = [(n, dict(pick=p)) for n, p in VARIANTS.items()] + [(n, dict(pick=lambda k, t, l: False, **kw)) for n, kw in PADDED.items()]
    for name, kw in jobs:
        txt, n_exp, n_keep = rewrite(lines, **kw)
        s = os.path.join(OUT, name + ".s")
        open(s, "w").write(txt)
        o, co = os.path.join(OUT, name + ".o"), os.path.join(OUT, name + ".co")
        subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
        subprocess.check_call([f"{LLVM}/ld.lld", "-shared", o, "-o", co])
        os.remove(o)
        print(f"{name:12s}: {n_exp:2d} packed instructions expanded, {n_keep:2d} kept -> {os.path.relpath(co, ROOT)}")


if __name__ == "__main__":
    sys.exit(main())
