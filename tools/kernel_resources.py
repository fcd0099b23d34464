"""Register / occupancy table of a HIP source's kernels (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py openstereo_amd/csrc/conv3d.hip [substring ...]

Part of the kernel workflow: an innocuous edit that costs a tile one wave per SIMD is worth
tens of percent, so the table is checked after every change of the conv kernel."""
import re
import subprocess
import sys

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]   # openstereo_amd/build.py HIPCC_FLAGS


def table(src, extra=()):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split("\n")[0].strip()
        g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs"),
                     g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(n.replace("osa::", "").replace("(ConvArgs)", ""),) + r[1:] for n, r in zip(names, rows)]


if __name__ == "__main__":
    flt = sys.argv[2:]
    print("vgpr agpr scratch occ sgpr lds  kernel")
    for n, v, a, sc, occ, sg, lds in table(sys.argv[1]):
        if not flt or any(f in n for f in flt):
            print(f"{v:4d} {a:4d} {sc:7d} {occ:3d} {sg:4d} {lds:5d} {n[:110]}")
