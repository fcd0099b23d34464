"""CPU: no kernel of the shipped gfx950 library carries a packed-fp32 VALU instruction (v_pk_mul / add / fma_f32).

r5 found the x4 fused head returning wrong 16-lane passes next to another stream's d-marching convolution only when its loop carried
these instructions.  Every kernel of a forward can be co-resident with a marching kernel in the timed configuration
(three sub-batch streams), so the property is enforced for the whole library at build time (openstereo_amd/build.py NO_PACKED_F32) and
checked here on the object code that ships -- a compiler bump, a new translation unit built with other flags or a hand-written
`v_pk_*_f32` fails this test instead of a user's disparity map.  The GPU side of the same contract is tests/test_gpu_concurrency.py."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernels allowed to carry packed-fp32 instructions: NONE.  An entry here needs a case in tests/test_gpu_concurrency.py that runs the
# kernel >= 200 times next to both marching instances.
ALLOW: set = set()


def test_no_packed_fp32_instruction_in_any_shipped_kernel(lib):
    import isa_lint
    from openstereo_amd import _lib, build
    assert build.NO_PACKED_F32[-1] == "-packed-fp32-ops" and all(f in build.HIPCC_FLAGS for f in build.NO_PACKED_F32)
    assert not build.EXTRA_FLAGS, "per-file flags must not re-enable what the global flags switch off"
    res = isa_lint.scan(_lib.LIB_PATH)
    assert len(res) > 300, f"only {len(res)} kernels found: the disassembly did not see the whole library"
    assert any("conv_march_kernel" in k for k in res) and any("upsample4_softargmin_kernel" in k for k in res)
    bad = {k: v[0] for k, v in res.items() if v[0] and k not in ALLOW}
    assert not bad, f"{len(bad)} kernels carry packed-fp32 instructions, e.g. {sorted(bad.items(), key=lambda kv: -kv[1])[:5]}"
