"""CPU: the C-ABI library loads and exports every symbol include/openstereo_amd.h declares (no
compute without a GPU), the ctypes table mirrors the header, and the product refuses CPU tensors."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "openstereo_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(osa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from openstereo_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    for n in names:
        assert re.search(rf"\bT {n}\b", exported), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} missing from the ctypes table"
        assert hasattr(lib, n)
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_argument_counts_match_header():
    from openstereo_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(rf"\b{name}\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} params, ctypes {len(args)}"


def test_version_and_target(lib):
    from openstereo_amd import _lib as L
    assert lib.osa_abi_version() == L.abi_version() >= 4
    assert lib.osa_target_arch() == b"gfx950"


def test_argument_validation_reports_errors_without_gpu(lib):
    """Validation happens before any launch, so it is testable on CPU."""
    rc = lib.osa_build_volume_f32(None, None, 10, 3, None, None, 0, None, 0, 3, 0, 1, 4, 8, 4, 1, None, None)
    assert rc != 0 and b"vol is NULL" in lib.osa_last_error()
    rc = lib.osa_build_volume_f32(1, 1, 10, 3, None, None, 0, 1, 0, 3, 0, 1, 4, 8, 4, 1, None, None)
    assert rc != 0 and b"not divisible" in lib.osa_last_error()          # cost_volume.py:61
    slack = 4 * 2 * 2 * 32 * 4                      # four prefetched tap steps
    assert lib.osa_conv3d_packed_floats(32, 32, 3, 3, 3) == (2 * 27 * 2 * 2 * 32 * 4) + slack
    assert lib.osa_deconv3d_packed_floats(64, 32, 3) == (4 * 27 * 2 * 2 * 32 * 4) + slack


def test_product_has_no_cpu_path():
    from openstereo_amd import ops, _lib
    x = torch.zeros(1, 8, 4, 8)
    with pytest.raises(_lib.EngineError):
        ops.build_gwc_volume(x, x, 4, 2)
    with pytest.raises(_lib.EngineError):
        ops.disparity_regression(torch.zeros(1, 4, 4, 4), 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "openstereo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_state_dict_layout_matches_reference_names():
    """Key names the reference's checkpoints use (DispProcessor.dres0.0.0.weight ...)."""
    from openstereo_amd.models.gwcnet import GwcNet
    sd = GwcNet().state_dict()
    for k in ("Backbone.feature_extraction.firstconv.0.0.weight", "Backbone.feature_extraction.layer2.0.downsample.0.weight",
              "Backbone.feature_extraction.lastconv.2.weight", "DispProcessor.dres0.0.0.weight",
              "DispProcessor.dres2.conv5.0.weight", "DispProcessor.dres4.redir2.1.running_var",
              "DispProcessor.classif3.2.weight"):
        assert k in sd, k
    assert len(sd) == 533
    assert sd["DispProcessor.dres0.0.0.weight"].shape == (32, 64, 3, 3, 3)
    assert sd["DispProcessor.dres2.conv5.0.weight"].shape == (128, 64, 3, 3, 3)


def test_argument_validation_of_round1_additions(lib):
    """Depthwise conv, 2-D transposed conv, fused redir branch, GRU combine: argument errors are reported
    (non-zero return + message) before anything is launched."""
    f = 16                                                      # any non-NULL, 16-byte aligned "pointer"
    rc = lib.osa_dwconv2d_nhwc_f32(f, f, None, None, None, f, 1, 8, 8, 6, 8, 8, 0, 3, 3, 1, 1, 1, 1, 1, 0, None, None)
    assert rc != 0 and b"multiples of 4" in lib.osa_last_error()         # C = 6
    rc = lib.osa_dwconv2d_nhwc_f32(f, f, None, None, None, f, 1, 8, 8, 8, 8, 8, 0, 3, 3, 3, 1, 1, 1, 1, 0, None, None)
    assert rc != 0 and b"stride" in lib.osa_last_error()
    rc = lib.osa_dwconv2d_nhwc_f32(None, f, None, None, None, f, 1, 8, 8, 8, 8, 8, 0, 3, 3, 1, 1, 1, 1, 1, 0, None, None)
    assert rc != 0 and b"NULL" in lib.osa_last_error()
    rc = lib.osa_deconv2d_nhwc_f32(f, f, None, None, None, f, 1, 4, 4, 8, 8, 8, 8, 0, 5, 1, 1, None, 0, 0, 0.0, None)
    assert rc != 0 and b"only (k=3" in lib.osa_last_error()              # k = 5 unsupported
    rc = lib.osa_deconv3d_redir_ndhwc_f32(f, f, None, None, f, 1, 2, 4, 4, 64, 64, 32, 32, 3, 1, 1,
                                          f, 128, 128, f, None, None, 1, 0.0, None)
    assert rc != 0 and b"64 channels" in lib.osa_last_error()            # redir input wider than 64 channels
    rc = lib.osa_gru_combine_f32(f, f, f, f, 10, 6, 8, 8, 8, 8, None, None)
    assert rc != 0 and b"multiple of 4" in lib.osa_last_error()
    assert lib.osa_deconv2d_packed_floats(64, 32, 3) == (4 * 9 * 2 * 2 * 32 * 4) + 4 * 2 * 2 * 32 * 4
