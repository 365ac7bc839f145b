"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/musicgan_hip.h declares (no compute
calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "musicgan_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("mg_conv3x3", "mg_conv3x3_pack", "mg_conv3x3_wgrad", "mg_conv1x1", "mg_conv1x1_wgrad", "mg_adam_step",
                 "mg_stft_1024", "mg_pixelnorm_fwd", "mg_gp_finish", "mg_last_error"):
        assert must in syms
    assert len(syms) >= 28


def test_library_builds_loads_and_exports_every_declared_symbol():
    from musicgan_amd import _build, _lib
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    raw = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} declared in include/musicgan_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype in musicgan_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in declared_symbols(), f"{name} bound but not declared in the header"
    assert lib.mg_version() >= 100
    # pure host-side size queries are callable without a GPU
    assert lib.mg_conv3x3_packed_floats(64, 48) == 8 * 72 * 48
    assert lib.mg_conv3x3_packed_floats(8, 8) == 1 * 72 * 16
    assert lib.mg_conv3x3_wgrad_ws_bytes(64, 64, 48, 128, 128) > 0
    assert lib.mg_conv1x1_wgrad_ws_bytes(64, 2, 48, 128 * 128) > 0


def test_adam_descriptor_layout_matches_header():
    from musicgan_amd._lib import AdamTensor
    assert ctypes.sizeof(AdamTensor) == 48  # 4 pointers + int64 + 2 floats


def test_ops_refuse_cpu_tensors_loudly():
    import pytest
    import torch
    from musicgan_amd import _lib, ops
    from musicgan_amd.networks import Generator
    with pytest.raises(_lib.MusicGanHipError):
        ops.axpby(1.0, torch.zeros(4))
    gen = Generator(8)
    with pytest.raises(_lib.MusicGanHipError):
        gen(torch.zeros(1, 8, 2, 2), 1.0)
