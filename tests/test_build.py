"""The gfx950 build: every source compiles, and the matrix-core kernels keep their accumulators in registers.

A kernel whose accumulator array ends up in scratch memory (a lambda that stopped being inlined, a loop whose invariants were
hoisted) still reports "0 spilled registers" and passes every parity test -- at a quarter of the speed.  The build keeps hipcc's
resource remarks per kernel (musicgan_amd/_build.py); this test bounds the scratch size of the kernels that matter."""
import re

from musicgan_amd import _build

# (regex on the mangled name, allowed scratch bytes per lane)
HOT = [
    (r"wino3x3_mfmaILi2ELi2ELi4E", 0),
    (r"wino3x3_mfmaILi2ELi2ELi2E", 0),
    (r"wino3x3_mfmaILi1ELi3ELi4E", 16),   # 12-wave tiling at its 168-register ceiling: three spilled dwords outside the MFMA loop
    (r"wino3x3_mfmaILi1ELi[23]ELi2E", 0),
    (r"wino3x3_mfmaILi1ELi2ELi4E", 0),
    (r"wino3x3_stripILi[12]E", 0),        # one / two out-channel tiles per wave: two waves per SIMD, everything in registers
    (r"wino3x3_stripILi3E", 0),           # three tiles: 192 accumulators, one wave per SIMD (AGPRs, no scratch)
    (r"wino_wgrad_mfma", 0),
    (r"wino_wgrad_rows_mfma", 0),         # row-staged form: up to 256 accumulators + one raw operand set, two waves per SIMD
    (r"wino_wgrad_narrow_mfmaILi2ELi2ELb0ELb0E", 28),  # the general (per-lane addressed) 2 x 2 form: 7 dwords; off the default path
                                                       # (maps >= 16 wide take the scalar-addressed form, 2 x 2 blocks the row-staged one)
    (r"wino_wgrad_narrow_mfma(?!ILi2ELi2ELb0ELb0E)", 0),
    (r"wino_wgrad_group_mfma", 0),
    (r"conv3x3_mfma", 0),
    (r"wgrad3x3_mfma", 0),
    (r"upconv3x3_mfma", 0),
    (r"downconv4x4s2_mfma", 0),
    (r"smallconv_k", 0),
    (r"smallnet_k", 0),
    (r"crc32_f64_chunks_k", 0),
    (r"stft1024_kernel", 0),
    (r"codec_row_pass", 0),
]


def test_hot_kernels_do_not_use_scratch_memory():
    _build.build()
    usage = _build.resource_usage()
    assert len(usage) > 100, "resource remarks missing: was the library built by musicgan_amd._build?"
    for pat, limit in HOT:
        hits = {k: v for k, v in usage.items() if re.search(pat, k)}
        assert hits, f"no kernel matches {pat}"
        for name, u in hits.items():
            scratch = u.get("ScratchSize [bytes/lane]", 0)
            assert scratch <= limit, f"{name}: {scratch} B/lane of scratch memory (allowed {limit}); VGPRs {u.get('VGPRs')}"
            assert u.get("VGPRs", 0) > 0
