"""Static guard over the SHIPPED device code (no GPU): the packed-fp32 hazard of DESIGN.md section 7.

Measured on MI355X (tools/probes/pk_probe.*): `v_pk_mul_f32` / `v_pk_add_f32` whose LOW result lane takes the HIGH
half of a source (`op_sel:[0,1]`; hipcc's SLP vectoriser forms exactly that for a complex multiply) returned wrong
values in lanes 48-63 whenever waves of the VAE's 3x3x3 convolution shared the CUs.  The library is therefore built
with -fno-slp-vectorize; this test makes that property of the BINARY a checked one, so that a stale object file, a
changed flag or a hand-written packed instruction cannot bring the hazardous form back unnoticed."""
import os
import sys

import self_forcing_amd as sfa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import code_objects as co  # noqa: E402

# kernels that may hold packed fp32 at all, and in which exact form: the generated instruction stream of the
# 64-row attention kernel scales O by exp2(m_old - m_new) with plain v_pk_mul_f32 in its (rare) rescale path; its
# waves own their SIMD's whole register file (one wave per SIMD) and never run beside another wave's MFMAs.
ALLOWED = {"attention_r64_kernel": {"v_pk_mul_f32"}}


def _lib_path():
    path = sfa._lib.LIB_PATH
    if not os.path.exists(path):
        sfa._lib.build()
    return path


def test_one_code_object_per_translation_unit():
    objs = co.code_objects(_lib_path())
    assert len(objs) >= 7, f"expected the device code of every .hip source, found {len(objs)} gfx950 code objects"
    kernels = {k for k, _ in co.instructions(_lib_path())}
    for name in ("gemm_bf16_kernel", "attention_r64_kernel", "conv_igemm_kernel", "qkv_norm_rope_cache_kernel"):
        assert any(name in k for k in kernels), f"{name} not found in the disassembly"


def test_no_hazardous_packed_fp32_in_shipped_kernels():
    census = co.packed_f32_census(_lib_path())
    bad = []
    for kernel, forms in census.items():
        allowed = next((v for k, v in ALLOWED.items() if k in kernel), set())
        for form, n in forms.items():
            if "op_sel:" in form:                    # (op_sel_hi alone is a different modifier and matches "op_sel_hi:")
                bad.append(f"{kernel}: {n} x {form}  <- the miscomputing form")
            elif form not in allowed:
                bad.append(f"{kernel}: {n} x {form}  <- packed fp32 outside the allow-list (built without -fno-slp-vectorize?)")
    assert not bad, "\n".join(bad)
    # the allow-listed kernel really is the only holder, and only of the plain form
    assert all(any(k in kernel for k in ALLOWED) for kernel in census), census.keys()


def test_compressed_bundle_headers_are_cut_by_their_total_size():
    """`.hip_fatbin` sections of PyTorch's own libraries hold zstd-compressed `CCOB` bundles back to back (libtorch_hip.so:
    274 of them, no plain one): the census tool cuts them out by the total size in each header (version 2: u32, version 3:
    u64) and skips a 'CCOB' that does not parse as a header (bytes inside compressed data)."""
    import struct
    v2 = b"CCOB" + struct.pack("<HHIIQ", 2, 1, 24 + 10, 100, 0xABCD) + b"x" * 10
    v3 = b"CCOB" + struct.pack("<HHQQQ", 3, 1, 32 + 7, 55, 0x1234) + b"CCOByyy"          # payload that contains the magic
    blob = b"\0" * 5 + v2 + b"\0" * 3 + v3 + b"CCOB" + struct.pack("<HH", 9, 7) + b"junk"
    got = list(co.compressed_bundles(blob))
    assert got == [v2, v3]
    assert list(co.compressed_bundles(b"no bundles here")) == []
