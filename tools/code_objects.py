"""Unbundle and disassemble the gfx950 code objects of a HIP shared library (no GPU needed).

Used by tests/test_code_objects.py to guard the packed-fp32 hazard of DESIGN.md section 7 (packed fp32 VALU
instructions whose LOW result lane takes the HIGH half of a source -- `op_sel` -- returned wrong values in lanes
48-63 beside the VAE's 3x3x3 convolution), and handy for reading a kernel's ISA:

    python tools/code_objects.py [lib.so] [--kernel NAME] [--grep REGEX]

A hipcc-linked library keeps one `__CLANG_OFFLOAD_BUNDLE__` blob per translation unit back to back in its
`.hip_fatbin` section; `clang-offload-bundler` only sees the first, so the bundle headers are parsed here
(magic, entry count, then {offset, size, triple length, triple} per entry).
"""
from __future__ import annotations

import argparse
import os
import re
import struct
import subprocess
import tempfile
from typing import Dict, Iterator, List, Tuple

LLVM_BIN = os.environ.get("SF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path: str, arch: str = "gfx950") -> List[bytes]:
    """The device code objects for `arch` embedded in `lib_path`, one per translation unit."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path],
                       check=True, capture_output=True)
        blob = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        (n_entries,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n_entries):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            pos += 24
            triple = blob[pos:pos + tlen].decode()
            pos += tlen
            if arch in triple and size:
                out.append(blob[base + off:base + off + size])
    return out


def disassemble(code_object: bytes, arch: str = "gfx950") -> str:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object)
        f.flush()
        res = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", f"--mcpu={arch}", f.name],
                             check=True, capture_output=True, text=True)
    return res.stdout


def instructions(lib_path: str, arch: str = "gfx950") -> Iterator[Tuple[str, str]]:
    """(kernel symbol, instruction text) for every instruction of every device function in the library."""
    label = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")
    for co in code_objects(lib_path, arch):
        kernel = "?"
        for line in disassemble(co, arch).splitlines():
            m = label.match(line.strip())
            if m:
                kernel = m.group(1)
                continue
            line = line.strip()
            if not line or line.startswith("Disassembly") or line.endswith("file format elf64-amdgpu"):
                continue
            text = line.split("//")[0].strip()
            if text:
                yield kernel, text


PACKED_F32 = re.compile(r"^v_pk_(mul|add|fma)_f32\b")


def packed_f32_census(lib_path: str) -> Dict[str, Dict[str, int]]:
    """kernel -> {instruction form -> count} for the packed fp32 VALU instructions; the form keeps the modifiers
    (op_sel / op_sel_hi / neg_lo / neg_hi) and drops the registers."""
    out: Dict[str, Dict[str, int]] = {}
    for kernel, text in instructions(lib_path):
        if not PACKED_F32.match(text):
            continue
        mods = " ".join(re.findall(r"(?:op_sel_hi|op_sel|neg_lo|neg_hi):\[[0-9,]+\]", text))
        form = (text.split()[0] + " " + mods).strip()
        out.setdefault(kernel, {}).setdefault(form, 0)
        out[kernel][form] += 1
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(here, "..", "self-forcing_amd", "csrc", "libsf_hip.so"))
    ap.add_argument("--kernel", default="")
    ap.add_argument("--grep", default="")
    ap.add_argument("--census", action="store_true", help="packed fp32 instruction forms per kernel")
    a = ap.parse_args()
    if a.census:
        for k, forms in sorted(packed_f32_census(a.lib).items()):
            for form, n in sorted(forms.items()):
                print(f"{n:6d}  {form:50s} {k}")
        return
    rx = re.compile(a.grep) if a.grep else None
    for kernel, text in instructions(a.lib):
        if a.kernel and a.kernel not in kernel:
            continue
        if rx and not rx.search(text):
            continue
        print(f"{kernel}: {text}")


if __name__ == "__main__":
    main()
