"""Unbundle and disassemble the gfx950 code objects of a HIP shared library (no GPU needed).

Used by tests/test_code_objects.py to guard the packed-fp32 hazard of DESIGN.md section 7 (packed fp32 VALU
instructions whose LOW result lane takes the HIGH half of a source -- `op_sel` -- returned wrong values in lanes
48-63 beside the VAE's 3x3x3 convolution), and handy for reading a kernel's ISA:

    python tools/code_objects.py [lib.so] [--kernel NAME] [--grep REGEX]

A hipcc-linked library keeps one offload bundle per translation unit back to back in its `.hip_fatbin` section;
`clang-offload-bundler` only sees the first, so the section is cut into bundles here: plain ones
(`__CLANG_OFFLOAD_BUNDLE__`: magic, entry count, then {offset, size, triple length, triple} per entry) are parsed
directly, COMPRESSED ones (`CCOB`: magic, u16 version, u16 method, total size, ...; what PyTorch's own libraries ship:
libtorch_hip.so holds 274 zstd-compressed bundles and no plain one) are cut out by their header's total size and handed
to `clang-offload-bundler --unbundle` one by one.

    python tools/code_objects.py --census-stream /path/to/libtorch_hip.so [--kernel-regex RX] [--json out.json]

streams the disassembly of a large library bundle by bundle (nothing is kept in memory) and counts the packed-fp32
forms per kernel -- the census of the torch kernels that run inside a rollout.
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
MAGIC_COMPRESSED = b"CCOB"


def _fatbin(lib_path: str) -> bytes:
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path],
                       check=True, capture_output=True)
        return open(fat, "rb").read()


def _plain_entries(blob: bytes, base: int, arch: str) -> Iterator[bytes]:
    (n_entries,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
    pos = base + len(MAGIC) + 8
    for _ in range(n_entries):
        off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
        pos += 24
        triple = blob[pos:pos + tlen].decode()
        pos += tlen
        if arch in triple and size:
            yield blob[base + off:base + off + size]


def compressed_bundles(blob: bytes) -> Iterator[bytes]:
    """The `CCOB` bundles of a .hip_fatbin section, each cut out by the total size its header states (version 2:
    u32 total, u32 uncompressed, u64 hash; version 3: u64, u64, u64); a 'CCOB' that does not parse as a header
    (bytes inside compressed data) is skipped."""
    pos = 0
    while True:
        pos = blob.find(MAGIC_COMPRESSED, pos)
        if pos < 0:
            return
        ver, method = struct.unpack_from("<HH", blob, pos + 4)
        if ver == 2:
            (total,) = struct.unpack_from("<I", blob, pos + 8)
        elif ver == 3:
            (total,) = struct.unpack_from("<Q", blob, pos + 8)
        else:
            total = 0
        if ver not in (2, 3) or method > 1 or total < 24 or pos + total > len(blob):
            pos += 4
            continue
        yield blob[pos:pos + total]
        pos += total


def _unbundle_compressed(bundle: bytes, arch: str) -> bytes:
    """gfx`arch` code object of one compressed bundle, b"" if it has none (clang-offload-bundler decompresses it)."""
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "bundle.bin"), os.path.join(tmp, "out.co")
        open(src, "wb").write(bundle)
        tool = os.path.join(LLVM_BIN, "clang-offload-bundler")
        listed = subprocess.run([tool, "--list", "--type=o", f"--input={src}"], capture_output=True, text=True)
        target = next((t for t in listed.stdout.split() if t.endswith(arch) and t.startswith("hip")), None)
        if listed.returncode != 0 or target is None:
            return b""
        subprocess.run([tool, "--unbundle", "--type=o", f"--targets={target}", f"--input={src}", f"--output={dst}"],
                       check=True, capture_output=True)
        return open(dst, "rb").read()


def iter_code_objects(lib_path: str, arch: str = "gfx950") -> Iterator[bytes]:
    """The device code objects for `arch` embedded in `lib_path` (plain and compressed bundles), one at a time."""
    blob = _fatbin(lib_path)
    for m in re.finditer(re.escape(MAGIC), blob):
        yield from _plain_entries(blob, m.start(), arch)
    for bundle in compressed_bundles(blob):
        co = _unbundle_compressed(bundle, arch)
        if co:
            yield co


def code_objects(lib_path: str, arch: str = "gfx950") -> List[bytes]:
    """All of them at once (small libraries: ours holds one per translation unit)."""
    return list(iter_code_objects(lib_path, arch))


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


def census_stream(lib_path: str, kernel_regex: str = "", arch: str = "gfx950", progress=None) -> Dict[str, Dict[str, int]]:
    """`packed_f32_census` for libraries too large to hold disassembled (libtorch_hip.so: 274 bundles): one code object
    at a time, llvm-objdump's output filtered by grep to symbol labels and packed-fp32 lines before Python sees it.
    kernel_regex restricts the result to matching (mangled) kernel names.  Returns kernel -> {form -> count} for
    kernels that hold ANY packed fp32 instruction."""
    label = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")
    rx = re.compile(kernel_regex) if kernel_regex else None
    out: Dict[str, Dict[str, int]] = {}
    n_kernels = 0
    for i, co in enumerate(iter_code_objects(lib_path, arch)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dump = subprocess.Popen([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", f"--mcpu={arch}", f.name], stdout=subprocess.PIPE)
            flt = subprocess.Popen(["grep", "-E", r"^[0-9a-f]+ <|v_pk_(mul|add|fma)_f32"], stdin=dump.stdout, stdout=subprocess.PIPE, text=True)
            dump.stdout.close()
            kernel = "?"
            for line in flt.stdout:
                line = line.strip()
                m = label.match(line)
                if m:
                    kernel = m.group(1)
                    n_kernels += 1
                    continue
                if rx and not rx.search(kernel):
                    continue
                text = line.split("//")[0].strip()
                mods = " ".join(re.findall(r"(?:op_sel_hi|op_sel|neg_lo|neg_hi):\[[0-9,]+\]", text))
                form = (text.split()[0] + " " + mods).strip()
                out.setdefault(kernel, {}).setdefault(form, 0)
                out[kernel][form] += 1
            flt.wait()
            dump.wait()
        if progress:
            progress(i, n_kernels, len(out))
    out["__symbols_scanned__"] = {"count": n_kernels}
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(here, "..", "self-forcing_amd", "csrc", "libsf_hip.so"))
    ap.add_argument("--kernel", default="")
    ap.add_argument("--grep", default="")
    ap.add_argument("--census", action="store_true", help="packed fp32 instruction forms per kernel")
    ap.add_argument("--census-stream", action="store_true", help="--census for a large library, bundle by bundle (compressed bundles too)")
    ap.add_argument("--kernel-regex", default="", help="with --census-stream: only kernels whose mangled name matches")
    ap.add_argument("--json", default="", help="with --census-stream: write the result here")
    a = ap.parse_args()
    if a.census_stream:
        import json
        import sys
        res = census_stream(a.lib, a.kernel_regex, progress=lambda i, n, k: print(f"bundle {i}: {n} symbols, {k} with packed fp32", file=sys.stderr, flush=True))
        if a.json:
            json.dump(res, open(a.json, "w"), indent=1, sort_keys=True)
        hazard = {k: v for k, v in res.items() if any("op_sel:" in f for f in v)}
        print(f"{res['__symbols_scanned__']['count']} symbols scanned; {len(res) - 1} hold packed fp32; {len(hazard)} hold an op_sel form")
        for k, forms in sorted(hazard.items()):
            print(k, {f: n for f, n in forms.items() if "op_sel:" in f})
        return
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
