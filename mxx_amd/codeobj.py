"""Identity of the device code inside libgpupoly.so: a hash of every kernel's ISA text.

Measurement plumbing, not part of the hot path.  The roofline records under `profiles/` carry instruction and byte counts
collected by `rocprofv3 --pmc` passes of an earlier run; `bench.py` prices today's kernels with them only while the kernel
they were counted on is still the kernel in the library.  The identity is taken from the library itself - the gfx950 code
objects hipcc embedded in `.hip_fatbin` - so it needs no git metadata (a GPU box gets a snapshot without `.git`) and no GPU.

Layout read here (clang offload bundle, uncompressed, one bundle per translation unit):
    "__CLANG_OFFLOAD_BUNDLE__" | u64 entries | entries x { u64 offset, u64 size, u64 triple_len, triple } | code objects
and each `hipv4-amdgcn-amd-amdhsa--gfx950` entry is an ELF64 whose STT_FUNC symbols are the kernels (their `.kd`
descriptors - register counts, LDS size - are STT_OBJECT symbols of the same name + ".kd" in .rodata).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import struct
import subprocess

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_cache: dict = {}


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgpupoly.so")


def kernel_base(name: str) -> str:
    """`void ntt14::fwd_kernel<unsigned int, false>(unsigned int*, ...)` -> `ntt14::fwd_kernel` (all template
    instances of a kernel pool under one name, as in the profiles' records)"""
    s = name.strip().strip("()").strip()
    if s.startswith("void "):
        s = s[5:]
    for ch in "<(":
        k = s.find(ch)
        if k > 0:
            s = s[:k]
    return s.strip()


def _code_objects(blob: bytes, arch: str):
    pos = 0
    while True:
        pos = blob.find(_MAGIC, pos)
        if pos < 0:
            return
        (count,) = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        cur = pos + len(_MAGIC) + 8
        if count > 64:  # not a header (the magic can only be followed by a small entry count)
            pos += len(_MAGIC)
            continue
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24 : cur + 24 + tlen].decode("ascii", "replace")
            cur += 24 + tlen
            if arch in triple and size:
                yield blob[pos + off : pos + off + size]
        pos += len(_MAGIC)


def _elf_functions(elf: bytes):
    """(name, text bytes, descriptor bytes) of every STT_FUNC symbol of an ELF64 little-endian code object"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append({"type": typ, "addr": addr, "off": off, "size": size, "link": link, "entsize": entsize})
    for sec in secs:
        if sec["type"] != 2:  # SHT_SYMTAB
            continue
        strtab = secs[sec["link"]]
        syms = {}
        for j in range(sec["size"] // 24):
            st_name, st_info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, sec["off"] + j * 24)
            end = elf.find(b"\0", strtab["off"] + st_name)
            nm = elf[strtab["off"] + st_name : end].decode("ascii", "replace")
            if 0 < shndx < len(secs) and size:
                s = secs[shndx]
                if s["type"] == 8:  # SHT_NOBITS
                    continue
                start = s["off"] + (value - s["addr"])
                syms[nm] = (st_info & 0xF, elf[start : start + size])
        for nm, (typ, body) in syms.items():
            if typ == 2:  # STT_FUNC
                yield nm, body, syms.get(nm + ".kd", (1, b""))[1]


def _demangle(names):
    tool = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        if len(out) >= len(names):
            return dict(zip(names, out))
    except Exception:  # noqa: BLE001 - no demangler: the mangled names are still stable keys
        pass
    return {n: n for n in names}


def kernel_isa_hashes(path: str | None = None, arch: str = "gfx950") -> dict:
    """{kernel base name: 16 hex digits} over the ISA text + kernel descriptor of every instance of that kernel in the
    library, plus "*" = all device code.  Empty when the library holds no code object for `arch`."""
    path = path or library_path()
    key = (path, os.path.getmtime(path), arch)
    if key in _cache:
        return _cache[key]
    blob = open(path, "rb").read()
    funcs = []
    for elf in _code_objects(blob, arch):
        funcs.extend(_elf_functions(elf))
    names = _demangle(sorted({f[0] for f in funcs}))
    groups: dict = {}
    for nm, text, kd in funcs:
        groups.setdefault(kernel_base(names[nm]), []).append((nm, text, kd))
    out = {}
    everything = hashlib.sha256()
    for base in sorted(groups):
        h = hashlib.sha256()
        for nm, text, kd in sorted(groups[base]):
            for part in (nm.encode(), text, kd):
                h.update(struct.pack("<Q", len(part)))
                h.update(part)
        out[base] = h.hexdigest()[:16]
        everything.update(h.digest())
    if out:
        out["*"] = everything.hexdigest()[:16]
    _cache[key] = out
    return out


def stale_kernels(recorded: dict | None, current: dict | None = None, kernels=None) -> list:
    """kernels of `recorded` ({base: hash}, as stored in a profiles record) whose ISA differs from the library's today;
    a record without hashes is stale as a whole (["*"])"""
    if not recorded:
        return ["*"]
    current = kernel_isa_hashes() if current is None else current
    names = [k for k in (kernels if kernels is not None else recorded) if k != "*"]
    return [k for k in names if recorded.get(k) is None or recorded.get(k) != current.get(k)]


if __name__ == "__main__":
    import json
    import sys

    print(json.dumps(kernel_isa_hashes(sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
