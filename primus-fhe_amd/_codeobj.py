"""Register / scratch figures of the kernels INSIDE the built libpfhe_hip.so, read from the gfx950 code objects'
own metadata (no compiler run, no GPU): the .hip_fatbin section holds one clang offload bundle per translation unit,
each with a gfx950 ELF whose AMDGPU note (msgpack) lists every kernel with its vgpr_count, sgpr_count, LDS and scratch
size.  bench.py uses it to check that a committed counter profile was taken on the kernels it is about to time."""
from __future__ import annotations

import re
import struct
import subprocess

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(data: bytes, want: bytes):
    """(offset, size) of ELF64 section `want`."""
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    str_off = hdr(shstrndx)[4]
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = hdr(i)[:6]
        end = data.index(b"\0", str_off + name)
        if data[str_off + name:end] == want:
            return off, size, typ
    return None


def _notes(elf: bytes):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for i in range(shnum):
        _name, typ, _f, _a, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        if typ != 7:  # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    except Exception:
        return list(names)
    return [re.sub(r"\(.*", "", n.replace("void ", "").replace("pfhe::(anonymous namespace)::", "").replace("pfhe::", ""))
            for n in out]


def _gfx950_elfs(so_path: str):
    """every gfx950 code object (ELF image) inside the library's .hip_fatbin section"""
    data = open(so_path, "rb").read()
    sec = _section(data, b".hip_fatbin")
    if sec is None:
        return
    fat = data[sec[0]:sec[0] + sec[1]]
    for m in re.finditer(re.escape(_MAGIC), fat):
        o = m.start()
        num, = struct.unpack_from("<Q", fat, o + 24)
        p = o + 32
        for _ in range(num):
            eo, es, tl = struct.unpack_from("<QQQ", fat, p)
            p += 24
            triple = fat[p:p + tl]
            p += tl
            if b"gfx950" in triple and es:
                yield fat[o + eo:o + eo + es]


def _func_symbols(elf: bytes):
    """(name, bytes of the function) for every FUNC symbol of an ELF64 image (.symtab + the section it lives in)"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
    for i in range(shnum):
        _name, typ, _f, _a, off, size, link, _info, _al, entsize = hdr(i)
        if typ != 2 or not entsize:  # SHT_SYMTAB
            continue
        str_off = hdr(link)[4]
        for j in range(size // entsize):
            st_name, st_info, _other, shndx, value, st_size = struct.unpack_from("<IBBHQQ", elf, off + j * entsize)
            if (st_info & 0xF) != 2 or not st_size or not 0 < shndx < shnum:  # STT_FUNC, defined
                continue
            _n, _t, _fl, s_addr, s_off, s_size = hdr(shndx)[:6]
            begin = s_off + (value - s_addr)
            if value < s_addr or begin + st_size > s_off + s_size:
                continue
            end = elf.index(b"\0", str_off + st_name)
            yield elf[str_off + st_name:end].decode(), elf[begin:begin + st_size]


def kernel_code_hashes(so_path: str) -> dict:
    """{kernel name as the profiles print it: first 16 hex digits of the SHA-256 of the kernel's machine code} — the bytes
    of the kernel's function symbol in its gfx950 code object.  A counter profile names the hashes of the kernels it was
    taken on; bench.py reports the profile's traffic only while the library it times still holds the same code (a kernel can
    change what it moves without changing its register allocation)."""
    import hashlib

    rows = {}
    for elf in _gfx950_elfs(so_path):
        kernels = set()
        try:
            import msgpack
            for name, ntype, desc in _notes(elf):
                if name == b"AMDGPU" and ntype == 32:
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    kernels |= {k[".name"] for k in meta.get("amdhsa.kernels", [])}
        except Exception:
            kernels = None
        for name, code in _func_symbols(elf):
            if kernels is None or name in kernels:
                rows[name] = hashlib.sha256(code).hexdigest()[:16]
    names = list(rows)
    return {pretty: rows[n] for n, pretty in zip(names, _demangle(names))}


def kernel_resources(so_path: str) -> dict:
    """{kernel name as the profiles print it: {"vgpr": .., "agpr": .., "sgpr": .., "scratch": .., "lds": ..}}"""
    import msgpack

    data = open(so_path, "rb").read()
    sec = _section(data, b".hip_fatbin")
    if sec is None:
        return {}
    fat = data[sec[0]:sec[0] + sec[1]]
    rows = {}
    for m in re.finditer(re.escape(_MAGIC), fat):
        o = m.start()
        num, = struct.unpack_from("<Q", fat, o + 24)
        p = o + 32
        for _ in range(num):
            eo, es, tl = struct.unpack_from("<QQQ", fat, p)
            p += 24
            triple = fat[p:p + tl]
            p += tl
            if b"gfx950" not in triple or es == 0:
                continue
            elf = fat[o + eo:o + eo + es]
            for name, ntype, desc in _notes(elf):
                if name != b"AMDGPU" or ntype != 32:
                    continue
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    rows[k[".name"]] = {"vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count", 0),
                                        "sgpr": k.get(".sgpr_count"), "scratch": k.get(".private_segment_fixed_size"),
                                        "lds": k.get(".group_segment_fixed_size")}
    names = list(rows)
    return {pretty: rows[n] for n, pretty in zip(names, _demangle(names))}


if __name__ == "__main__":
    import os
    import sys

    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfhe_hip.so")
    res = kernel_resources(so)
    code = kernel_code_hashes(so)
    for k in sorted(res):
        if len(sys.argv) <= 2 or any(s in k for s in sys.argv[2:]):
            print(f"{k:80s} {res[k]} code {code.get(k)}")
