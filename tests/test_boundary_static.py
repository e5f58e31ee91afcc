"""Static guard of the drop-in boundary (no GPU, no Rust toolchain): the three descriptions of the C ABI —
include/pfhe.h, the Rust declarations of integration/primus_ntt_hip/src/ffi.rs and the ctypes table of
primus-fhe_amd/_lib.py — must name the same functions with the same arity and the same argument / result widths, the
built library must export exactly those symbols, and the Rust shim must implement every method the reference's traits
require (tests/golden/trait_methods.json: primus_ntt/src/ntt/mod.rs:16-113, dcrt/mod.rs:19-135).

The shim is uncompiled in this image; without this test a prototype edited in pfhe.h would turn into undefined behaviour
for its first user.  Each file is parsed by code of this test (tools/gen_ffi_rs.py writes ffi.rs but is not used here).
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pfhe.h")
FFI = os.path.join(ROOT, "integration", "primus_ntt_hip", "src", "ffi.rs")
LIBRS = os.path.join(ROOT, "integration", "primus_ntt_hip", "src", "lib.rs")

# canonical width classes: i32 u32 u64 u8 usize f64 void, and "ptr" (every pointer is one machine word)
C_SCALAR = {"int": "i32", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "double": "f64",
            "void": "void", "char": "i8"}
RUST_SCALAR = {"c_int": "i32", "usize": "usize", "u64": "u64", "u32": "u32", "u8": "u8", "f64": "f64", "c_char": "i8"}


def c_class(t: str):
    t = t.replace("const", " ").strip()
    if "*" in t:
        base = t.replace("*", " ").split()[0]
        return "ptr:" + C_SCALAR.get(base, "handle") + ("*" * (t.count("*") - 1))
    return C_SCALAR[t.split()[0]]


def header_prototypes(text: str):
    body = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.sub(r"^\s*#.*$", "", body, flags=re.M).replace('extern "C" {', "")
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pfhe_\w+)\s*\(([^;{}]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                ty = re.match(r"(.*?)(\w+)$", a.strip()).group(1)
                params.append(c_class(ty))
        assert name not in out, f"{name} declared twice in pfhe.h"
        out[name] = (c_class(ret), params)
    return out


def rust_class(t: str):
    t = t.strip()
    if t.startswith("*"):
        depth = t.count("*")
        base = re.sub(r"\*(const|mut)\s+", "", t).strip()
        return "ptr:" + {"c_void": "void"}.get(base, RUST_SCALAR.get(base, "handle")) + ("*" * (depth - 1))
    return RUST_SCALAR[t]


def rust_prototypes(text: str):
    out = {}
    for m in re.finditer(r"pub fn (pfhe_\w+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", text, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(4)
        params = [rust_class(a.split(":", 1)[1]) for a in args.split(",") if ":" in a]
        assert name not in out, f"{name} declared twice in ffi.rs"
        out[name] = (rust_class(ret) if ret else "void", params)
    return out


def ctypes_class(t):
    if t is None:
        return "void"
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return "ptr"
    return {C.c_int: "i32", C.c_size_t: "usize", C.c_uint64: "u64", C.c_uint32: "u32", C.c_uint8: "u8", C.c_double: "f64"}[t]


def ctypes_prototypes():
    """What primus-fhe_amd/_lib.py declares: _declare() run against a recorder instead of the loaded library."""
    sys.path.insert(0, ROOT)
    from primus_fhe_amd import _lib

    class Fn:
        restype, argtypes = "unset", None

    class Recorder:
        def __init__(self):
            self.fns = {}

        def __getattr__(self, name):
            if name.startswith("pfhe_"):
                return self.fns.setdefault(name, Fn())
            raise AttributeError(name)

    rec = Recorder()
    _lib._declare(rec)
    return {n: (ctypes_class(f.restype), [ctypes_class(a) for a in f.argtypes]) for n, f in rec.fns.items()}


def width(cls: str) -> str:
    """pointer classes compare as pointers (ctypes does not say what a void pointer points at); on LP64 ctypes' c_size_t
    and c_uint64 are one and the same class, so the two compare as one 64-bit width here (ffi.rs keeps them apart)"""
    return "ptr" if cls.startswith("ptr") else {"usize": "u64"}.get(cls, cls)


@pytest.fixture(scope="module")
def header():
    return header_prototypes(open(HEADER).read())


def test_header_parses_completely(header):
    text = open(HEADER).read()
    # every `pfhe_xxx(` that opens a declaration in the header was understood by the parser above
    declared = set(re.findall(r"\b(pfhe_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert declared == set(header), sorted(declared ^ set(header))
    assert len(header) >= 219


def test_ffi_rs_declares_every_prototype_with_the_same_widths(header):
    rust = rust_prototypes(open(FFI).read())
    assert set(rust) == set(header), {"only in pfhe.h": sorted(set(header) - set(rust)), "only in ffi.rs": sorted(set(rust) - set(header))}
    for name, (ret, params) in header.items():
        rret, rparams = rust[name]
        assert len(params) == len(rparams), (name, params, rparams)
        assert ret == rret, (name, ret, rret)
        assert params == rparams, (name, params, rparams)     # incl. what every pointer points at (u32 / u64 / handle / void)
    # status codes
    ctext, rtext = open(HEADER).read(), open(FFI).read()
    assert dict(re.findall(r"(PFHE_\w+)\s*=\s*(\d+)", ctext)) == dict(re.findall(r"pub const (PFHE_\w+): c_int = (\d+);", rtext))
    # opaque handle types
    assert set(re.findall(r"typedef struct (pfhe_\w+) \1;", ctext)) == \
        set(re.search(r"opaque!\((.*?)\);", rtext).group(1).replace(" ", "").split(","))


def test_generated_ffi_rs_is_current():
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ffi_rs.py"), "--check"], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stdout + rc.stderr


def test_ctypes_table_matches_the_header(header):
    py = ctypes_prototypes()
    assert set(py) == set(header), {"only in pfhe.h": sorted(set(header) - set(py)), "only in _lib.py": sorted(set(py) - set(header))}
    for name, (ret, params) in header.items():
        pret, pparams = py[name]
        assert len(params) == len(pparams), (name, params, pparams)
        assert width(ret) == width(pret), (name, ret, pret)
        assert [width(p) for p in params] == [width(p) for p in pparams], (name, params, pparams)


def test_adding_an_argument_to_a_prototype_is_caught(header):
    """The acceptance test of VERDICT r5 item 3: edit one prototype of pfhe.h (in memory) and the comparison fails."""
    text = open(HEADER).read()
    edited = text.replace("int pfhe_rns_create(const uint64_t *moduli, size_t count, int device, pfhe_rns **out);",
                          "int pfhe_rns_create(const uint64_t *moduli, size_t count, int device, int flags, pfhe_rns **out);")
    assert edited != text
    h2 = header_prototypes(edited)
    rust = rust_prototypes(open(FFI).read())
    assert h2["pfhe_rns_create"][1] != rust["pfhe_rns_create"][1]
    narrowed = text.replace("uint64_t pfhe_ntt_modulus(const pfhe_ntt *table);", "uint32_t pfhe_ntt_modulus(const pfhe_ntt *table);")
    assert narrowed != text and header_prototypes(narrowed)["pfhe_ntt_modulus"][0] != rust["pfhe_ntt_modulus"][0]


def test_library_exports_exactly_the_declared_symbols(header):
    so = os.path.join(ROOT, "primus-fhe_amd", "libpfhe_hip.so")
    if not os.path.exists(so):
        pytest.skip("libpfhe_hip.so not built")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("pfhe_")}
    assert exported == set(header), {"not exported": sorted(set(header) - exported), "not declared": sorted(exported - set(header))}


def impl_methods(text: str, trait: str):
    """{implementing type: set of fn names} for every `impl <trait> for X { ... }` block of lib.rs"""
    out = {}
    for m in re.finditer(r"impl\s+" + trait + r"\s+for\s+(\w+)\s*\{", text):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        out[m.group(1)] = set(re.findall(r"\bfn\s+(\w+)", text[m.end():i]))
    return out


def test_rust_shim_implements_every_required_trait_method():
    traits = json.load(open(os.path.join(ROOT, "tests", "golden", "trait_methods.json")))
    text = open(LIBRS).read()
    ntt, dcrt = impl_methods(text, "NttTable"), impl_methods(text, "DcrtTable")
    assert {"HipNttTable", "HipU32NttTable"} <= set(ntt) and {"HipDcrtTable", "HipU32DcrtTable"} <= set(dcrt)
    for ty, fns in ntt.items():
        assert set(traits["NttTable"]["required"]) <= fns, (ty, sorted(set(traits["NttTable"]["required"]) - fns))
        assert fns <= set(traits["NttTable"]["required"]) | set(traits["NttTable"]["provided"]), (ty, fns)
    for ty, fns in dcrt.items():
        assert set(traits["DcrtTable"]["required"]) <= fns, (ty, sorted(set(traits["DcrtTable"]["required"]) - fns))
        assert fns <= set(traits["DcrtTable"]["required"]) | set(traits["DcrtTable"]["provided"]), (ty, fns)
    # every C function the shim calls exists in the header
    called = set(re.findall(r"\b(?:ffi::)?(pfhe_\w+)\s*\(", text))
    declared = set(header_prototypes(open(HEADER).read()))
    assert called <= declared, sorted(called - declared)


@pytest.mark.skipif(not os.path.isdir("/root/reference/crates/primus_ntt"), reason="the reference tree is not on this machine")
def test_trait_method_list_matches_the_reference():
    """Where the reference is present (the build container, not the GPU box) the committed list is checked against its
    trait definitions."""
    traits = json.load(open(os.path.join(ROOT, "tests", "golden", "trait_methods.json")))
    for trait, path in (("NttTable", "ntt/mod.rs"), ("DcrtTable", "dcrt/mod.rs")):
        src = open(os.path.join("/root/reference/crates/primus_ntt/src", path)).read()
        start = src.index("pub trait " + trait)
        depth, i = 0, src.index("{", start)
        body_start = i + 1
        while True:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
            if depth == 0:
                break
        body = src[body_start:i]
        req, prov = [], []
        for m in re.finditer(r"\bfn\s+(\w+)[^;{]*?([;{])", body, flags=re.S):
            (req if m.group(2) == ";" else prov).append(m.group(1))
        assert req == traits[trait]["required"] and prov == traits[trait]["provided"], (trait, req, prov)
