#!/usr/bin/env python3
"""Disassembly of the gfx950 kernels of a built libpfhe_hip.so whose (demangled) name contains PATTERN, with static
instruction counts — the ISA evidence the experiment logs quote.

    python tools/disasm_kernel.py PATTERN [--lib path/to/libpfhe_hip.so] [--out DIR] [--counts-only]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "primus-fhe_amd"))
import _codeobj  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern")
    ap.add_argument("--lib", default=os.path.join(ROOT, "primus-fhe_amd", "libpfhe_hip.so"))
    ap.add_argument("--out", default="")
    ap.add_argument("--counts-only", action="store_true")
    args = ap.parse_args()
    for idx, elf in enumerate(_codeobj._gfx950_elfs(args.lib)):
        names = [n for n, _ in _codeobj._func_symbols(elf)]
        pretty = dict(zip(names, _codeobj._demangle(names)))
        want = [n for n in names if args.pattern in pretty[n]]
        if not want:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
            path = f.name
        try:
            for n in want:
                txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + n, path],
                                     capture_output=True, text=True, check=True).stdout
                ins = [l.split()[0] for l in txt.splitlines() if re.match(r"^\s+[a-z_0-9]+\s", l) or re.match(r"^\s+s_endpgm", l)]
                c = collections.Counter(ins)
                valu = sum(v for k, v in c.items() if k.startswith("v_"))
                lds = sum(v for k, v in c.items() if k.startswith("ds_"))
                vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
                print(f"{pretty[n]}: {len(ins)} instructions, VALU {valu} (v_mad_u64_u32 {c['v_mad_u64_u32']}, v_lshl_add_u64 "
                      f"{c['v_lshl_add_u64']}, v_mov_b32 {c['v_mov_b32'] + c['v_mov_b32_e32'] + c['v_mov_b32_e64']}), DS {lds}, VMEM {vmem} (global_load_lds "
                      f"{sum(v for k, v in c.items() if k.startswith('global_load_lds'))}), s_barrier {c['s_barrier']}, "
                      f"s_waitcnt {c['s_waitcnt']}")
                if args.out:
                    os.makedirs(args.out, exist_ok=True)
                    open(os.path.join(args.out, re.sub(r"[^A-Za-z0-9_]+", "_", pretty[n]) + ".s"), "w").write(txt)
                elif not args.counts_only:
                    print(txt)
        finally:
            os.unlink(path)


if __name__ == "__main__":
    main()
