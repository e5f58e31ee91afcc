#!/usr/bin/env python3
"""Static cycle model of the N = 2^16 transform kernels: instruction counts by cost class from the compiler's gfx950
assembly of csrc/pfhe_ntt.hip, priced with the per-wave-instruction issue costs measured by tools/microbench5.hip
(profiles/r02_microbench5_instruction_costs.txt, 4 waves per SIMD), against the measured kernel durations.

The three kernels are straight-line code per workgroup (no loops: every stage is unrolled), so static counts are
dynamic counts; the few instructions on not-taken paths (`valid` guards) are counted too (< 1 %).  The backward jumps
the pipelined kernel shows are the block layout of its `valid` / `has_str` guards (each executed once), not loops.

    python tools/cycle_model.py [--ms block=3.17,strided=2.18,pipe=0.563 --clocks block=2.10,strided=2.21,pipe=1.98] > profiles/r02_cycle_model.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "primus-fhe_amd", "csrc", "pfhe_ntt.hip")
ASM = "/tmp/pfhe_cycle_model.s"

# cycles per wave-instruction at 4 waves per SIMD (microbench5, W=4 rows).  Classes: "half" = v_add/sub/and/or/xor/
# not/mov/lshrrev with VGPR / inline operands (2.5); "full" = everything else on the VALU (4.2-4.3); "carry" = the
# v_add_co / v_addc_co / v_sub_co / v_subb_co family (4.4); v_cndmask reading vcc written by the previous instruction
# is far slower in isolation (22) but the kernels space it out - priced as "full".
HALF = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_mov_b64", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
CARRY = {"v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32"}
COST = {"half": 2.51, "half_sgpr": 4.28, "full": 4.28, "carry": 4.41}

KERNELS = {
    "block": ("ntt_block_kernelINS_7PmArithELi12ELb0ELb0E", "block pass, forward (12 stages on 2^12 coefficients)"),
    "strided": ("ntt_strided_kernelINS_7PmArithELi4ELi2ELb0ELb0E", "strided pass, forward (4 stages, 2 columns per thread)"),
    "pipe": ("ntt_pipe_fwd_kernelINS_7PmArithELi12E", "pipelined kernel, forward (block pass + one-column strided pass)"),
}


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def classify(line):
    t = line.split()
    op = base(t[0])
    if not op.startswith("v_"):
        return None
    if op in CARRY:
        return "carry"
    if op in HALF:
        # an SGPR (or a literal that is not an inline constant) among the sources makes it a full-rate-class op
        srcs = " ".join(t[1:]).split(",")[1:]
        if any(re.match(r"\s*(s\d|s\[|vcc|exec|0x)", x) for x in srcs):
            return "half_sgpr"
        return "half"
    return "full"


def kernel_body(txt, mangled):
    m = re.search(r"^(_ZN4pfhe\w*" + mangled + r"\w*):.*$", txt, re.M)
    if not m:
        raise SystemExit("kernel not found: " + mangled)
    end = txt.index("s_endpgm", m.end())
    last = txt.rfind("s_endpgm", m.end(), txt.index(".Lfunc_end", m.end()))
    return txt[m.end():max(end, last)].split("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="", help="measured durations, e.g. block=3.19,strided=2.16,pipe=0.397")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--pipe-launches", type=int, default=9, help="launches of the pipelined kernel per transform (tiles + 1)")
    ap.add_argument("--clocks", default="", help="sustained shader clock per kernel in GHz (tools/microbench8_clock.hip), "
                                                 "e.g. block=2.10,strided=2.21,pipe=1.98")
    args = ap.parse_args()
    measured = dict((k, float(v)) for k, v in (kv.split("=") for kv in args.ms.split(",") if kv))
    clocks = dict((k, float(v)) for k, v in (kv.split("=") for kv in args.clocks.split(",") if kv))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", ASM, SRC], stderr=subprocess.DEVNULL)
    txt = open(ASM).read()
    polys = 4096 * 3
    isa = {}
    print("Static cycle model, N = 2^16, 3 x 61-bit pseudo-Mersenne primes, 4096 RNS polynomials (12 288 limb transforms)")
    print(f"costs (cycles per wave-instruction at 4 waves/SIMD, microbench5): {COST};  {args.clock_ghz} GHz, 256 CUs x 4 SIMDs\n")
    for key, (mangled, title) in KERNELS.items():
        body = kernel_body(txt, mangled)
        cls = collections.Counter()
        ops = collections.Counter()
        other = collections.Counter()
        back = 0
        labels = {}
        for i, l in enumerate(body):
            if l.startswith(".LBB"):
                labels[l.split(":")[0]] = i
        for i, l in enumerate(body):
            ls = l.strip()
            if not ls or ls.startswith(";") or ls.startswith("."):
                continue
            c = classify(ls)
            op = base(ls.split()[0])
            if c:
                cls[c] += 1
                ops[op] += 1
            else:
                kind = ("ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_"))
                        else "smem" if op.startswith("s_load") else "barrier" if op == "s_barrier"
                        else "waitcnt" if op == "s_waitcnt" else "salu")
                other[kind] += 1
                if op.startswith("s_cbranch") or op == "s_branch":
                    tgt = ls.split()[-1]
                    if tgt in labels and labels[tgt] < i:
                        back += 1
        valu = sum(cls.values())
        cycles = sum(COST[k] * v for k, v in cls.items())
        # waves: block / pipe: 16 workgroups of 4 waves per limb-polynomial; strided: N/32 threads per limb-polynomial
        waves = polys * (16 * 4 if key != "strided" else (65536 // 32) // 64)
        if key == "pipe":
            waves = polys * 16 * 4  # (+ 1/12 for the two edge launches, ignored)
        per_simd = waves / 1024.0
        ms = per_simd * cycles / (args.clock_ghz * 1e6)
        print(f"{title}\n  VALU instructions per wave: {valu}  "
              + "  ".join(f"{k} {v}" for k, v in sorted(cls.items())) + f"   -> {cycles:.0f} issue cycles per wave")
        print("  top ops: " + ", ".join(f"{o} {c}" for o, c in ops.most_common(9)))
        print("  other:   " + ", ".join(f"{k} {v}" for k, v in sorted(other.items())) + f";  backward jumps: {back}")
        line = f"  {waves} waves = {per_simd:.0f} per SIMD -> VALU issue floor {ms:.3f} ms per {polys} limb transforms at {args.clock_ghz} GHz"
        if key in clocks:
            ms = ms * args.clock_ghz / clocks[key]
            line += f", {ms:.3f} ms at the sustained {clocks[key]} GHz"
        if key in measured:
            tot = measured[key] * (args.pipe_launches if key == "pipe" else 1)
            line += f";  measured {tot:.3f} ms -> {ms / tot:.0%} of the time is VALU issue at these costs"
        print(line + "\n")
        if key == "block":
            app = [i for i, l in enumerate(body) if "#ASMSTART" in l]
            noapp = [i for i, l in enumerate(body) if "#ASMEND" in l]
            for a0 in app:  # the first asm block that is a butterfly (the short ones are register pins)
                e0 = min(e for e in noapp if e > a0)
                if e0 - a0 > 20:
                    isa[key] = [l.rstrip() for l in body[a0:e0 + 1]]
                    break
    if "block" in isa:
        print("ISA listing: the first asm block of the block pass as compiled (two interleaved forward butterflies, uniform\n"
              "twiddles in SGPRs, WITH the fold of x = 18 instructions each (the 14-instruction form lacks the first four); generator: tools/gen_pm_asm.py; all variants: primus-fhe_amd/csrc/pfhe_pm_asm.hpp)")
        print("\n".join(isa["block"]) + "\n")
    print("HBM floor of the two-pass plan: 2 x 12.885 GB read+written; at the 6.0 TB/s this chip sustains for streaming read+write "
          "(strided pass alone: 2.16 ms per 12.885 GB) = 4.3 ms per step.")


if __name__ == "__main__":
    main()
