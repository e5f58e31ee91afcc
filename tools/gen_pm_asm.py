#!/usr/bin/env python3
"""Generates primus-fhe_amd/csrc/pfhe_pm_asm.hpp: the pseudo-Mersenne butterflies of the NTT kernels as hand-scheduled
gfx950 instruction sequences (inline asm with fixed temporary registers).

Why generated asm and not C++: the butterfly is a chain of v_mad_u64_u32 whose 64-bit addends are built from HALVES of
earlier results ({t1.hi, carry}, {t1.lo, t3.lo & mask}, ...).  LLVM inline asm cannot name a half of a 64-bit operand,
and left to itself the compiler re-associates the sums and copies halves around (22 -> 29 instructions per butterfly,
tools/microbench6.hip).  With the temporaries in fixed physical registers every half has a name; values that cross the
block boundary are passed BOTH as a 64-bit operand and as two 32-bit operands (same registers, no copies) and are
written only by instructions with a 64-bit destination.

Cost model (tools/microbench5.hip, MI355X, 4 waves per SIMD, cycles per wave-instruction): v_add/sub/and/or/xor/not/
mov/lshrrev with VGPR or inline operands 2.3; everything else 4.2 (v_mad_u64_u32, v_lshl_add_u64, v_alignbit_b32,
shifts left, min/max, any 32-bit op with an SGPR operand), v_add_co / v_addc_co 4.4.

    forward, no fold : 5 mad + mov + addc + alignbit + and + 2 lshl_add_u64 + sub_co + subb  = 13 instr, 51 cycles
    forward, fold    : + lshrrev + and + mov + mad                                           = 17 instr, 62 cycles
                       (round 5: lshrrev + and placed by the compiler in front of the block, no mov  = 16 instr, 60 cycles)
    (until late in round 2 the tail was not, not, 2 x lshl_add_u64: 14 / 18 instructions for the same cycles — but the
    kernels run at the package power cap, where an instruction less is worth more than its issue slot)
    inverse          : lshl_add_u64 + (lshrrev, and, mad) + lshl_add_u64 + sub_co + subb + multiply (10) = 18 instr

Run from the repository root:  python tools/gen_pm_asm.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "primus-fhe_amd", "csrc",
                   "pfhe_pm_asm.hpp")

# temporary register sets (pairs must start at an even register): A, B, E, C
# cy: where the set keeps carries (every v_mad_u64_u32 writes a carry-out, wanted or not): interleaved butterflies
# must not share it
# LOW register numbers: a kernel's register count is its highest register + 1, and v2..v17 are always in use anyway
# (the work-item ids arrive in v0..v2 and are consumed before the first butterfly)
SETS = [dict(A=(2, 3), B=(4, 5), E=(6, 7), C=(8, 9), cy="vcc"),
        dict(A=(10, 11), B=(12, 13), E=(14, 15), C=(16, 17), cy="%[cyb]")]
if os.environ.get("PM_ASM_HIGH_TEMPS"):  # experiment: the first placement tried (forces 128 registers on every kernel)
    SETS = [dict(A=(120, 121), B=(122, 123), E=(124, 125), C=(126, 127), cy="vcc"),
            dict(A=(112, 113), B=(114, 115), E=(116, 117), C=(118, 119), cy="%[cyb]")]


# forward y' = X + 3q - T as a 64-bit borrow chain into two 32-bit outputs (3 instructions) instead of not, not, two 64-bit
# adds (4): one instruction fewer per butterfly (1846 -> 1742 in the block pass); PM_ASM_NOT_ADD=1 restores the old tail
SUBC = not os.environ.get("PM_ASM_NOT_ADD")
# forward fold: shift and mask of x as C++ in front of the asm block (round 5; PM_ASM_FOLD_INSIDE=1 restores the 17-instruction
# form with the three instructions and the copy inside the block)
FOLD_OUTSIDE = not os.environ.get("PM_ASM_FOLD_INSIDE")
CANON_OUTSIDE = bool(os.environ.get("PM_ASM_CANON_OUTSIDE"))  # experiment only


def pair(p):
    return f"v[{p[0]}:{p[1]}]"


def mul_seq(t, s, y0, y1, out=None):
    """twiddle product of the 64-bit value whose halves are y0, y1 (register names or operand refs): result T in
    pair A (or written to `out`); uses A, B, E.  Operand names carry the suffix s (one per interleaved butterfly)."""
    A, B, E, cy = t["A"], t["B"], t["E"], t["cy"]
    dst = out if out else pair(A)
    return [
        f"v_mad_u64_u32 {pair(A)}, {cy}, {y0}, %[w0{s}], 0",
        f"v_mad_u64_u32 {pair(A)}, {cy}, {y1}, %[v0{s}], {pair(A)}",      # carry-out of column 0
        f"v_mov_b32 v{E[0]}, v{A[1]}",
        f"v_addc_co_u32_e64 v{E[1]}, {cy}, 0, 0, {cy}",                    # E = {t1.hi, carry}
        f"v_mad_u64_u32 {pair(B)}, {cy}, {y0}, %[w1{s}], {pair(E)}",
        f"v_mad_u64_u32 {pair(B)}, {cy}, {y1}, %[v1{s}], {pair(B)}",      # B = S >> 32
        f"v_alignbit_b32 v{E[0]}, v{B[1]}, v{B[0]}, %[sh1]",               # hp = S >> (K+1)
        f"v_and_b32 v{A[1]}, %[m1], v{B[0]}",                              # A = S mod 2^(K+1)
        f"v_mad_u64_u32 {dst}, {cy}, v{E[0]}, %[c2], {pair(A)}",           # T = hp*2c + L
    ]


def fwd_seq(t, s, fold, outside=False):
    A, B, E, C, cy = t["A"], t["B"], t["E"], t["C"], t["cy"]
    seq = []
    if fold and outside:
        # the shift and the mask are C++ in front of the block (x is dead afterwards: the compiler masks its high half in
        # place), so the copy of the low half into the addend pair is gone: 16 instructions
        seq += [f"v_mad_u64_u32 {pair(C)}, {cy}, %[e{s}], %[c], %[xm{s}]"]  # X = (x >> K)*c + (x mod 2^K)
        X = pair(C)
    elif fold:
        seq += [
            f"v_lshrrev_b32 v{E[0]}, %[sh], %[x1{s}]",
            f"v_and_b32 v{C[1]}, %[m], %[x1{s}]",
            f"v_mov_b32 v{C[0]}, %[x0{s}]",
            f"v_mad_u64_u32 {pair(C)}, {cy}, v{E[0]}, %[c], {pair(C)}",   # X = (x >> K)*c + (x mod 2^K)
        ]
        X = pair(C)
    else:
        X = f"%[x{s}]"
    seq += mul_seq(t, s, f"%[y0{s}]", f"%[y1{s}]")
    if SUBC:
        seq += [
            f"v_lshl_add_u64 %[xo{s}], {X}, 0, {pair(A)}",                 # x' = X + T
            f"v_lshl_add_u64 {pair(B)}, {X}, 0, %[q3]",                    # X + 3q
            f"v_sub_co_u32_e64 %[yo0{s}], {cy}, v{B[0]}, v{A[0]}",
            f"v_subb_co_u32_e64 %[yo1{s}], {cy}, v{B[1]}, v{A[1]}, {cy}",  # y' = X + 3q - T, as two halves
        ]
        return seq
    seq += [
        f"v_lshl_add_u64 %[xo{s}], {X}, 0, {pair(A)}",                     # x' = X + T
        f"v_not_b32 v{A[0]}, v{A[0]}",
        f"v_not_b32 v{A[1]}, v{A[1]}",
        f"v_lshl_add_u64 {pair(B)}, {X}, 0, %[q3p1]",                      # X + 3q + 1
        f"v_lshl_add_u64 %[yo{s}], {pair(B)}, 0, {pair(A)}",               # y' = X + 3q - T
    ]
    return seq


def inv_seq(t, s):
    A, B, E, C, cy = t["A"], t["B"], t["E"], t["C"], t["cy"]
    seq = [
        f"v_lshl_add_u64 {pair(A)}, %[x{s}], 0, %[y{s}]",                  # A = x + y
        f"v_lshl_add_u64 {pair(C)}, %[x{s}], 0, %[q3]",                    # C = x + 3q
        f"v_lshrrev_b32 v{E[0]}, %[sh], v{A[1]}",
        f"v_and_b32 v{A[1]}, %[m], v{A[1]}",
        f"v_sub_co_u32_e64 v{C[0]}, {cy}, v{C[0]}, %[y0{s}]",
        f"v_subb_co_u32_e64 v{C[1]}, {cy}, v{C[1]}, %[y1{s}], {cy}",       # C = x + 3q - y
        f"v_mad_u64_u32 %[xo{s}], {cy}, v{E[0]}, %[c], {pair(A)}",         # x' = fold(x + y)
    ]
    seq += mul_seq(t, s, f"v{C[0]}", f"v{C[1]}", out=f"%[yo{s}]")         # y' = (x + 3q - y) * w
    return seq


def canon_seq(t, s):
    """any 64-bit x -> [0, q): fold (x mod~ q < 2^K + 2^31 < 2q), then subtract q unless that borrows"""
    A, E, C, cy = t["A"], t["E"], t["C"], t["cy"]
    if CANON_OUTSIDE:  # experiment (r05_experiments.txt item 12): shift and mask as C++ in front of the block, as in the butterflies
        return [
            f"v_mad_u64_u32 {pair(C)}, {cy}, %[e{s}], %[c], %[xm{s}]",
            f"v_lshl_add_u64 {pair(A)}, {pair(C)}, 0, %[negq]",
            f"v_cmp_gt_i32_e64 {cy}, 0, v{A[1]}",
            f"v_cndmask_b32_e64 %[o0{s}], v{A[0]}, v{C[0]}, {cy}",
            f"v_cndmask_b32_e64 %[o1{s}], v{A[1]}, v{C[1]}, {cy}",
        ]
    return [
        f"v_lshrrev_b32 v{E[0]}, %[sh], %[x1{s}]",
        f"v_and_b32 v{C[1]}, %[m], %[x1{s}]",
        f"v_mov_b32 v{C[0]}, %[x0{s}]",
        f"v_mad_u64_u32 {pair(C)}, {cy}, v{E[0]}, %[c], {pair(C)}",
        f"v_lshl_add_u64 {pair(A)}, {pair(C)}, 0, %[negq]",              # A = X - q (negative iff X < q: X < 2q < 2^63)
        f"v_cmp_gt_i32_e64 {cy}, 0, v{A[1]}",
        f"v_cndmask_b32_e64 %[o0{s}], v{A[0]}, v{C[0]}, {cy}",
        f"v_cndmask_b32_e64 %[o1{s}], v{A[1]}, v{C[1]}, {cy}",
    ]


def gen_canon(ways):
    sfx = ["a", "b"][:ways]
    sets = SETS[:ways]
    lines = interleave([canon_seq(t, s) for t, s in zip(sets, sfx)])
    outs, ins = [], []
    for s in sfx:
        outs += [f'[o0{s}] "=&v"(o0{s})', f'[o1{s}] "=&v"(o1{s})']
    if ways == 2:
        outs += ['[cyb] "=&s"(cyb)']
    pre = ""
    for s in sfx:
        if CANON_OUTSIDE:
            pre += (f"    const u32 e{s} = (u32)(x{s} >> 32) >> ar.vsh;\n"
                    f"    const u64 xm{s} = ((u64)((u32)(x{s} >> 32) & ar.vmask) << 32) | (u32)x{s};\n")
            ins += [f'[e{s}] "v"(e{s})', f'[xm{s}] "v"(xm{s})']
        else:
            ins += [f'[x0{s}] "v"((u32)x{s})', f'[x1{s}] "v"((u32)(x{s} >> 32))']
    if CANON_OUTSIDE:
        ins += ['[c] "s"(ar.c)', '[negq] "s"(0ull - ar.q)']
    else:
        ins += ['[sh] "v"(ar.vsh)', '[m] "v"(ar.vmask)', '[c] "s"(ar.c)', '[negq] "s"(0ull - ar.q)']
    return pre + emit_asm(lines, outs, ins, clobbers_of(sets, ["A", "E", "C"]), indent="    ")


def interleave(seqs):
    out = []
    for i in range(max(len(q) for q in seqs)):
        for q in seqs:
            if i < len(q):
                out.append(q[i])
    return out


def emit_asm(lines, outs, ins, clobbers, indent="    "):
    body = "\n".join(f'{indent}    "{l}\\n\\t"' for l in lines[:-1]) + f'\n{indent}    "{lines[-1]}"'
    return (f"{indent}asm(\n{body}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins)}\n"
            f"{indent}    : {', '.join(chr(34) + c + chr(34) for c in clobbers)});\n")


def clobbers_of(sets, keys):
    regs = []
    for t in sets:
        for k in keys:
            regs += [f"v{t[k][0]}", f"v{t[k][1]}"]
    return ["vcc"] + regs


def gen_fwd(ways, fold, uni, outside=False):
    tc = "s" if uni else "v"
    sfx = ["a", "b"][:ways]
    sets = SETS[:ways]
    lines = interleave([fwd_seq(t, s, fold, outside) for t, s in zip(sets, sfx)])
    outs, ins = [], []
    for s in sfx:
        if SUBC:
            outs += [f'[xo{s}] "=&v"(xo{s})', f'[yo0{s}] "=&v"(yo0{s})', f'[yo1{s}] "=&v"(yo1{s})']
        else:
            outs += [f'[xo{s}] "=&v"(xo{s})', f'[yo{s}] "=&v"(yo{s})']
    if ways == 2:
        outs += ['[cyb] "=&s"(cyb)']
    for s in sfx:
        if fold and outside:
            ins += [f'[e{s}] "v"(e{s})', f'[xm{s}] "v"(xm{s})']
        elif fold:
            ins += [f'[x0{s}] "v"((u32)x{s})', f'[x1{s}] "v"((u32)(x{s} >> 32))']
        else:
            ins += [f'[x{s}] "v"(x{s})']
        ins += [f'[y0{s}] "v"((u32)y{s})', f'[y1{s}] "v"((u32)(y{s} >> 32))']
        ins += [f'[w0{s}] "{tc}"((u32)w{s}.w)', f'[w1{s}] "{tc}"((u32)(w{s}.w >> 32))',
                f'[v0{s}] "{tc}"((u32)w{s}.w2)', f'[v1{s}] "{tc}"((u32)(w{s}.w2 >> 32))']
    ins += ['[sh1] "s"(ar.sh + 1)', '[m1] "v"(ar.vmask1)', '[c2] "s"(ar.c2)',
            '[q3] "s"(ar.q3)' if SUBC else '[q3p1] "s"(ar.q3 + 1)']
    if fold and outside:
        ins += ['[c] "s"(ar.c)']
    elif fold:
        ins += ['[sh] "v"(ar.vsh)', '[m] "v"(ar.vmask)', '[c] "s"(ar.c)']
    keys = ["A", "B", "E"] + (["C"] if fold else [])
    pre = ""
    if fold and outside:
        for s in sfx:
            pre += (f"        const u32 e{s} = (u32)(x{s} >> 32) >> ar.vsh;\n"
                    f"        const u64 xm{s} = ((u64)((u32)(x{s} >> 32) & ar.vmask) << 32) | (u32)x{s};\n")
    return pre + emit_asm(lines, outs, ins, clobbers_of(sets, keys), indent="        ")


def gen_inv(ways, uni):
    tc = "s" if uni else "v"
    sfx = ["a", "b"][:ways]
    sets = SETS[:ways]
    lines = interleave([inv_seq(t, s) for t, s in zip(sets, sfx)])
    outs, ins = [], []
    for s in sfx:
        outs += [f'[xo{s}] "=&v"(xo{s})', f'[yo{s}] "=&v"(yo{s})']
    if ways == 2:
        outs += ['[cyb] "=&s"(cyb)']
    for s in sfx:
        ins += [f'[x{s}] "v"(x{s})', f'[y{s}] "v"(y{s})', f'[y0{s}] "v"((u32)y{s})', f'[y1{s}] "v"((u32)(y{s} >> 32))']
        ins += [f'[w0{s}] "{tc}"((u32)w{s}.w)', f'[w1{s}] "{tc}"((u32)(w{s}.w >> 32))',
                f'[v0{s}] "{tc}"((u32)w{s}.w2)', f'[v1{s}] "{tc}"((u32)(w{s}.w2 >> 32))']
    ins += ['[sh1] "s"(ar.sh + 1)', '[m1] "v"(ar.vmask1)', '[c2] "s"(ar.c2)', '[q3] "s"(ar.q3)', '[sh] "v"(ar.vsh)',
            '[m] "v"(ar.vmask)', '[c] "s"(ar.c)']
    return emit_asm(lines, outs, ins, clobbers_of(sets, ["A", "B", "E", "C"]), indent="        ")


HEADER = '''// pfhe_pm_asm.hpp — GENERATED by tools/gen_pm_asm.py; do not edit by hand.
//
// Pseudo-Mersenne NTT butterflies (q = 2^K - c) as hand-scheduled gfx950 instruction sequences with fixed temporary
// registers (v2..v17), one butterfly or two interleaved ones per asm block.  See the generator for the cost
// model and for why this is not C++.  The arithmetic is PmArith's (pfhe_ntt_device.hpp): twiddle product
// T = fold(y0*w + y1*w2) <= 3q with w2 = w*2^32 mod q; forward x' = X + T, y' = X + 3q - T with X = x or fold(x);
// inverse x' = fold(x + y), y' = (x + 3q - y)*w.  `A` must provide q3, c, c2, sh = K - 32 and, held in VGPRs (an SGPR operand doubles the cost of a
// v_and_b32 or v_lshrrev_b32), vsh = K - 32, vmask = 2^(K-32)-1 and vmask1 = 2^(K-31)-1.
// UNI: the twiddle is wave-uniform and sits in SGPRs.  A::kFoldOutside: the shift and the mask of a forward fold are C++ in
// front of the block (16 instructions per butterfly instead of 17; a kernel at its register limit may prefer the other form).
#pragma once

namespace pfhe {

'''


def main():
    src = HEADER
    # forward
    for ways in (1, 2):
        args = ", ".join(f"u64 &x{s}, u64 &y{s}, TW w{s}" for s in ["a", "b"][:ways])
        src += (f"template <bool FOLD, bool UNI, class A, class TW>\n__device__ __forceinline__ void pm_fwd_bfly{ways}"
                f"(const A &ar, {args}) {{\n")
        src += "    " + ", ".join(f"u64 xo{s}, yo{s}" for s in ["a", "b"][:ways]).replace(", u64", "; u64") + ";\n"
        if SUBC:
            src += "    u32 " + ", ".join(f"yo0{s}, yo1{s}" for s in ["a", "b"][:ways]) + ";\n"
        if ways == 2:
            src += "    u64 cyb;\n"
        first = True
        for fold in (False, True):
            for uni in (False, True):
                cond = f"{'FOLD' if fold else '!FOLD'} && {'UNI' if uni else '!UNI'}"
                src += f"    {'if' if first else 'else if'} constexpr ({cond}) {{\n"
                if fold and FOLD_OUTSIDE:
                    src += ("        if constexpr (A::kFoldOutside) {\n" + gen_fwd(ways, fold, uni, True).replace("\n        ", "\n            ").replace("        ", "            ", 1)
                            + "        } else {\n" + gen_fwd(ways, fold, uni, False).replace("\n        ", "\n            ").replace("        ", "            ", 1) + "        }\n")
                else:
                    src += gen_fwd(ways, fold, uni)
                src += "    }"
                src += "\n" if False else " "
                first = False
        src = src.rstrip() + "\n"
        for s in ["a", "b"][:ways]:
            if SUBC:
                src += f"    x{s} = xo{s};\n    y{s} = ((u64)yo1{s} << 32) | yo0{s};\n    (void)yo{s};\n"
            else:
                src += f"    x{s} = xo{s};\n    y{s} = yo{s};\n"
        src += "}\n\n"
    for ways in (1, 2):
        args = ", ".join(f"u64 &x{s}, u64 &y{s}, TW w{s}" for s in ["a", "b"][:ways])
        src += (f"template <bool UNI, class A, class TW>\n__device__ __forceinline__ void pm_inv_bfly{ways}"
                f"(const A &ar, {args}) {{\n")
        src += "    " + ", ".join(f"u64 xo{s}, yo{s}" for s in ["a", "b"][:ways]).replace(", u64", "; u64") + ";\n"
        if ways == 2:
            src += "    u64 cyb;\n"
        src += "    if constexpr (UNI) {\n" + gen_inv(ways, True) + "    } else {\n" + gen_inv(ways, False) + "    }\n"
        for s in ["a", "b"][:ways]:
            src += f"    x{s} = xo{s};\n    y{s} = yo{s};\n"
        src += "}\n\n"
    for ways in (1, 2):
        args = ", ".join(f"u64 &x{s}" for s in ["a", "b"][:ways])
        src += (f"// any 64-bit value -> canonical residue in [0, q)\ntemplate <class A>\n__device__ __forceinline__ void "
                f"pm_canon{ways}(const A &ar, {args}) {{\n")
        src += "    u32 " + ", ".join(f"o0{s}, o1{s}" for s in ["a", "b"][:ways]) + ";\n"
        if ways == 2:
            src += "    u64 cyb;\n"
        src += gen_canon(ways)
        for s in ["a", "b"][:ways]:
            src += f"    x{s} = ((u64)o1{s} << 32) | o0{s};\n"
        src += "}\n\n"
    src += "}  // namespace pfhe\n"
    with open(OUT, "w") as f:
        f.write(src)
    print("wrote", OUT, len(src.splitlines()), "lines")


if __name__ == "__main__":
    main()
