#!/usr/bin/env python3
"""Generates primus-fhe_amd/csrc/pfhe_mont_asm.hpp: NTT butterflies for odd primes 2^48 <= q < 2^61 (any odd q < 2^61 for the butterflies themselves) as hand-scheduled gfx950
instruction sequences (inline asm with fixed temporaries, same conventions as tools/gen_pm_asm.py).

Arithmetic (MontArith, pfhe_ntt_device.hpp): one-word Montgomery reduction with a split multiplicand.  The table holds
each twiddle as {wm = w * 2^32 mod q, wm2 = w * 2^64 mod q}; for the 32-bit halves y0, y1 of ANY 64-bit y

    S = y0 * wm + y1 * wm2            ==  y * w * 2^32  (mod q),   S < 2^33 * q
    m = (S mod 2^32) * (-q^-1 mod 2^32) mod 2^32
    T = (S + m * q) / 2^32            ==  y * w         (mod q),   T < 3q

seven 32 x 32 multiplies (six v_mad_u64_u32 + one v_mul_lo_u32) against the ten of the reference's Shoup product
(w * y - floor(w' * y / 2^64) * q, primus_factor/src/shoup_factor/mod.rs:124-131); canonical results are the same residues.
Column 0 (y0*wm0 + y1*wm2_0 + m*q0, low word zero by construction) can carry twice; both carries join the high word of the
column-1 addend.  Columns 1 (three products below 2^61 + the carry word) cannot overflow for q < 2^61.

Forward lazy domain: values below 2^63 + 3q < 2^64.  The fold of x tests ONE bit instead of comparing: with F = c*q the
largest multiple of q that is <= 2^63 (c >= 4), X = x - F when bit 63 of x is set (x - F < 3q + 2^63 - F < 4q),
X = x otherwise, so X < 2^63 either way; x' = X + T < 2^63 + 3q, y' = X + 3q - T < 2^63 + 3q.  As instructions
(MontArith::fold, in front of the asm block so that the compiler places it): v_ashrrev_i32 by 31 (the mask), two
v_and_b32 with the halves of 2^64 - F, one v_lshl_add_u64 — 11.8 issue cycles where compare-and-select (v_sub_co,
v_subb_co, 2 v_cndmask) took 17.4.  FOLD = false leaves the fold out: the first stage of a transform, whose inputs are
below 4q by the reference's contract.  The closing reduction of a transform (MontArith::canon) takes any value of the domain.
Inverse (inputs below F): x' = (x + y) - F if that is >= F (< F), y' = (x + F - y) * w < 3q <= F.

    forward : 4 (sign-bit fold) + 10 (product) + 4 = 18 instructions   (Shoup form, compiled: ~30)
    inverse : 1 + 3 + 4 + 10                       = 18 instructions

Run from the repository root:  python tools/gen_mont_asm.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "primus-fhe_amd", "csrc",
                   "pfhe_mont_asm.hpp")

# temporaries: pairs A, B, E, C (even-aligned) and the single register M (the Montgomery factor m), per interleaved butterfly
SETS = [dict(A=(2, 3), B=(4, 5), E=(6, 7), C=(8, 9), M=18, cy="vcc"),
        dict(A=(10, 11), B=(12, 13), E=(14, 15), C=(16, 17), M=19, cy="%[cyb]")]


def pair(p):
    return f"v[{p[0]}:{p[1]}]"


def mul_seq(t, s, y0, y1, out=None):
    """T = y * w mod~ q (< 3q) for the value with halves y0, y1; result in pair A or `out`; uses A, B, E, M."""
    A, B, E, M, cy = t["A"], t["B"], t["E"], t["M"], t["cy"]
    dst = out if out else pair(A)
    return [
        f"v_mad_u64_u32 {pair(A)}, {cy}, {y0}, %[w0{s}], 0",
        f"v_mad_u64_u32 {pair(A)}, {cy}, {y1}, %[v0{s}], {pair(A)}",       # column 0, first carry
        f"v_addc_co_u32_e64 v{E[1]}, {cy}, 0, 0, {cy}",
        f"v_mul_lo_u32 v{M}, v{A[0]}, %[qinv]",                             # m
        f"v_mad_u64_u32 {pair(A)}, {cy}, v{M}, %[q0], {pair(A)}",           # + m*q0: low word 0, second carry
        f"v_mov_b32 v{E[0]}, v{A[1]}",
        f"v_addc_co_u32_e64 v{E[1]}, {cy}, v{E[1]}, 0, {cy}",               # E = {column 0 >> 32, carries}
        f"v_mad_u64_u32 {pair(B)}, {cy}, {y0}, %[w1{s}], {pair(E)}",
        f"v_mad_u64_u32 {pair(B)}, {cy}, {y1}, %[v1{s}], {pair(B)}",
        f"v_mad_u64_u32 {dst}, {cy}, v{M}, %[q1], {pair(B)}",               # T = (S + m*q) >> 32
    ]


def fwd_seq(t, s):
    A, B, C, cy = t["A"], t["B"], t["C"], t["cy"]
    seq = mul_seq(t, s, f"%[y0{s}]", f"%[y1{s}]")                          # X = fold(x) arrives as an operand (A::fold)
    seq += [
        f"v_lshl_add_u64 %[xo{s}], %[x{s}], 0, {pair(A)}",                  # x' = X + T
        f"v_lshl_add_u64 {pair(B)}, %[x{s}], 0, %[q3]",                     # X + 3q
        f"v_sub_co_u32_e64 %[yo0{s}], {cy}, v{B[0]}, v{A[0]}",
        f"v_subb_co_u32_e64 %[yo1{s}], {cy}, v{B[1]}, v{A[1]}, {cy}",       # y' = X + 3q - T
    ]
    return seq


def inv_seq(t, s):
    A, B, C, cy = t["A"], t["B"], t["C"], t["cy"]
    seq = [
        f"v_lshl_add_u64 {pair(A)}, %[x{s}], 0, %[y{s}]",                   # A = x + y  (< 2F)
        f"v_lshl_add_u64 {pair(C)}, %[x{s}], 0, %[q4]",                     # C = x + F
        f"v_sub_co_u32_e64 v{C[0]}, {cy}, v{C[0]}, %[y0{s}]",
        f"v_subb_co_u32_e64 v{C[1]}, {cy}, v{C[1]}, %[y1{s}], {cy}",        # C = x + F - y
        f"v_add_co_u32_e64 v{B[0]}, {cy}, v{A[0]}, %[nqf0]",
        f"v_addc_co_u32_e64 v{B[1]}, {cy}, v{A[1]}, %[nqf1], {cy}",         # + (2^64 - F): carries iff x + y >= F
        f"v_cndmask_b32_e64 %[xo0{s}], v{A[0]}, v{B[0]}, {cy}",
        f"v_cndmask_b32_e64 %[xo1{s}], v{A[1]}, v{B[1]}, {cy}",            # x' = (x + y) - F when that is >= 0
    ]
    seq += mul_seq(t, s, f"v{C[0]}", f"v{C[1]}", out=f"%[yo{s}]")          # y' = (x + F - y) * w
    return seq


def interleave(seqs):
    out = []
    for i in range(max(len(q) for q in seqs)):
        for q in seqs:
            if i < len(q):
                out.append(q[i])
    return out


def emit_asm(lines, outs, ins, clobbers, indent):
    body = "\n".join(f'{indent}    "{l}\\n\\t"' for l in lines[:-1]) + f'\n{indent}    "{lines[-1]}"'
    return (f"{indent}asm(\n{body}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins)}\n"
            f"{indent}    : {', '.join(chr(34) + c + chr(34) for c in clobbers)});\n")


def clobbers_of(sets, keys):
    regs = []
    for t in sets:
        for k in keys:
            regs += [f"v{t[k][0]}", f"v{t[k][1]}"] if k != "M" else [f"v{t[k]}"]
    return ["vcc"] + regs


def tw_ins(s, tc):
    return [f'[w0{s}] "{tc}"((u32)w{s}.w)', f'[w1{s}] "{tc}"((u32)(w{s}.w >> 32))',
            f'[v0{s}] "{tc}"((u32)w{s}.w2)', f'[v1{s}] "{tc}"((u32)(w{s}.w2 >> 32))']


CONST_MUL = ['[qinv] "s"(ar.qinv)', '[q0] "s"((u32)ar.q)', '[q1] "s"((u32)(ar.q >> 32))']


def gen_fwd(ways, uni):
    sfx, sets = ["a", "b"][:ways], SETS[:ways]
    lines = interleave([fwd_seq(t, s) for t, s in zip(sets, sfx)])
    outs, ins = [], []
    for s in sfx:
        outs += [f'[xo{s}] "=&v"(xo{s})', f'[yo0{s}] "=&v"(yo0{s})', f'[yo1{s}] "=&v"(yo1{s})']
    if ways == 2:
        outs += ['[cyb] "=&s"(cyb)']
    for s in sfx:
        ins += [f'[x{s}] "v"(X{s})', f'[y0{s}] "v"((u32)y{s})',
                f'[y1{s}] "v"((u32)(y{s} >> 32))'] + tw_ins(s, "s" if uni else "v")
    ins += CONST_MUL + ['[q3] "s"(ar.q3)']
    return emit_asm(lines, outs, ins, clobbers_of(sets, ["A", "B", "E", "M"]), "        ")


def gen_inv(ways, uni):
    sfx, sets = ["a", "b"][:ways], SETS[:ways]
    lines = interleave([inv_seq(t, s) for t, s in zip(sets, sfx)])
    outs, ins = [], []
    for s in sfx:
        outs += [f'[xo0{s}] "=&v"(xo0{s})', f'[xo1{s}] "=&v"(xo1{s})', f'[yo{s}] "=&v"(yo{s})']
    if ways == 2:
        outs += ['[cyb] "=&s"(cyb)']
    for s in sfx:
        ins += [f'[x{s}] "v"(x{s})', f'[y{s}] "v"(y{s})', f'[y0{s}] "v"((u32)y{s})', f'[y1{s}] "v"((u32)(y{s} >> 32))']
        ins += tw_ins(s, "s" if uni else "v")
    ins += CONST_MUL + ['[q4] "s"(ar.qf)', '[nqf0] "v"(ar.vnqf_0)', '[nqf1] "v"(ar.vnqf_1)']
    return emit_asm(lines, outs, ins, clobbers_of(sets, ["A", "B", "E", "C", "M"]), "        ")


def gen_mul(uni):
    t = SETS[0]
    lines = mul_seq(t, "a", "%[y0a]", "%[y1a]", out="%[o]")
    outs = ['[o] "=&v"(o)']
    ins = ['[y0a] "v"((u32)y)', '[y1a] "v"((u32)(y >> 32))'] + tw_ins("a", "s" if uni else "v") + CONST_MUL
    src = emit_asm(lines, outs, ins, clobbers_of([t], ["A", "B", "E", "M"]), "        ")
    return src.replace("wa.", "w.")


HEADER = '''// pfhe_mont_asm.hpp — GENERATED by tools/gen_mont_asm.py; do not edit by hand.
//
// NTT butterflies for odd primes 2^48 <= q < 2^61 (MontArith, pfhe_ntt_device.hpp; the butterflies hold for any odd q < 2^61, the closing quotient-estimate reduction needs q >= 2^48): twiddles {w*2^32 mod q, w*2^64 mod q},
// T = (y0*wm + y1*wm2 + m*q) / 2^32 < 3q with m = (low word) * (-q^-1) mod 2^32 — seven 32 x 32 multiplies; forward
// X = x - F if bit 63 of x is set (F = the largest multiple of q below 2^63), x' = X + T, y' = X + 3q - T (below 2^63 + 3q);
// inverse x' = x + y - F if that is >= F, y' = (x + F - y) * w.
// Fixed temporaries v2..v19.  `A` provides q, q3 = 3q, qf = F, qinv = -q^-1 mod 2^32, fold() and, in VGPRs, the halves of
// 2^64 - F (vnqf_0, vnqf_1).  UNI: the twiddle is wave-uniform and sits in SGPRs.
#pragma once

namespace pfhe {

'''


def main():
    src = HEADER
    for ways in (1, 2):
        sfx = ["a", "b"][:ways]
        args = ", ".join(f"u64 &x{s}, u64 &y{s}, TW w{s}" for s in sfx)
        src += f"template <bool UNI, bool FOLD = true, class A, class TW>\n__device__ __forceinline__ void mont_fwd_bfly{ways}(const A &ar, {args}) {{\n"
        src += "    u64 " + ", ".join(f"xo{s}" for s in sfx) + ";\n    u32 " + ", ".join(f"yo0{s}, yo1{s}" for s in sfx) + ";\n"
        src += "    const u64 " + ", ".join(f"X{s} = FOLD ? ar.fold(x{s}) : x{s}" for s in sfx) + ";\n"
        if ways == 2:
            src += "    u64 cyb;\n"
        src += "    if constexpr (UNI) {\n" + gen_fwd(ways, True) + "    } else {\n" + gen_fwd(ways, False) + "    }\n"
        for s in sfx:
            src += f"    x{s} = xo{s};\n    y{s} = ((u64)yo1{s} << 32) | yo0{s};\n"
        src += "}\n\n"
    for ways in (1, 2):
        sfx = ["a", "b"][:ways]
        args = ", ".join(f"u64 &x{s}, u64 &y{s}, TW w{s}" for s in sfx)
        src += f"template <bool UNI, class A, class TW>\n__device__ __forceinline__ void mont_inv_bfly{ways}(const A &ar, {args}) {{\n"
        src += "    u64 " + ", ".join(f"yo{s}" for s in sfx) + ";\n    u32 " + ", ".join(f"xo0{s}, xo1{s}" for s in sfx) + ";\n"
        if ways == 2:
            src += "    u64 cyb;\n"
        src += "    if constexpr (UNI) {\n" + gen_inv(ways, True) + "    } else {\n" + gen_inv(ways, False) + "    }\n"
        for s in sfx:
            src += f"    x{s} = ((u64)xo1{s} << 32) | xo0{s};\n    y{s} = yo{s};\n"
        src += "}\n\n"
    src += ("// y * w mod~ q in [0, 3q) for any 64-bit y\ntemplate <bool UNI, class A, class TW>\n"
            "__device__ __forceinline__ u64 mont_mul1(const A &ar, u64 y, TW w) {\n    u64 o;\n"
            "    if constexpr (UNI) {\n" + gen_mul(True) + "    } else {\n" + gen_mul(False) + "    }\n    return o;\n}\n\n")
    src += "}  // namespace pfhe\n"
    with open(OUT, "w") as f:
        f.write(src)
    print("wrote", OUT, len(src.splitlines()), "lines")


if __name__ == "__main__":
    main()
