#!/usr/bin/env python3
"""Register / scratch report of the kernels of one HIP source, from the compiler's own remarks.

    python tools/kernel_resources.py primus-fhe_amd/csrc/pfhe_ntt.hip [substring ...] [--fail-on-scratch]

Compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage (device code only matters) and prints one line
per kernel: VGPRs, spilled VGPRs, scratch bytes per lane, occupancy, LDS.  With --fail-on-scratch the exit code is 1
when a listed kernel uses scratch (tests/test_kernel_resources.py builds on this).
"""
import re
import subprocess
import sys


def report(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null", *extra]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, nm in zip(rows, names):
        r["pretty"] = re.sub(r"\(.*", "", nm.replace("void ", "").replace("pfhe::(anonymous namespace)::", "").replace("pfhe::", ""))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fail = "--fail-on-scratch" in sys.argv
    src, subs = args[0], args[1:]
    bad = 0
    for r in report(src):
        if subs and not any(s in r["pretty"] for s in subs):
            continue
        scratch = r.get("ScratchSize", 0)
        bad += scratch > 0
        print(f"{r['pretty']:70s} vgpr {r.get('VGPRs', 0):4d} agpr {r.get('AGPRs', 0):3d} spill {r.get('VGPRs Spill', 0):3d} "
              f"scratch {scratch:4d} occ {r.get('Occupancy', 0)} lds {r.get('LDS Size', 0)}")
    sys.exit(1 if fail and bad else 0)


if __name__ == "__main__":
    main()
