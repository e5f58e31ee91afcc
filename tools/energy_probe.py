#!/usr/bin/env python3
"""Joules per lane-instruction of the VALU instruction classes the butterflies use (tools/microbench9_energy.hip): package
power and shader clock (rocm-smi, every 0.4 s) while every SIMD issues one instruction class from registers at 4 waves.
    gpurun -- 'hipcc --offload-arch=gfx950 -O2 -o tools/microbench9 tools/microbench9_energy.hip; python tools/energy_probe.py'
"""
import re
import statistics
import subprocess
import sys
import time

IDLE_W = None


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(c.group(1)) if c else None)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    idle = [sample()[0] for _ in range(3)]
    print("idle package power: %s W" % idle)
    rows = []
    for op in range(12):
        proc = subprocess.Popen(["tools/microbench9", str(op), str(secs)], stdout=subprocess.PIPE, text=True)
        t0 = time.time()
        samples = []
        while proc.poll() is None:
            s = sample()
            if time.time() - t0 > 1.5 and s[0] is not None:
                samples.append(s)
            time.sleep(0.4)
        out = proc.stdout.read()
        m = re.search(r"PHASE_END (.*) lane_instr_per_s ([0-9.e+]+)", out)
        if not m or not samples:
            print("op", op, "failed", out[-200:])
            continue
        name, rate = m.group(1), float(m.group(2))
        pw = statistics.median(s[0] for s in samples)
        ck = statistics.median(s[1] for s in samples if s[1])
        rows.append((name, rate, pw, ck))
    base = 292.0  # package at full clock without work (profiles/r02_power_probe.txt)
    print("%-42s %14s %9s %9s %12s %14s" % ("instruction", "G lane-inst/s", "power W", "sclk MHz", "cyc/wave-inst", "pJ/lane-inst"))
    for name, rate, pw, ck in rows:
        simds = 256 * 4
        cyc = ck * 1e6 * simds / (rate / 64)
        print("%-42s %14.0f %9.0f %9.0f %12.2f %14.1f" % (name, rate / 1e9, pw, ck, cyc, (pw - base) / rate * 1e12))


if __name__ == "__main__":
    main()
