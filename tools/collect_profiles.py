#!/usr/bin/env python3
"""Copy the summaries of the last tools/profile_round.sh + tools/profile_pmc_util.sh run (gpurun_out/) into
profiles/ under the given prefix (default r01_e)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r01_e"
src, util, dst = os.path.join(ROOT, "gpurun_out", "r01e"), os.path.join(ROOT, "gpurun_out", "util"), os.path.join(ROOT, "profiles")


def short(n):
    return n.replace("void ", "").replace("pfhe::(anonymous namespace)::", "").replace("pfhe::", "").split("(")[0]


tr = sorted(glob.glob(src + "/trace_bench/runc/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
st = sorted(glob.glob(src + "/trace_bench/runc/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
    d[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (grouped by kernel and grid size,",
         "because the timed steps launch the two passes per 1/8-batch tile on two streams while the roofline leg launches full-size passes)",
         f"{'kernel':62s} {'grid':>10s} {'n':>4s} {'avg ms':>8s} {'min ms':>8s} {'max ms':>8s}"]
out = []
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("__amd"):
        continue
    lines.append(f"{k:62s} {g:10d} {len(v):4d} {sum(v) / len(v) / 1e6:8.3f} {min(v) / 1e6:8.3f} {max(v) / 1e6:8.3f}")
    out.append({"kernel": k, "grid": g, "launches": len(v), "avg_ms": sum(v) / len(v) / 1e6})
open(f"{dst}/{prefix}_bench_kernel_trace.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(f"{dst}/{prefix}_bench_kernel_trace.json", "w"), indent=1)
shutil.copy(st, f"{dst}/{prefix}_bench_kernel_stats.csv")
for ext in ("txt", "json"):
    shutil.copy(f"{src}/r01_e_rocprof.{ext}", f"{dst}/{prefix}_rocprof.{ext}")
b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
json.dump(b, open(f"{dst}/{prefix}_bench.json", "w"), indent=1)
if os.path.exists(util + "/summary.txt"):
    head = "rocprofv3 --pmc <one counter per pass> -- python3 tools/profile_ntt.py (PFHE_PROFILE_BATCH=2048), tools/profile_pmc_util.sh; percentages, averaged over launches\n"
    open(f"{dst}/{prefix}_pmc_utilisation.txt", "w").write(head + open(util + "/summary.txt").read())
print("\n".join(lines[:12]))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.items() if k in ("value", "ms_per_step")},
      b["roofline"]["avg_launch_ms"], b["external_product"]["value"], b["polymul"]["value"], b["ntt_u32"]["value"],
      b.get("cpu_baseline", {}).get("value"), b.get("cpu_baseline", {}).get("external_product"))
