#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round2.sh run (gpurun_out/TAG/) into profiles/ under the prefix TAG:
bench line, kernel trace of the same bench command grouped by kernel and grid size, HBM-counter summary, utilisation
counters of the NTT kernels, and the external product's kernel breakdown + counters at the bench shape."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_a"
# (run on the GPU box, where only gpurun_out/ travels back: the summaries go to gpurun_out/TAG/summaries/ and are copied
# into profiles/ by hand afterwards:  cp gpurun_out/TAG/summaries/* profiles/)
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(src, "summaries")
os.makedirs(dst, exist_ok=True)


def short(n):
    return n.replace("void ", "").replace("pfhe::(anonymous namespace)::", "").replace("pfhe::", "").split("(")[0]


def grid_of(r):
    return int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


# ---- bench line + kernel trace of the same command
b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
json.dump(b, open(f"{dst}/{tag}_bench.json", "w"), indent=1)
tr, st = newest(src + "/trace_bench/**/*_kernel_trace.csv"), newest(src + "/trace_bench/**/*_kernel_stats.csv")
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    d[(short(r["Kernel_Name"]), grid_of(r))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --ext-total 0   (grouped by kernel and grid size:",
         "the timed steps are 25 launches (24 tiles of 256 MiB + 1) of ntt_pipe_fwd_kernel per transform, the stand-alone leg launches full-size passes)",
         f"{'kernel':64s} {'grid':>10s} {'n':>4s} {'avg ms':>8s} {'min ms':>8s} {'max ms':>8s}"]
out = []
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("__amd"):
        continue
    lines.append(f"{k:64s} {g:10d} {len(v):4d} {sum(v) / len(v) / 1e6:8.3f} {min(v) / 1e6:8.3f} {max(v) / 1e6:8.3f}")
    out.append({"kernel": k, "grid": g, "launches": len(v), "avg_ms": sum(v) / len(v) / 1e6})
open(f"{dst}/{tag}_bench_kernel_trace.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(f"{dst}/{tag}_bench_kernel_trace.json", "w"), indent=1)
shutil.copy(st, f"{dst}/{tag}_bench_kernel_stats.csv")
for ext in ("txt", "json"):
    if os.path.exists(f"{src}/rocprof.{ext}"):
        shutil.copy(f"{src}/rocprof.{ext}", f"{dst}/{tag}_rocprof.{ext}")


def counter_table(prefix, title):
    table = collections.defaultdict(dict)
    for dpath in sorted(glob.glob(f"{src}/{prefix}*/")):
        c = os.path.basename(dpath.rstrip("/"))[len(prefix):]
        f = newest(dpath + "/**/*counter_collection.csv")
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), grid_of(r))].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            table[k][c] = (sum(v) / len(v), len(v))
    cols = sorted({c for v in table.values() for c in v})
    rows = [title, f"{'kernel':60s} {'grid':>10s} {'n':>5s} " + " ".join(f"{c:>16s}" for c in cols)]
    for k in sorted(table, key=lambda k: (k[0], -k[1])):
        if k[0].startswith("__amd") or "fill" in k[0]:
            continue
        nlaunch = max(n for _, n in table[k].values())
        rows.append(f"{k[0]:60s} {k[1]:10d} {nlaunch:5d} " + " ".join(f"{table[k].get(c, (float('nan'), 0))[0]:16.2f}" for c in cols))
    return rows


open(f"{dst}/{tag}_pmc_utilisation.txt", "w").write("\n".join(counter_table(
    "util_", "rocprofv3 --pmc <one counter per pass> -- python3 tools/profile_ntt.py (PFHE_PROFILE_BATCH=2048); averages over launches")) + "\n")
# ---- external product at the bench shape
ep = ["tools/perf_extprod.py (batch 1024, default chunk of 128 ciphertexts): " + " | ".join(
    l.strip() for l in open(src + "/ep.log") if "ext-products" in l)]
stf = newest(src + "/ep_trace/**/*_kernel_stats.csv")
if stf:
    ep.append("rocprofv3 --kernel-trace --stats (same command):")
    for r in csv.DictReader(open(stf)):
        ep.append(f"  {short(r['Name']):58s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
ep += counter_table("ep_", "rocprofv3 --pmc <one counter per pass> -- python3 tools/perf_extprod.py (FETCH_SIZE / WRITE_SIZE in KiB as "
                    "reported: reads = 2 x FETCH_SIZE on gfx950)")
# whole-product HBM traffic: every launch of the COEFF_ONLY counter passes (coefficient form), 2 x FETCH_SIZE + WRITE_SIZE
# (KiB as reported; gfx950 correction of MI355X_MICROARCH.md), summed and divided by the products
try:  # the code the counters were taken on (bench.py compares it with the library it times)
    sys.path.insert(0, ROOT)
    import primus_fhe_amd as _p
    from primus_fhe_amd._codeobj import kernel_code_hashes
    ep_code = kernel_code_hashes(_p.library_path())
except Exception as _e:
    print("collect_profiles2: no code hashes:", _e)
    ep_code = {}


def product_traffic(prefix, products, form, alg_bytes):
    vg = {}

    def counter_sum(name):
        f = newest(f"{src}/{prefix}{name}/**/*counter_collection.csv")
        per = collections.defaultdict(float)
        if f:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not (k.startswith("__amd") or "fill" in k):
                    per[k] += float(r["Counter_Value"])
                    if r.get("VGPR_Count"):
                        vg[k] = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0)
        return per

    fe, wr = counter_sum("FETCH_SIZE"), counter_sum("WRITE_SIZE")
    if not (fe and wr):
        return None
    per_kernel = {k: (2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024 / products for k in sorted(set(fe) | set(wr))}
    return {"products": products, "form": form,
            "method": "sum over every kernel launch of 2*FETCH_SIZE + WRITE_SIZE (separate --pmc passes, KiB), / products",
            "bytes_per_product": sum(per_kernel.values()), "bytes_per_product_by_kernel": per_kernel,
            "vgpr_count_by_kernel": vg, "code_sha256_by_kernel": {k: ep_code.get(k) for k in vg},
            "algorithmic_bytes_per_product": alg_bytes}


traffic = product_traffic("ep_", 4 * int(os.environ.get("BATCH", "1024")),
                          "CrtGlwe x DcrtGgsw -> coefficient form, batch 1024, default chunk", 96 * 65536)
if traffic:
    json.dump(traffic, open(f"{dst}/{tag}_extprod_traffic.json", "w"), indent=1)
    ep.append("whole product: %.1f MB moved per product (algorithmic 6.29 MB): " % (traffic["bytes_per_product"] / 1e6) +
              ", ".join("%s %.1f" % (k, v / 1e6) for k, v in sorted(traffic["bytes_per_product_by_kernel"].items(), key=lambda kv: -kv[1])))
# the <u32> product: tools/perf_extprod32.py under COEFF_ONLY runs 1 + 5 products of the batch
t32 = product_traffic("ep32_", 6 * int(os.environ.get("BATCH", "1024")),
                      "CrtGlwe<u32> x DcrtGgsw over U32DcrtTable -> coefficient form, batch 1024, fused kernels", 48 * 65536)
if t32:
    json.dump(t32, open(f"{dst}/{tag}_extprod32_traffic.json", "w"), indent=1)
    ep.append("u32 product: %.1f MB moved per product (algorithmic 3.15 MB): " % (t32["bytes_per_product"] / 1e6) +
              ", ".join("%s %.1f" % (k, v / 1e6) for k, v in sorted(t32["bytes_per_product_by_kernel"].items(), key=lambda kv: -kv[1])))
    ep += ["tools/perf_extprod32.py: " + " | ".join(l.strip() for l in open(src + "/ep32.log") if "products/s" in l)]
    stf32 = newest(src + "/ep32_trace/**/*_kernel_stats.csv")
    if stf32:
        for r in csv.DictReader(open(stf32)):
            ep.append(f"  {short(r['Name']):58s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
    ep += counter_table("ep32_", "rocprofv3 --pmc <one counter per pass> -- python3 tools/perf_extprod32.py (COEFF_ONLY)")
open(f"{dst}/{tag}_extprod_pmc.txt", "w").write("\n".join(ep) + "\n")
print("\n".join(lines[:10]))
print("\n".join(ep[:14]))
