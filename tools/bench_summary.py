#!/usr/bin/env python3
"""Prints the headline numbers of a bench.py JSON line (stdin or file argument)."""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print(f"headline: {d['value'] / 1e6:.3f} M NTT/s  {d['ms_per_step']:.3f} ms/step  frac {d['hbm_roofline_frac']:.3f}  n_gpus {d['n_gpus']}")
if "roofline" in d:
    r = d["roofline"]
    print(f"roofline: {r['kernel']}  {r['avg_launch_ms']:.3f} ms  {r['achieved']:.0f} GB/s  frac {r['frac']:.3f}")
    if "roofline_passes" in d:
        r = d["roofline_passes"]
        print(f"stand-alone: {r['kernel']}  {r['avg_launch_ms']:.3f} ms  {r['achieved']:.0f} GB/s  frac {r['frac']:.3f}")
    print("kernels_ms:", {k: round(v, 3) for k, v in d.get("kernels_ms", {}).items()})
if "sustained" in d:
    u = d["sustained"]
    print(f"sustained: {u['value'] / 1e6:.3f} M NTT/s  {u['ms_per_step']:.3f} ms/step over {u['seconds']:.1f} s  frac {u['hbm_roofline_frac']:.3f}")
for k in ("intt", "ntt_generic_prime", "polymul", "ntt_u32"):
    if k in d:
        print(f"{k}: {d[k]['value'] / 1e3:.1f} k/s  {d[k]['ms_per_batch']:.3f} ms  frac {d[k]['hbm_roofline_frac']:.3f}")
if "ntt_2p14" in d:
    for w in ("forward", "inverse"):
        v = d["ntt_2p14"][w]
        print(f"ntt_2p14 {w}: {v['value'] / 1e6:.2f} M/s  {v['ms_per_batch']:.3f} ms  frac {v['hbm_roofline_frac']:.3f}")
e = d.get("external_product", {})
if e:
    print(f"external_product: {e['value'] / 1e3:.2f} k/s  {e['ms_per_batch']:.2f} ms/batch  frac {e['hbm_roofline_frac']:.4f}")
    if "roofline" in e:
        print("  roofline:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in e["roofline"].items() if k != "note"})
c5 = d.get("external_product_config5")
if c5:
    print(f"config5: {c5['value'] / 1e3:.2f} k/s  total {c5['batch_total']} scaling {c5['scaling']} seconds {c5['seconds']:.3f}")
c = d.get("cpu_baseline")
if c:
    print(f"cpu: {c['value'] / 1e3:.1f} k NTT/s on {c['cores']} cores ({c['backend']}); ext {c.get('external_product', {}).get('value')}; "
          f"polymul {c.get('polymul', {}).get('value')}")
