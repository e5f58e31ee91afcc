#!/usr/bin/env python3
"""Prints the kernel timeline (start, end, duration in us, relative to the first NTT kernel) of a rocprofv3
kernel-trace CSV: do the strided pass of tile k+1 and the block pass of tile k really run at the same time?"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ntt_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-26:]  # the last transform
t0 = int(rows[0]["Start_Timestamp"])
busy = []
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    k = "strided" if "strided" in r["Kernel_Name"] else "block  "
    print(f"{k} grid {int(r.get('Grid_Size', r.get('Grid_Size_X', 0))):>9d} queue {r.get('Queue_Id', '?'):>3s}  {s:9.1f} -> {e:9.1f} us  ({e - s:7.1f})")
    busy.append((s, e))
print("span %.1f us" % (max(e for _, e in busy) - min(s for s, _ in busy)))
