#!/usr/bin/env python3
"""Does a buffer just WRITTEN by one kernel come back from the Infinity Cache when the next kernel reads it?
Read bandwidth (torch sum over int64) of a buffer of S MiB right after it was written (fill), against the same read after
1 GiB of unrelated traffic evicted it; and the read-after-read case."""
import torch

dev = "cuda"
big = torch.empty(1 << 27, dtype=torch.int64, device=dev)  # 1 GiB evictor
for mib in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mib * (1 << 20) // 8
    x = torch.empty(n, dtype=torch.int64, device=dev)
    res = {}
    for mode in ("after_write", "after_evict", "after_read"):
        ts = []
        for _ in range(12):
            if mode == "after_write":
                x.fill_(3)
            elif mode == "after_evict":
                x.fill_(3)
                big.fill_(1)
            else:
                x.fill_(3)
                big.fill_(1)
                x.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x.sum()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = mib / 1024 * 1.073741824 / (ts[len(ts) // 2] * 1e-3) / 1e3  # TB/s
    print(f"{mib:5d} MiB  read after write {res['after_write']:.2f} TB/s  after eviction {res['after_evict']:.2f}  after read {res['after_read']:.2f}")
