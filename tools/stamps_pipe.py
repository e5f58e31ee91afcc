#!/usr/bin/env python3
"""Phase timeline of the pipelined forward kernel from a -DPFHE_STAMPS build (tools/build_variant.sh stamps -DPFHE_STAMPS -DPFHE_STAMPS_FULL_ONLY;
PFHE_LIB_PATH=.../libpfhe_hip_stamps.so): median cycles wave 0 of a workgroup spends in each phase, over the workgroups
of the last full launch (block pass of tile k-1 + strided pass of tile k)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
batch = int(os.environ.get("BATCH", "4096"))
n, L = 1 << 16, 3
os.environ["PFHE_OVERLAP_TILES"] = "2"   # two tiles: launch 1 of 3 is a full launch and the last one to write all stamps
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
for _ in range(3):
    t.transform_dev(x)
torch.cuda.synchronize()
wgs = 1 << 16
buf = np.zeros((wgs, 14), np.uint64)
rd = p.lib().pfhe_debug_read_stamps
rd.restype = C.c_int
rd.argtypes = [C.c_void_p, C.c_size_t]
assert rd(buf.ctypes.data, wgs) == 0
names = ["block: loads issued (direct 8-byte loads)", "-", "-> first register pass done", "exchange 1",
         "register pass 2", "exchange 2", "register pass 3 + finish", "(core end)", "write-back staging + barrier",
         "LDS read + global stores issued", "wait: block stores + strided loads landed", "strided: 4 stages in registers",
         "strided stores landed"]
d = np.diff(buf[:, :14].astype(np.int64), axis=1)
life = (buf[:, 13] - buf[:, 0]).astype(np.int64)
ok = (life > 0) & (life < 10**7)
print(f"workgroups sampled {ok.sum()}, wave-0 lifetime median {np.median(life[ok]):.0f} cycles, p10 {np.percentile(life[ok], 10):.0f}, "
      f"p90 {np.percentile(life[ok], 90):.0f}")
for i, nm in enumerate(names):
    col = d[ok, i]
    print(f"  {nm:48s} median {np.median(col):8.0f}  mean {col.mean():8.0f}  p90 {np.percentile(col, 90):8.0f}")
