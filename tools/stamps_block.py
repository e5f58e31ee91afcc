#!/usr/bin/env python3
"""Phase timeline of the forward block pass from a -DPFHE_STAMPS build (tools/build_variant.sh stamps -DPFHE_STAMPS):
median cycles wave 0 of a workgroup spends in each phase, over the first 65536 workgroups of a full-size launch."""
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
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
for _ in range(3):
    check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, C.c_void_p(x.data_ptr()), words, 0, 1, 0, None))
torch.cuda.synchronize()
wgs = 1 << 16
buf = np.zeros((wgs, 12), np.uint64)
rd = p.lib().pfhe_debug_read_stamps
rd.restype = C.c_int
rd.argtypes = [C.c_void_p, C.c_size_t]
assert rd(buf.ctypes.data, wgs) == 0
names = ["issue global loads -> landed", "LDS staging + barrier", "register pass 1 (uniform twiddles)", "exchange 1",
         "register pass 2 (4 twiddle sets per wave)", "exchange 2", "register pass 3 (per-lane twiddles) + finish",
         "(core end)", "write-back staging + barrier", "LDS read + global stores landed"]
d = np.diff(buf[:, :11].astype(np.int64), axis=1)
life = (buf[:, 10] - buf[:, 0]).astype(np.int64)
ok = life > 0
print(f"workgroups sampled {ok.sum()}, wave-0 lifetime median {np.median(life[ok]):.0f} cycles, p10 {np.percentile(life[ok], 10):.0f}, "
      f"p90 {np.percentile(life[ok], 90):.0f}")
for i, nm in enumerate(names):
    col = d[ok, i]
    print(f"  {nm:48s} median {np.median(col):8.0f}  mean {col.mean():8.0f}  p90 {np.percentile(col, 90):8.0f}")
