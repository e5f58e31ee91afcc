#!/usr/bin/env python3
"""Runs only the forward block pass (and strided pass) a few times at the bench shape — target for
rocprofv3 --pmc SQ counter collection."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
batch = int(os.environ.get("BATCH", "2048"))
n, L = 1 << 16, 3
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
for i in (0, 1, 1, 1):
    check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, C.c_void_p(x.data_ptr()), words, 0, i, 0, None))
torch.cuda.synchronize()
