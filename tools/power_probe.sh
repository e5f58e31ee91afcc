#!/bin/bash
# samples rocm-smi power / clocks while a long loop of transforms runs
cd $GRAFT_REPO_ROOT
python - <<'PY' &
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import primus_fhe_amd as p
from primus_fhe_amd._lib import check, u64p
Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
n, L, batch = 1 << 16, 3, 4096
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
def phase(name, fn, secs):
    torch.cuda.synchronize(); print("PHASE", name, time.time(), flush=True)
    t0 = time.time(); k = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); k += 20
    print("PHASE_END", name, time.time(), "ms_each", (time.time() - t0) / k * 1e3, flush=True)
phase("idle", lambda: time.sleep(0.05), 3)
phase("forward_transform", lambda: t.transform_dev(x), 6)
phase("block_pass_only", lambda: check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, C.c_void_p(x.data_ptr()), words, 0, 1, 0, None)), 6)
phase("strided_pass_only", lambda: check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, C.c_void_p(x.data_ptr()), words, 0, 0, 0, None)), 6)
PY
PID=$!
for i in $(seq 1 50); do
  echo "T $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | head -6
  sleep 0.5
done
wait $PID
