import sys, os
sys.path.insert(0, '/root/repo')
import ctypes as C, numpy as np, torch
import primus_fhe_amd as p
from primus_fhe_amd._lib import check, u64p
Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
n, L, batch = 1 << 16, 3, 4096
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
mods = np.array(Q61, np.uint64)
def fill(w, seed):
    x = torch.empty(w, dtype=torch.int64, device="cuda")
    check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), w, mods.ctypes.data_as(u64p), L, n, seed, None)); return x
a, b, c = fill(words, 1), fill(words, 2), fill(words, 3)
bs = b[:L * n].clone()
st = torch.cuda.current_stream()
def timed(fn, reps=10):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize(); return e0.elapsed_time(e1) / reps
for name, fn, nbytes in (("mul_assign shared b", lambda: t.mul_assign_dev(a, bs), 16 * words),
                         ("mul_assign per-element b", lambda: t.mul_assign_dev(a, b), 24 * words),
                         ("add_mul_assign per-element", lambda: t.add_mul_assign_dev(c, a, b), 32 * words)):
    ms = timed(fn)
    print(f"pointwise {name}: {ms:.3f} ms, {nbytes / ms / 1e6:.0f} GB/s")
