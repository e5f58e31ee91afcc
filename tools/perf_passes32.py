#!/usr/bin/env python3
"""Per-pass timing of the forward/inverse u32 RNS NTT (U32DcrtTable), HIP events on the launch stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402

Q30 = [1073479681, 1071513601, 1070727169]
log_n = int(os.environ.get("LOG_N", "16"))
batch = int(os.environ.get("BATCH", "4096"))
reps = int(os.environ.get("REPS", "5"))
n, L = 1 << log_n, 3
t = p.U32DcrtTable(log_n, Q30)
words = batch * L * n
x = torch.empty(words, dtype=torch.int32, device="cuda")
t.fill_uniform_dev(x, 1)
stream = torch.cuda.current_stream()


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for inverse in (False, True):
    for i in range(t.transform_num_passes()):
        ms = timed(lambda: t.transform_pass_dev(x, inverse, i))
        out.append(f"{t.transform_pass_name(inverse, i)}={ms:.3f}ms({8 * n * batch * L / ms / 1e6:.0f}GB/s)")
for inverse, fn in ((0, t.transform_dev), (1, t.inverse_transform_dev)):
    ms = timed(lambda: fn(x))
    out.append(f"{'inv' if inverse else 'fwd'}_total={ms:.3f}ms({batch * L / ms / 1e3:.3f}M NTT/s)")
y = x.clone()
ms = timed(lambda: t.mul_assign_dev(x, y))
out.append(f"mul_assign={ms:.3f}ms({12 * n * batch * L / ms / 1e6:.0f}GB/s)")
print(f"u32 logN={log_n} batch={batch}:", "  ".join(out))
