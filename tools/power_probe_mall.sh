#!/bin/bash
# Package power and rate of a device-to-device copy whose footprint fits the 256 MiB Infinity Cache (2 x SIZE_MIB) against
# one that does not: what a byte costs when it comes from / goes to the Infinity Cache instead of HBM.
cd $GRAFT_REPO_ROOT
for mib in 32 64 96 3072; do
python - $mib <<'PY' &
import sys, time, torch
mib = int(sys.argv[1])
x = torch.empty(mib << 17, dtype=torch.int64, device="cuda"); y = torch.empty_like(x)
x.zero_(); y.zero_(); torch.cuda.synchronize()
reps = max(20, 4096 // mib)
for _ in range(reps): y.copy_(x)
torch.cuda.synchronize()
print("PHASE copy", mib, time.time(), flush=True)
t0 = time.time(); k = 0
while time.time() - t0 < 5:
    for _ in range(reps):
        y.copy_(x); x.copy_(y)
    torch.cuda.synchronize(); k += 2 * reps
dt = (time.time() - t0) / k
print("PHASE_END copy", mib, "MiB: us_each %.1f  %.2f TB/s moved (read+write)" % (dt * 1e6, 2 * mib * 1.048576e6 / dt / 1e12), flush=True)
PY
PID=$!
for i in $(seq 1 40); do if ! kill -0 $PID 2>/dev/null; then break; fi
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk clock level" | sed "s/.*: //"  | tr '\n' ' '; echo
  sleep 0.7
done
wait $PID
done
