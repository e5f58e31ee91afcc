#!/bin/bash
# companion of power_probe.sh: package power while a plain device-to-device copy of the bench batch (6 GiB read + 6 GiB
# written per copy) runs for 6 s: the energy of moving the bytes alone
cd $GRAFT_REPO_ROOT
python - <<'PY' &
import time, torch
x = torch.empty(4096 * 3 * 65536, dtype=torch.int64, device="cuda"); y = torch.empty_like(x)
x.zero_(); torch.cuda.synchronize()
print("PHASE copy", time.time(), flush=True)
t0 = time.time(); k = 0
while time.time() - t0 < 6:
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize(); k += 20
print("PHASE_END copy", time.time(), "ms_each", (time.time() - t0) / k * 1e3, flush=True)
PY
PID=$!
for i in $(seq 1 120); do if ! kill -0 $PID 2>/dev/null; then break; fi
  echo "T $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -3
  sleep 0.5
done
wait $PID
