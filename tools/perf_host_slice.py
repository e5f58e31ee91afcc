import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import primus_fhe_amd as p
Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
for log_n, L, batch in ((12,1,1),(16,1,1),(16,3,1),(16,3,16)):
    t = p.U64DcrtTable(log_n, Q61[:L])
    a = np.random.default_rng(0).integers(0, Q61[0]-10**6, batch*L<<log_n, dtype=np.uint64)
    t.transform_slice(a)
    t0=time.perf_counter()
    for _ in range(20): t.transform_slice(a)
    dt=(time.perf_counter()-t0)/20
    print(f"transform_slice logN={log_n} L={L} batch={batch}: {dt*1e6:.0f} us per call ({a.nbytes/dt/1e9:.2f} GB/s)")
