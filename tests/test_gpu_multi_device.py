"""One process driving several shards from host threads ("one host thread + stream per device"), through
handles that carry their device.  With one GPU the shards share device 0 (the code path, not the scaling); on a box with
more devices the same test spreads them over all of them.  The repository's measured multi-GPU path is one process per
GPU (bench.py --gpus N, tests/test_gpu_shard.py)."""
import numpy as np
import pytest

from golden_inputs import fill_uniform_words
from gpu_util import to_host
from pyref import Q61

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,total", [(13, 7), (16, 5)], ids=["n8192-7ct", "config4-shape-n65536-5ct"])
def test_threads_per_device_union_equals_unsharded_and_oracle(orc, log_n, total):
    """Ciphertexts of the external product (k = 1, 3 primes of 61 bits, log B = 30, ell = 6) and RNS polynomials of the forward
    transform, split over host threads with a handle set per shard.  Two shapes: N = 2^13 with 7 ciphertexts (ragged 3 + 2 +
    2), and BASELINE config 4's ring and gadget — N = 2^16 — with 5 ciphertexts (the batch is what is reduced, not the
    shape).  VERDICT r4 item 13: earlier text called the first one "config 4"; it is not."""
    import torch

    import primus_fhe_amd as p
    from primus_fhe_amd.shard import fill_job_shard, run_on_devices, shard_range

    ndev = torch.cuda.device_count()
    shards = max(3, ndev)                      # ragged on purpose when there is one device
    devices = [r % ndev for r in range(shards)]
    k = 1
    n, L = 1 << log_n, 3
    W = (k + 1) * L * n
    seed_g, seed_k = 0x5EED000000000011, 0x5EED000000000012

    def worker(rank, device, begin, end):
        with torch.cuda.device(device):
            stream = torch.cuda.Stream(device=device)
            table = p.U64DcrtTable(log_n, Q61, device=device)
            base = p.RNSBase(Q61, device=device)
            ctx = p.DcrtGlevContext(table, base, p.BigUintApproxSignedBasis(base, 30), k)
            ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int64, device=f"cuda:{device}")
            fill_job_shard(p.lib(), device, ggsw.data_ptr(), 0, 1, ctx.ggsw_len(), Q61, n, seed_k)
            units = end - begin
            g = torch.empty(max(1, units) * W, dtype=torch.int64, device=f"cuda:{device}")
            fill_job_shard(p.lib(), device, g.data_ptr(), begin, units, W, Q61, n, seed_g)
            out = torch.empty_like(g)
            if units:
                p.mul_dcrt_ggsw_to_dev(g[:units * W], ggsw, out[:units * W], ctx, into_coeff_form=True, stream=stream)
                table.transform_dev(g[:units * W], stream=stream)      # and a plain transform of the shard's input
            stream.synchronize()
            return (begin, end), to_host(out[:units * W]), to_host(g[:units * W])

    results = run_on_devices(devices, total, worker)
    assert [r[0] for r in results] == [shard_range(total, shards, r) for r in range(shards)]
    prod = np.concatenate([r[1] for r in results])
    fwd = np.concatenate([r[2] for r in results])
    # the job's input on the host model of the synthetic fill, the oracle on every ciphertext
    gh = fill_uniform_words(seed_g, 0, total * W, Q61, n)
    kh = fill_uniform_words(seed_k, 0, (k + 1) * 6 * (k + 1) * L * n, Q61, n)
    o, ob = orc.U64DcrtTable(log_n, Q61), orc.RNSBase(Q61)
    obasis = orc.BigUintApproxSignedBasis(ob, 30)
    assert obasis.decompose_length == 6
    for e in range(total):
        r = orc.mul_dcrt_ggsw_to(o, ob, obasis, k, gh[e * W:(e + 1) * W].copy(), kh)
        o.inverse_transform_slice(r)
        assert np.array_equal(prod[e * W:(e + 1) * W], r), "ciphertext %d" % e
    exp = gh.copy()
    o.transform_slice(exp)
    assert np.array_equal(fwd, exp)
