"""BASELINE config 5 on the hardware there is (one GPU): the sharded external-product job through the HIP library.

* two and EIGHT ranks on ONE device, started by bench.py's own `--gpus N` entry (gloo barrier): the union of the ranks'
  HIP outputs equals the unsharded HIP output and the oracle, bit for bit (ragged splits);
* the full 8192-ciphertext job on one GPU, unsharded and as the eight `shard_range(8192, 8, r)` pieces, with oracle
  checks on the ciphertexts either side of every shard boundary.

The sharding contract (primus_ntt/src/dcrt/mod.rs:19 `Send + Sync`): ciphertexts are independent, a
rank's result depends only on which ciphertexts it owns.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_inputs import fill_uniform_words
from gpu_util import to_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

LOG_N, K, LOG_BASIS = 16, 1, 30
N = 1 << LOG_N


def _setup(p, chunk=0):
    import bench
    table = p.U64DcrtTable(LOG_N, bench.Q61, device=0)
    base = p.RNSBase(bench.Q61, device=0)
    basis = p.BigUintApproxSignedBasis(base, LOG_BASIS)
    ctx = p.DcrtGlevContext(table, base, basis, K, chunk)
    return bench, table, ctx


def _oracle_setup(orc, moduli):
    o, obase = orc.U64DcrtTable(LOG_N, moduli), orc.RNSBase(moduli)
    return o, obase, orc.BigUintApproxSignedBasis(obase, LOG_BASIS)


def _oracle_product(orc, o, obase, obasis, glwe, ggsw):
    """CrtGlwe::mul_dcrt_ggsw_to + DcrtGlwe::into_coeff_form (glwe/crt.rs:200-227, macros/mod.rs:892-937)."""
    r = orc.mul_dcrt_ggsw_to(o, obase, obasis, K, glwe.copy(), ggsw)
    o.inverse_transform_slice(r)
    return r


@pytest.mark.parametrize("world,batch,total,extra", [
    (2, 3, 5, []),                          # ragged: 3 + 2
    (8, 2, 13, ["--skip-device-check"]),    # BASELINE config 5's world size: eight rank processes, ragged 2,2,2,2,2,1,1,1
])
def test_ranks_on_one_device_through_bench_entry(tmp_path, orc, world, batch, total, extra):
    """`bench.py --gpus N` exactly as the driver's launcher would run the N ranks — N real processes, rendezvous on
    127.0.0.1, barrier + max-reduce (gloo), N dumps — on the one device there is.  The union of the ranks' HIP outputs
    must equal the unsharded HIP result and the oracle."""
    import torch

    import primus_fhe_amd as p
    from primus_fhe_amd.shard import fill_job_shard, shard_range

    # batch: per-rank RNS polynomials of the NTT leg; total: ciphertexts of the config-5 job
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--one-device", "--dist-backend", "gloo",
           "--batch", str(batch), "--ext-batch", "2", "--ext-total", str(total), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--dump-dir", str(tmp_path)] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == 2 and line["value"] > 0
    c5 = line["external_product_config5"]
    assert c5["n_gpus"] == world and c5["batch_total"] == total and c5["scaling"] == "strong"
    assert line["external_product"]["n_gpus"] == world

    dumps = [np.load(os.path.join(tmp_path, "rank%d.npz" % rk)) for rk in range(world)]
    ranges = [tuple(int(v) for v in d["config5_range"]) for d in dumps]
    assert ranges == [shard_range(total, world, rk) for rk in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    if world == 2:
        assert ranges == [(0, 3), (3, 5)]
    assert [tuple(d["ntt_range"]) for d in dumps] == [(rk * batch, (rk + 1) * batch) for rk in range(world)]
    ntt_union = np.concatenate([d["ntt_out"] for d in dumps])
    c5_union = np.concatenate([d["config5_out"] for d in dumps])

    bench, table, ctx = _setup(p)
    L, W = 3, 2 * 3 * N
    # ---- NTT leg: the job of world*batch RNS polynomials, unsharded on the HIP path, and the oracle on the host model
    #      of the synthetic input (which also pins pfhe_fill_uniform_dev's position-only seeding) ----
    x = torch.empty(world * batch * L * N, dtype=torch.int64, device="cuda")
    fill_job_shard(p.lib(), 0, x.data_ptr(), 0, world * batch, L * N, bench.Q61, N, bench.SEED_NTT)
    host_in = fill_uniform_words(bench.SEED_NTT, 0, world * batch * L * N, bench.Q61, N)
    assert np.array_equal(to_host(x), host_in)
    table.transform_dev(x)
    assert np.array_equal(ntt_union, to_host(x)), "union of the ranks' NTT shards != unsharded HIP transform"
    o, obase, obasis = _oracle_setup(orc, bench.Q61)
    exp = host_in.copy()
    o.transform_slice(exp)
    assert np.array_equal(ntt_union, exp), "sharded NTT != oracle"
    # ---- config 5: the job of `total` ciphertexts ----
    g = torch.empty(total * W, dtype=torch.int64, device="cuda")
    fill_job_shard(p.lib(), 0, g.data_ptr(), 0, total, W, bench.Q61, N, bench.SEED_CONFIG5)
    ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int64, device="cuda")
    fill_job_shard(p.lib(), 0, ggsw.data_ptr(), 0, 1, ctx.ggsw_len(), bench.Q61, N, bench.SEED_GGSW)
    out = torch.empty_like(g)
    p.mul_dcrt_ggsw_to_dev(g, ggsw, out, ctx, into_coeff_form=True)
    assert np.array_equal(c5_union, to_host(out)), "union of the ranks' products != unsharded HIP products"
    gh, kh = to_host(g), to_host(ggsw)
    for e in range(total):
        exp = _oracle_product(orc, o, obase, obasis, gh[e * W:(e + 1) * W], kh)
        assert np.array_equal(c5_union[e * W:(e + 1) * W], exp), "ciphertext %d of the sharded job != oracle" % e


def test_config5_job_of_8192_ciphertexts_sharded_eight_ways(orc):
    """The whole config-5 job on one GPU: once unsharded (one call over 8192 ciphertexts = 3.2 G words per operand, past
    32-bit word indices), once as the eight ranks' shards.  Shards must equal the matching slices of the unsharded
    result; the ciphertexts either side of every shard boundary are checked against the oracle."""
    import torch

    import primus_fhe_amd as p
    from primus_fhe_amd.shard import fill_job_shard, shard_range

    total, world = 8192, 8
    free, _ = torch.cuda.mem_get_info()
    W = 2 * 3 * N
    if free < (2 * total + 2 * (total // world)) * W * 8 + (8 << 30):
        pytest.skip("needs ~60 GiB of free HBM")
    bench, table, ctx = _setup(p)
    g = torch.empty(total * W, dtype=torch.int64, device="cuda")
    fill_job_shard(p.lib(), 0, g.data_ptr(), 0, total, W, bench.Q61, N, bench.SEED_CONFIG5)
    ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int64, device="cuda")
    fill_job_shard(p.lib(), 0, ggsw.data_ptr(), 0, 1, ctx.ggsw_len(), bench.Q61, N, bench.SEED_GGSW)
    out = torch.empty_like(g)
    p.mul_dcrt_ggsw_to_dev(g, ggsw, out, ctx, into_coeff_form=True)
    torch.cuda.synchronize()

    o, obase, obasis = _oracle_setup(orc, bench.Q61)
    kh = to_host(ggsw)
    per = total // world
    gs = torch.empty(per * W, dtype=torch.int64, device="cuda")
    os_ = torch.empty_like(gs)
    for rk in range(world):
        b, e = shard_range(total, world, rk)
        assert (b, e) == (rk * per, (rk + 1) * per)
        fill_job_shard(p.lib(), 0, gs.data_ptr(), b, e - b, W, bench.Q61, N, bench.SEED_CONFIG5)
        assert torch.equal(gs, g[b * W:e * W]), "rank %d's input is not its slice of the job" % rk
        os_.zero_()
        p.mul_dcrt_ggsw_to_dev(gs, ggsw, os_, ctx, into_coeff_form=True)
        assert torch.equal(os_, out[b * W:e * W]), "rank %d's products differ from the unsharded job's" % rk
        for ct in (b, e - 1):  # either side of the boundaries b and e
            gin = fill_uniform_words(bench.SEED_CONFIG5, ct * W, W, bench.Q61, N)
            exp = _oracle_product(orc, o, obase, obasis, gin, kh)
            got = to_host(os_[(ct - b) * W:(ct - b + 1) * W])
            assert np.array_equal(got, exp), "ciphertext %d (rank %d) != oracle" % (ct, rk)


def test_bench_under_the_launcher_with_rccl(tmp_path):
    """The driver's form: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` over RCCL (backend "nccl").
    One GPU here, so N = 1 with --force-dist: the process group is initialised on the device, the timing barrier and the
    max-over-ranks all-reduce run through RCCL, and the JSON line comes out of rank 0."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2",
           "--warmup", "1", "--batch", "16", "--ext-batch", "4", "--ext-total", "8", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["external_product_config5"]["batch_total"] == 8
    assert "roofline" in line
