"""N > 1 path on CPU: two gloo ranks shard a batch (no data-path collective), and the union of the
per-rank results equals the unsharded result.  The per-rank "device work" here is the oracle —
tests may use it; the point is the sharding / barrier / max-reduce logic bench.py relies on."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import oracle
    from primus_fhe_amd.shard import shard_range, timed_steps

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    moduli = [2305843009211596801, 2305843009210023937, 2305843009208713217]
    log_n, batch = 6, 7  # 7 does not divide evenly: ragged shards
    n = 1 << log_n
    rng = np.random.default_rng(0)  # same seed on every rank -> same global batch
    data = np.concatenate([rng.integers(0, m, n, dtype=np.uint64) for _ in range(batch) for m in moduli])
    table = oracle.U64DcrtTable(log_n, moduli)
    b, e = shard_range(batch, world, rank)
    mine = data[b * 3 * n:e * 3 * n].copy()
    calls = []

    def step():
        calls.append(1)
        if len(calls) == 1:
            table.transform_slice(mine)

    dt = timed_steps(step, steps=2, warmup=1, sync=lambda: None, dist=dist, device="cpu")
    # BASELINE config 5 (strong scaling): a fixed job of `batch` units split over the ranks — the code path bench.py
    # runs for its external_product_config5 leg, with the oracle standing in for the device work
    from primus_fhe_amd.shard import strong_scaling_leg
    seen = []

    def run_shard(begin, end):
        seen.append((begin, end))
        work = data[begin * 3 * n:end * 3 * n].copy()
        table.transform_slice(work)

    leg = strong_scaling_leg(batch, world, rank, run_shard, steps=2, warmup=1, sync=lambda: None, dist=dist, device="cpu")
    units = torch.tensor([leg["units_this_rank"]], dtype=torch.int64)
    dist.all_reduce(units)
    leg_ok = (int(units.item()) == batch and leg["scaling"] == "strong" and leg["batch_total"] == batch and
              len(seen) == 3 and all(r == (b, e) for r in seen) and abs(leg["value"] - batch * 2 / leg["seconds"]) < 1e-6)
    # gather the shards (test-only collective) and compare with the unsharded transform
    parts = [None] * world
    dist.all_gather_object(parts, (b, e, mine))
    # the bench line's `dist` object: what the process group itself reports
    from primus_fhe_amd.shard import device_identity, dist_evidence
    ev = dist_evidence(dist, device_identity(None))
    dist.destroy_process_group()
    if rank == 0:
        full = data.copy()
        table.transform_slice(full)
        got = np.concatenate([p[2] for p in sorted(parts, key=lambda t: t[0])])
        q.put((np.array_equal(got, full) and leg_ok, [(p[0], p[1]) for p in parts], dt, len(calls), ev))


def test_shard_range_properties():
    from primus_fhe_amd.shard import shard_range
    for total in (0, 1, 7, 8, 4096, 8191):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_gloo_sharding_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, ranges, dt, ncalls, ev = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert sorted(ranges) == [(0, 4), (4, 7)]
    assert dt > 0 and ncalls == 3  # 1 warm-up + exactly 2 timed steps
    # the `dist` object of the bench line: backend and world size from the process group, one identity per rank
    assert ev["backend"] == "gloo" and ev["world_size"] == 2 and ev["distinct_devices"] == 2
    assert sorted(d["rank"] for d in ev["devices"]) == [0, 1] and len({d["pid"] for d in ev["devices"]}) == 2


def test_job_seed_makes_shard_input_position_only():
    """A shard's synthetic input filled with job_seed(seed, offset) at word 0 is the job's input at `offset`."""
    from golden_inputs import fill_uniform_words
    from primus_fhe_amd.shard import job_seed
    moduli, n, unit = [2305843009211596801, 2305843009210023937, 97], 16, 2 * 3 * 16
    job = fill_uniform_words(0x5EED000000000005, 0, 7 * unit, moduli, n)
    for b, e in ((0, 4), (4, 7), (2, 3)):
        shard = fill_uniform_words(job_seed(0x5EED000000000005, b * unit), 0, (e - b) * unit, moduli, n)
        assert np.array_equal(shard, job[b * unit:e * unit])


def test_bench_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` with no launcher environment starts N rank processes itself:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* per child, the same argv, only rank 0's stdout relayed."""
    import subprocess
    import types

    import bench
    started = []

    class FakeProc:
        def __init__(self, argv, env=None, stdout=None):
            started.append((argv, env, stdout))

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--one-device", "--steps", "2"])
    args = types.SimpleNamespace(gpus=4, one_device=True, master_port=0)
    assert bench.spawn_ranks(args) == 0
    assert len(started) == 4
    ports = {env["MASTER_PORT"] for _, env, _ in started}
    assert len(ports) == 1
    for r, (argv, env, stdout) in enumerate(started):
        assert argv[1].endswith("bench.py") and argv[2:] == ["--gpus", "4", "--one-device", "--steps", "2"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "4", "127.0.0.1")
        assert (stdout is None) == (r == 0)


def test_bench_rejects_world_size_mismatch(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "WORLD_SIZE" in str(ei.value)


def test_bench_gpus_defaults_to_the_launchers_world_size(monkeypatch):
    """`torchrun --nproc-per-node N bench.py` without --gpus: the launcher's WORLD_SIZE is the number of GPUs
    (ADVICE r3); without a launcher the default is one."""
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.parse_args().gpus == 4
    monkeypatch.delenv("WORLD_SIZE")
    assert bench.parse_args().gpus == 1
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--skip-device-check"])
    a = bench.parse_args()
    assert a.gpus == 8 and a.skip_device_check


def test_traffic_figures_carry_their_provenance():
    """roofline.traffic is read from a committed profile: the line names the file, its git blob id and the SHA-256 of the
    machine code of every kernel it was taken on, and is null when the library being timed holds other code (VERDICT r3
    item 7, r4 item 10)."""
    import subprocess

    import bench
    import primus_fhe_amd as p
    if not os.path.exists(p.library_path()):
        pytest.skip("libpfhe_hip.so not built")
    from primus_fhe_amd._codeobj import kernel_resources
    res = kernel_resources(p.library_path())
    assert res["ntt_pipe_fwd_kernel<PmArith, 12>"]["vgpr"] <= 128 and res["ntt_pipe_fwd_kernel<PmArith, 12>"]["scratch"] == 0
    path = os.path.join(ROOT, "profiles", "r03_d_rocprof.json")   # (any committed file: only its blob id is read here)
    blob = subprocess.run(["git", "hash-object", path], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    assert bench.git_blob_hash(path) == blob
    k = "ntt_pipe_fwd_kernel<PmArith, 12>"
    from primus_fhe_amd._codeobj import kernel_code_hashes
    code = kernel_code_hashes(p.library_path())
    assert set(code) == set(res) and all(len(h) == 16 for h in code.values())   # every kernel of the library has a hash
    built = bench.built_code_hash(k)
    assert built == code[k] and bench.built_vgprs(k) == res[k]["vgpr"]
    prov, ok = bench.provenance(path, {k: built})
    assert ok and prov["profile_git_blob"] == blob and prov["kernels_checked"][k]["built_code_sha256"] == built
    # the same registers, other code: stale (VERDICT r4 item 10 — a kernel can change its traffic and keep its registers)
    prov, ok = bench.provenance(path, {k: "0" * 16})
    assert not ok and not prov["profile_matches_build"]
    assert not bench.provenance(path, {k: None})[1]   # a profile that names no code hash (rounds 1-4) is unverifiable
    assert not bench.provenance(path, {})[1]
    assert not bench.provenance(path, {"no_such_kernel": built})[1]
    # the newest committed profiles: their traffic is reported exactly when they describe the kernels of this tree
    t = bench.pmc_traffic("ntt_pipe_fwd_kernel", 4096)
    assert t is not None
    if t["provenance"]["profile_matches_build"]:
        assert 0.9e9 < t["bytes_per_launch"] < 1.2e9, t     # two passes over a 256 MiB tile
    else:
        assert t["bytes_per_launch"] is None
    e = bench.extprod_traffic()
    assert e is not None
    if e["provenance"]["profile_matches_build"]:
        assert 6 * 96 * 65536 < e["bytes_per_product"] < 12 * 96 * 65536, e
    else:
        assert e["bytes_per_product"] is None


def test_bench_spawner_ends_the_other_ranks_when_one_dies(monkeypatch):
    """A rank that exits non-zero must not leave its peers waiting at a barrier: the launcher terminates them (by handle)
    and returns the failing code."""
    import subprocess
    import types

    import bench

    class P:
        def __init__(self, rc):
            self.rc, self.terminated = rc, False

        def poll(self):
            return self.rc

        def terminate(self):
            self.terminated = True
            self.rc = -15

        def wait(self, timeout=None):
            return self.rc

        def kill(self):
            self.rc = -9

    procs = [P(None), P(3), P(None)]
    it = iter(procs)
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: next(it))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--one-device"])
    rc = bench.spawn_ranks(types.SimpleNamespace(gpus=3, one_device=True, master_port=0))
    assert rc == 15 or rc == 3
    assert procs[0].terminated and procs[2].terminated and not procs[1].terminated


def test_bench_line_fields_of_round_6():
    """the line carries config 3 with a per-element multiplicand beside the shared one, a device-copy
    bandwidth measured in the run (`roofline.peak_measured`, `frac_of_measured`), the prime shape and the generic-prime
    roofline, roofline objects for the inverse / N = 2^14 / u32 legs, the u32 external product, and `oracle_pin`.  Checked on
    the helpers (CPU) and on the newest bench line committed under profiles/ (what the GPU box printed)."""
    import glob
    import json

    import bench
    pin = bench.oracle_pin()
    have = os.path.exists(os.path.join(ROOT, "tests", "golden", "reference_digests.json"))
    assert pin.startswith("pinned" if have else "unpinned")
    r = bench.leg_roofline("k<A, 1>", 4, 2.0, 8e9, {"bytes_per_launch": 4e9, "source": "x", "method": "m", "provenance": {}}, 6000.0)
    assert r["avg_launch_ms"] == 0.5 and r["achieved"] == 4000.0 and r["frac"] == 0.5 and abs(r["frac_of_measured"] - 2 / 3) < 1e-12
    assert r["algorithmic_bytes_per_launch"] == 2e9 and r["traffic_vs_algorithmic"] == 2.0
    assert bench.leg_roofline("k", 1, 1.0, 1e9, None)["traffic"] is None
    t = bench.profile_traffic("ntt_pipe_inv_kernel", {0: "PmArith", 2: "false"})
    assert t is not None and "profile_matches_build" in t["provenance"]
    assert bench.profile_traffic("no_such_kernel", {}) is None
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*_bench.json")))
    if not lines:
        pytest.skip("no round-6 bench line committed yet")
    b = json.load(open(lines[-1]))
    assert b["oracle_pin"].startswith(("pinned", "unpinned"))
    assert "pseudo-Mersenne" in b["config"]["prime_shape"]
    assert b["device_copy"]["GBps"] > 3000
    for obj in (b["roofline"], b["roofline_generic"], b["intt"]["roofline"], b["ntt_2p14"]["forward"]["roofline"],
                b["ntt_2p14"]["inverse"]["roofline"], b["ntt_u32"]["roofline"], b["ntt_u32"]["inverse"]["roofline"],
                b["polymul"]["roofline"], b["polymul_per_element"]["roofline"], b["external_product"]["roofline"],
                b["external_product_u32"]["roofline"]):
        assert obj["bound"] == "hbm" and 0 < obj["frac"] < 1 and obj["peak_measured"] == b["device_copy"]["GBps"]
        assert abs(obj["frac_of_measured"] * obj["peak_measured"] - obj["frac"] * obj["peak"]) < 1e-6 * obj["peak"]
    assert b["polymul_per_element"]["algorithmic_bytes_per_product"] == 72 * 65536
    assert b["polymul"]["algorithmic_bytes_per_product"] == 48 * 65536
    assert b["roofline_generic"]["frac"] < b["roofline"]["frac"]
