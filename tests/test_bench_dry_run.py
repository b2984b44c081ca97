"""CPU: the multi-GPU bookkeeping of bench.py without a GPU and without RCCL (a `--gpus N` run has never touched RCCL here: gpurun boxes have
one GPU).  `bench.dry_run` walks the N-rank split on the host; `bench.unit_inputs` -- what a rank really generates -- must be
`shard.shard_bh` of the global tensors, bit for bit, so that the ranks of one run hold the shards of ONE problem without ever building it."""
import json
import os
import subprocess
import sys

import pytest
import torch

import util  # noqa: F401  (sys.path)
import bench
from sageattention_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["c3", "c2", "c5", "h28"])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_dry_run_partitions_the_units_and_accounts_every_flop(name, world):
    cfg = bench.CONFIGS[name]
    weak = bench.dry_run(cfg, world, weak=True)
    assert weak["units_total"] == cfg["B"] * cfg["Hkv"] * world and len(weak["ranks"]) == world
    per = cfg["B"] * cfg["Hkv"]
    for r in weak["ranks"]:
        assert r["units"] == [r["rank"] * per, (r["rank"] + 1) * per]
        assert r["q_shape"] == [1, per * (cfg["H"] // cfg["Hkv"]), cfg["N"], cfg["D"]]
        # what main() computes: value = (FLOPs of this rank's folded problem) * world / time == global FLOPs / time
        assert r["flops"] * world == pytest.approx(weak["global_flops"], rel=1e-12)
        assert r["flops"] == pytest.approx(bench.flops(cfg), rel=1e-12)          # per-rank work is the single-GPU workload
    strong = bench.dry_run(cfg, world, weak=False)                                # the C5 replay's split: one global call
    assert sum(r["flops"] for r in strong["ranks"]) == pytest.approx(bench.flops(cfg), rel=1e-12)
    assert strong["ranks"][0]["units"][0] == 0 and strong["ranks"][-1]["units"][1] == cfg["B"] * cfg["Hkv"]
    sizes = [r["units"][1] - r["units"][0] for r in strong["ranks"]]
    assert max(sizes) - min(sizes) <= 1


def test_rank_units_are_shard_bh_of_the_global_problem():
    cfg = dict(B=2, H=6, Hkv=3, N=40, D=16, causal=True, pv="fp8", dtype="bf16", workload="tiny")
    cpu = torch.device("cpu")
    for world in (1, 2, 3, 4):
        q, k, v = bench.make_inputs(cfg, cpu, 7, batch_mult=world)               # the global problem (only this test ever builds it)
        assert q.shape == (2 * world, 6, 40, 16) and k.shape == (2 * world, 3, 40, 16)
        for rank in range(world):
            qs, ks, vs, (lo, hi) = shard.shard_bh(q, k, v, rank, world)
            qr, kr, vr, units = bench.rank_inputs(cfg, cpu, 7, rank, world, weak=True)
            assert units == (lo, hi) and torch.equal(qs, qr) and torch.equal(ks, kr) and torch.equal(vs, vr)
    q, k, v = bench.make_inputs(cfg, cpu, 7)
    for world in (2, 4):                                                         # strong split (the replay): uneven unit counts
        got = [bench.rank_inputs(cfg, cpu, 7, r, world, weak=False) for r in range(world)]
        assert torch.equal(torch.cat([g[0] for g in got], dim=1).view_as(q), q) and torch.equal(torch.cat([g[2] for g in got], dim=1).view_as(v), v)
    assert not torch.equal(bench.make_inputs(cfg, cpu, 8)[0], q)                 # the seed matters


def test_dry_run_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", "8", "--config", "c5", "--replay"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])["dry_run"]
    assert d["world"] == 8 and d["scaling"] == "strong" and [r["units"][1] - r["units"][0] for r in d["ranks"]] == [12] * 8


def test_rank_records_are_checked():
    """`ranks_seen` of a --gpus N run: every rank present once, and over RCCL one distinct device per rank."""
    recs = [dict(rank=r, local_rank=r, device=r, device_count=8, name="MI355X", pci_bus_id=None, uuid=f"GPU-{r:02d}", pid=100 + r) for r in range(8)]
    seen = bench.check_ranks(recs, 8, "nccl")
    assert seen["world_size"] == 8 and len(seen["ranks"]) == 8 and seen["backend"] == "nccl"
    dup = [dict(r) for r in recs]
    dup[5]["uuid"] = dup[2]["uuid"]
    with pytest.raises(AssertionError, match="share a device"):
        bench.check_ranks(dup, 8, "nccl")
    bench.check_ranks(dup, 8, "gloo")                                  # (the one-GPU gloo dry run shares cuda:0 on purpose)
    with pytest.raises(AssertionError, match="missing"):
        bench.check_ranks(recs[:7] + [recs[0]], 8, "nccl")


def test_dist_check_with_eight_cpu_ranks():
    """The driver's 8-GPU launch line on CPU ranks: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8` with
    --dist-check runs bench.py's process-group set-up, rank records, unit split, barrier and MAX-over-ranks time reduction over gloo."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dist-check"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dist_check"] and d["n_gpus"] == 8 and d["ranks_seen"]["world_size"] == 8 and d["ranks_seen"]["backend"] == "gloo"
    assert sorted(r["rank"] for r in d["ranks_seen"]["ranks"]) == list(range(8))
    assert len(d["ms_per_step_per_rank"]) == 8 and d["ms_per_step"] == pytest.approx(max(d["ms_per_step_per_rank"]), rel=1e-3)
    assert d["ms_per_step"] == pytest.approx(10.0 * 1.07, rel=1e-3)                # the slowest (fake) rank decides
    cfg = bench.CONFIGS["c3"]
    assert d["value"] == pytest.approx(bench.flops(cfg) * 8 / (d["ms_per_step"] * 1e-3) / 1e12, rel=1e-3)
