"""CPU, world_size 2 over gloo: the ring schedule of sageattention_amd/ring.py (K/V shards travel as isend/irecv,
partial states are merged by log-sum-exp).  The attention step and the merge are injected fp32 torch references
(the product defaults are HIP-only), so this checks the host logic: every rank's rows must equal full attention
over the concatenated sequence, causal and not, HND and NHD."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from sageattention_amd import ring


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, layout, causal, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    B, Hq, Hkv, L, D = 2, 4, 2, 96 * world, 64
    q, k, v = torch.randn(B, Hq, L, D).half(), torch.randn(B, Hkv, L, D).half(), torch.randn(B, Hkv, L, D).half()
    lo, hi = rank * L // world, (rank + 1) * L // world
    qs, ks, vs = (x[:, :, lo:hi] for x in (q, k, v))
    if layout == "NHD":
        qs, ks, vs = (x.transpose(1, 2).contiguous() for x in (qs, ks, vs))
    o, lse = ring.ring_sageattn(qs, ks, vs, tensor_layout=layout, is_causal=causal, return_lse=True,
                                attn_fn=util.attn_with_lse_f32, merge_fn=util.merge_states_torch)
    full_o, full_lse = util.attn_with_lse_f32(q, k, v, is_causal=causal)
    want = full_o[:, :, lo:hi]
    if layout == "NHD":
        want = want.transpose(1, 2)
    assert o.shape == qs.shape and o.dtype == qs.dtype
    assert (o.float() - want.float()).abs().max().item() <= 2e-3, (o.float() - want.float()).abs().max().item()
    assert (lse - full_lse[:, :, lo:hi]).abs().max().item() <= 1e-4
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("layout,causal", [("HND", False), ("HND", True), ("NHD", True)])
def test_two_rank_ring_equals_full_attention(tmp_path, layout, causal):
    mp.spawn(_worker, args=(2, _free_port(), layout, causal, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def _zz_worker(rank, world, port, layout, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    B, Hq, Hkv, L, D = 1, 4, 2, 64 * 2 * world, 64
    q, k, v = torch.randn(B, Hq, L, D).half(), torch.randn(B, Hkv, L, D).half(), torch.randn(B, Hkv, L, D).half()
    full_o, full_lse = util.attn_with_lse_f32(q, k, v, is_causal=True)
    if layout == "NHD":
        q, k, v, full_o = (x.transpose(1, 2).contiguous() for x in (q, k, v, full_o))
    qs, ks, vs = (ring.zigzag_shard(x, rank, world, layout) for x in (q, k, v))
    o, lse = ring.ring_sageattn(qs, ks, vs, tensor_layout=layout, is_causal=True, return_lse=True, shard_order="zigzag",
                                attn_fn=util.attn_with_lse_f32, merge_fn=util.merge_states_torch)
    want = ring.zigzag_shard(full_o, rank, world, layout)
    want_lse = ring.zigzag_shard(full_lse.unsqueeze(-1), rank, world, "HND").squeeze(-1)
    assert o.shape == qs.shape and (o.float() - want.float()).abs().max().item() <= 2e-3
    assert (lse - want_lse).abs().max().item() <= 1e-4
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,layout", [(2, "HND"), (3, "NHD")])
def test_zigzag_ring_equals_full_causal_attention(tmp_path, world, layout):
    mp.spawn(_zz_worker, args=(world, _free_port(), layout, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_zigzag_schedule_is_balanced():
    for w in (1, 2, 4, 8):
        work = []
        for r in range(w):
            tot = 0.0
            for _, _, calls in ring.zigzag_schedule(r, w):
                for qp, kp, mode in calls:
                    nq, nk = (2 if qp == "all" else 1), (2 if kp == "all" else 1)
                    tot += nq * nk * (0.5 if mode == "causal" else 1.0)
            work.append(tot)
        assert max(work) == min(work) == 2.0 * w, work            # 2 c^2 per step on every rank


def test_shard_schedule():
    assert ring.shard_schedule(0, 1, True) == [(0, 0, "causal")]
    assert ring.shard_schedule(2, 4, False) == [(0, 2, "full"), (1, 1, "full"), (2, 0, "full"), (3, 3, "full")]
    assert ring.shard_schedule(1, 4, True) == [(0, 1, "causal"), (1, 0, "full"), (2, 3, "skip"), (3, 2, "skip")]
    for w in (1, 2, 3, 8):          # every rank sees every shard exactly once
        for r in range(w):
            assert sorted(j for _, j, _ in ring.shard_schedule(r, w, False)) == list(range(w))


def test_single_process_ring_is_plain_attention():
    torch.manual_seed(0)
    q, k, v = torch.randn(1, 2, 40, 64).half(), torch.randn(1, 2, 40, 64).half(), torch.randn(1, 2, 40, 64).half()
    o = ring.ring_sageattn(q, k, v, is_causal=True, attn_fn=util.attn_with_lse_f32, merge_fn=util.merge_states_torch)
    want, _ = util.attn_with_lse_f32(q, k, v, is_causal=True)
    assert torch.equal(o, want)
