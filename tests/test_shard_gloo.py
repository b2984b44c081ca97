"""CPU, world_size 2 over gloo: the N>1 path of bench.py is a (batch, kv-head) shard with no data-path
collective.  Each rank computes ITS units with the (CPU) oracle; gathered shards must equal the unsharded
result bit-for-bit, and the bench's timing reduction (MAX over ranks) must work over the process group."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from sageattention_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_bits, k_bits, v_bits, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    q, k, v = (util.from_bits(x, 0) for x in (q_bits, k_bits, v_bits))
    qs, ks, vs, (lo, hi) = shard.shard_bh(q, k, v, rank, world)
    o, _, _ = oracle.sageattn_dense(util.bits(qs), util.bits(ks), util.bits(vs), 0, is_causal=True, pv="f8",
                                    qk_quant_gran="per_warp")
    mine = torch.from_numpy(o.astype(np.int32))                      # [1, units*g, L, D]
    sizes = [None] * world
    dist.all_gather_object(sizes, (lo, hi, mine.shape[1]))          # host-side bookkeeping only
    parts = [torch.empty((1, s[2]) + tuple(mine.shape[2:]), dtype=torch.int32) for s in sizes]
    dist.all_gather(parts, mine)                                    # test harness only, not the data path
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                        # what bench.py does with elapsed time
    assert abs(t.item() - 0.1 * world) < 1e-12
    if rank == 0:
        np.save(out_path, torch.cat(parts, dim=1).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_equals_unsharded(tmp_path, oracle_mod):
    B, Hq, Hkv, L, D = 2, 4, 2, 200, 64
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, Hq, L, D, generator=g).half()
    k = torch.randn(B, Hkv, L, D, generator=g).half()
    v = torch.randn(B, Hkv, L, D, generator=g).half()
    out_path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), util.bits(q), util.bits(k), util.bits(v), out_path), nprocs=2, join=True)
    gathered = np.load(out_path).astype(np.uint16).reshape(B, Hq, L, D)
    full, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 0, is_causal=True, pv="f8",
                                           qk_quant_gran="per_warp")
    assert (gathered == full).all()
