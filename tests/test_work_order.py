"""Host logic (CPU): the causal work order of the 128-row attention kernels -- which (head, query block) item workgroup blockIdx
takes.  The mapping and the launcher's plan live in csrc/sage_work_order.h as host + device code; the C ABI's sage_debug_work_* entry
points run exactly that code on the host, so this file checks the shipped functions (no restatement): every item exactly once for
any head count / block count / group size, the eight XCDs get equal weight, single-round grids pair long with short blocks on a CU,
the group-size rule on the BASELINE shapes.  On the GPU, tests/test_gpu_soak.py::test_causal_work_order_does_not_change_a_bit pins
that the kernels' outputs are bit-equal under every order.  The reference leaves the order to the hardware: blockIdx.x = query
block, ascending (csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:720-738)."""
import ctypes

import pytest

from sageattention_amd import _cabi


def plan(nheads, nqblk, head_dim=128, forced=-1, pv_fp8=True, kv_len=None):
    lib = _cabi.load()
    g, f, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    grid = lib.sage_debug_work_order_plan(nheads, nqblk, 128 * nqblk if kv_len is None else kv_len, head_dim, int(pv_fp8), forced,
                                          ctypes.byref(g), ctypes.byref(f), ctypes.byref(l))
    assert grid > 0
    return g.value, l.value, f.value, grid


def items(nheads, nqblk, grp, left, fold, nwg):
    lib = _cabi.load()
    h, r = ctypes.c_int(), ctypes.c_int()
    out = []
    for bid in range(nwg):
        rc = lib.sage_debug_work_item(bid, nwg, nheads, nqblk, grp, fold, left, ctypes.byref(h), ctypes.byref(r))
        assert rc in (0, 1)
        out.append((h.value, r.value) if rc == 1 else None)
    return out


@pytest.mark.parametrize("head_dim", [128, 64])
@pytest.mark.parametrize("forced", [-1, 1, 3, 64])
def test_every_item_exactly_once(head_dim, forced):
    for nheads in list(range(1, 19)) + [28, 32, 60, 64]:
        for nqblk in [2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64, 100]:
            grp, left, fold, nwg = plan(nheads, nqblk, head_dim, forced)
            got = [it for it in items(nheads, nqblk, grp, left, fold, nwg) if it is not None]
            assert len(got) == len(set(got)) == nheads * nqblk, (nheads, nqblk, forced)
            assert all(0 <= h < nheads and 0 <= r < nqblk for h, r in got)


def test_head_major_order_is_the_contiguous_split():
    """group 0 (non-causal, masked, split-KV launches; SAGE_ORDER_GROUP=0): XCD x takes a contiguous run of the head-major list."""
    nheads, nqblk = 12, 10
    got = items(nheads, nqblk, 0, 0, 0, nheads * nqblk)
    assert sorted(got) == [(h, r) for h in range(nheads) for r in range(nqblk)]
    for x in range(8):
        run = [h * nqblk + r for h, r in got[x::8]]
        assert run == list(range(run[0], run[0] + len(run)))


def test_xcds_get_equal_weight_and_single_rounds_are_folded():
    for nheads, nqblk in [(64, 64), (28, 64), (12, 64), (4, 128), (60, 32), (7, 48), (9, 256)]:
        grp, left, fold, nwg = plan(nheads, nqblk)
        w = [0] * 8
        for bid, it in enumerate(items(nheads, nqblk, grp, left, fold, nwg)):
            if it is not None:
                w[bid & 7] += 2 * (nqblk - it[1])         # 64-key tiles of the block (rank 0 = the longest)
        assert max(w) <= 1.02 * sum(w) / 8, (nheads, nqblk, w)
    # one round: in-XCD indices i and i + 32 share a CU (tools/microbench/ubench7_dispatch.hip) -> long + short
    for nheads, nqblk in [(64, 8), (8, 48), (4, 128)]:
        grp, left, fold, nwg = plan(nheads, nqblk)
        assert fold and nwg // 8 <= 64
        its = items(nheads, nqblk, grp, left, fold, nwg)
        sums = [(nqblk - its[i * 8][1]) + (nqblk - its[(i + 32) * 8][1]) for i in range(nwg // 8 - 32)]
        assert max(sums) - min(sums) <= 1, (nheads, nqblk, sums)


def test_group_size_rule_on_the_baseline_shapes():
    assert [plan(64, n)[0] for n in (8, 16, 32, 64, 128, 256)] == [8, 8, 4, 2, 1, 1]          # the N = 1k .. 32k sweep at B*H = 64
    assert plan(64, 32, pv_fp8=False)[0] == 4 and plan(64, 64, head_dim=64)[0] == 4              # C2; D = 64 FP8 at N = 8k
    assert plan(64, 64)[2] == 0 and plan(64, 8)[2] == 1 and plan(64, 8, head_dim=64)[2] == 0     # fold: single-round grids, two workgroups per CU
    lib = _cabi.load()
    assert lib.sage_debug_work_item(5, 4, 1, 4, 1, 0, 0, None, None) == -1


# ---- packed (varlen) batches: the device-built work list, run on the host through sage_debug_varlen_items ------------------------------
def varlen_items(lq, lk, causal, hq, hkv, head_dim=128):
    import numpy as np
    lib = _cabi.load()
    lqa, lka = np.asarray(lq, np.int32), np.asarray(lk, np.int32)
    n = int(((lqa + 127) // 128).sum())
    out = np.zeros((max(n, 1), 2), np.int32)
    hdr = np.zeros(8, np.int32)
    grid = lib.sage_debug_varlen_items(lqa.ctypes.data_as(ctypes.c_void_p), lka.ctypes.data_as(ctypes.c_void_p), len(lq), int(causal), hq, hkv,
                                       head_dim, 0, out.ctypes.data_as(ctypes.c_void_p), max(n, 1), hdr.ctypes.data_as(ctypes.c_void_p))
    assert grid >= 0 and hdr[0] == n
    return [tuple(x) for x in out[:n]], hdr[:4].tolist(), grid


def _weight(lk, j, causal):
    ntk = (lk + 63) // 64
    return min(ntk, 2 * j + 2) if causal else ntk


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_work_list_is_every_query_block_once_heaviest_first(causal):
    import random
    rnd = random.Random(3)
    cases = [([256, 512, 1000, 1024, 2048, 4096, 8192, 16384],) * 2, ([1, 127, 128, 129, 700, 64, 1000], [5, 64, 200, 77, 1000, 640, 3]),
             ([0, 300, 0], [10, 300, 5]), ([128] * 40,) * 2]
    cases += [([rnd.randrange(0, 3000) for _ in range(n)], [rnd.randrange(0, 3000) for _ in range(n)]) for n in (1, 2, 9, 33, 200)]
    for lq, lk in cases:
        its, (nitems, grp, fold, left), grid = varlen_items(lq, lk, causal, 32, 8)
        want = [(s, j) for s in range(len(lq)) for j in range((lq[s] + 127) // 128)]
        assert sorted(its) == want                                                       # a permutation of the blocks that exist
        w = [_weight(lk[s], j, causal) for s, j in its]
        assert all(a >= b for a, b in zip(w, w[1:]))                                     # heaviest first
        # ties: sequence index ascending, then the later block first (a total order: the device and the host agree on it)
        assert all((w[i] > w[i + 1]) or (its[i][0] < its[i + 1][0]) or (its[i][0] == its[i + 1][0] and its[i][1] > its[i + 1][1])
                   for i in range(len(its) - 1))


def test_varlen_launch_deals_every_head_item_pair_once_and_keeps_gqa_groups_on_one_xcd():
    for lq, hq, hkv, head_dim in [([256, 512, 1000, 1024, 2048, 4096, 8192, 16384], 32, 8, 128), ([700, 64, 1000, 3], 12, 4, 64),
                                  ([5000], 4, 1, 128), ([128] * 9, 8, 8, 128), ([300, 200], 28, 4, 128), ([100] * 50, 64, 8, 128)]:
        for causal in (False, True):
            its, (nitems, grp, fold, left), grid = varlen_items(lq, lq, causal, hq, hkv, head_dim)
            assert grid == 8 * (left * ((nitems + 7) // 8) + (hq // 8) * nitems) and left == hq % 8
            gqa = hq // hkv
            assert grp >= 1 and (hq // 8 == 0 or grp % gqa == 0 or grp == hq // 8)       # whole GQA groups (or every head the XCD owns)
            got = [it for it in items(hq, nitems, grp, left, fold, grid) if it is not None]
            assert len(got) == len(set(got)) == hq * nitems
            # a head that an XCD owns stays on it: its K/V (all sequences) streams through one L2
            owner = {}
            for bid, it in enumerate(items(hq, nitems, grp, left, fold, grid)):
                if it is not None and it[0] >= left:
                    assert owner.setdefault(it[0], bid & 7) == bid & 7
            if hq >= 8 and (hq // 8) % gqa == 0 and left == 0:
                for h, x in owner.items():
                    assert owner[(h // gqa) * gqa] == x                                    # the query heads of one kv head share the XCD


def test_c4_work_list_shape():
    """BASELINE.json configs[3]: 262 query blocks in all, 8384 workgroups (the grid sized by max_seqlen_q launched 32768)."""
    lens = [256, 512, 1000, 1024, 2048, 4096, 8192, 16384]
    its, (nitems, grp, fold, left), grid = varlen_items(lens, lens, True, 32, 8)
    assert nitems == 262 and grid == 8 * 4 * 262 and left == 0 and grp == 4 and fold == 0
    assert its[0] == (7, 127) and its[1] == (7, 126)


def test_ticket_queues_of_the_persistent_launch_partition_the_logical_grid():
    """The persistent launch (csrc/sage_attn_kernel.h, `PERS_OK` kernels) deals the logical workgroup indices 0 .. nwg - 1 into 32 queues: index
    i = 32 k + 8 s + x belongs to XCD x = i & 7 (the work order's locality) and sub-queue s = (i >> 3) & 3, ticket k.  Restated here: the queues
    partition the grid for every nwg (also one that is no multiple of 32), the count formula the kernel uses for "how much is left" is the size
    of the queue, and the first round -- a launch of G workgroups, G a multiple of 32, takes i = blockIdx.x without a ticket -- is exactly the
    tickets k < G / 32 of every queue, so that the counters can start at zero."""
    for nwg in (8, 24, 32, 40, 512, 6144, 8384, 13344, 13344 + 8):
        seen = []
        for q in range(32):
            x, s = q >> 2, q & 3
            s8x = 8 * s + x
            cnt = (nwg - s8x + 31) >> 5 if nwg > s8x else 0            # the kernel's formula
            idx = [32 * k + s8x for k in range(cnt + 2) if 32 * k + s8x < nwg]
            assert len(idx) == cnt, (nwg, q)
            assert all((i & 7) == x and ((i >> 3) & 3) == s for i in idx)
            seen += idx
        assert sorted(seen) == list(range(nwg))
    for G in (512, 768):
        first = G >> 5
        for q in range(32):
            s8x = 8 * (q & 3) + (q >> 2)
            assert all(32 * k + s8x < G for k in range(first)) and 32 * first + s8x >= G
