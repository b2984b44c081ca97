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
