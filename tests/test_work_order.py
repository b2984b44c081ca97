"""Host logic (CPU): the causal work order of the 128-row attention kernels -- which (head, query block) item workgroup blockIdx
takes -- restated from csrc/sage_attn.hip (set_work_order + the kernel's "work item" block) and checked exhaustively: every item
exactly once for any head count / block count / group size, the eight XCDs get equal weight, single-round grids pair long with
short blocks on a CU.  The HIP code itself is pinned by tests/test_gpu_soak.py::test_causal_work_order_does_not_change_a_bit
(outputs bit-equal under every order).  The reference leaves the order to the hardware: blockIdx.x = query block, ascending
(csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:720-738)."""
import pytest


def plan(nheads, nqblk, head_dim=128, forced=-1, pv_fp8=True):
    hpx, left = nheads // 8, nheads % 8
    wg = 3 if head_dim == 64 else 2
    g_bal = (2 * 32 * wg + nqblk) // (nqblk + 1)
    g_l2 = (4 << 20) // (nqblk * 128 * head_dim * (2 if pv_fp8 else 3))      # self-attention: Lk = 128 * nqblk
    grp = forced if forced > 0 else max(g_bal, min(2 * g_bal, g_l2))
    grp = max(1, min(grp, hpx))
    cnt = left * ((nqblk + 7) // 8) + hpx * nqblk
    one_sorted_list = (left == 0 and grp >= hpx) or hpx == 0
    fold = wg == 2 and 32 < cnt <= 64 and one_sorted_list
    return grp, left, fold, 8 * cnt


def item(bid, nwg, nheads, nqblk, grp, left, fold):
    xcd, idx = bid & 7, bid >> 3
    qq = nwg >> 3
    r = qq - 1 - (idx - 32) if (fold and idx >= 32) else idx
    hpx = nheads >> 3
    left_cnt = left * ((nqblk + 7) >> 3)
    if r < left_cnt:
        octet = r // left
        head = r - octet * left
        qrank = 8 * octet + ((7 - xcd) if (octet & 1) else xcd)
        if qrank >= nqblk:
            return None
    else:
        r -= left_cnt
        gsz = grp * nqblk
        gi = r // gsz
        within = r - gi * gsz
        gc = min(hpx - gi * grp, grp)
        qrank = within // gc
        head = left + xcd * hpx + gi * grp + (within - qrank * gc)
    return head, qrank


@pytest.mark.parametrize("head_dim", [128, 64])
@pytest.mark.parametrize("forced", [-1, 1, 3, 64])
def test_every_item_exactly_once(head_dim, forced):
    for nheads in list(range(1, 34)) + [40, 60, 64]:
        for nqblk in list(range(2, 20)) + [31, 32, 33, 64, 100, 128]:
            grp, left, fold, nwg = plan(nheads, nqblk, head_dim, forced)
            seen = set()
            for bid in range(nwg):
                it = item(bid, nwg, nheads, nqblk, grp, left, fold)
                if it is None:
                    continue
                assert it not in seen and 0 <= it[0] < nheads and 0 <= it[1] < nqblk, (nheads, nqblk, forced, bid, it)
                seen.add(it)
            assert len(seen) == nheads * nqblk, (nheads, nqblk, forced)


def test_xcds_get_equal_weight_and_single_rounds_are_folded():
    for nheads, nqblk in [(64, 64), (28, 64), (12, 64), (4, 128), (60, 32), (7, 48), (9, 256)]:
        grp, left, fold, nwg = plan(nheads, nqblk)
        w = [0] * 8
        for bid in range(nwg):
            it = item(bid, nwg, nheads, nqblk, grp, left, fold)
            if it is not None:
                w[bid & 7] += 2 * (nqblk - it[1])         # 64-key tiles of the block (rank 0 = the longest)
        assert max(w) <= 1.02 * sum(w) / 8, (nheads, nqblk, w)
    assert [plan(64, n)[0] for n in (8, 16, 32, 64, 128, 256)] == [8, 8, 4, 2, 1, 1]          # the N = 1k .. 32k sweep at B*H = 64
    assert plan(64, 32, pv_fp8=False)[0] == 4 and plan(64, 64, head_dim=64)[0] == 4              # C2; D = 64 FP8 at N = 8k
    # one round: in-XCD indices i and i + 32 share a CU (tools/microbench/ubench7_dispatch.hip) -> long + short
    for nheads, nqblk in [(64, 8), (8, 48), (4, 128)]:
        grp, left, fold, nwg = plan(nheads, nqblk)
        assert fold and nwg // 8 <= 64
        sums = []
        for i in range(nwg // 8 - 32):
            a = item(i * 8, nwg, nheads, nqblk, grp, left, fold)
            b = item((i + 32) * 8, nwg, nheads, nqblk, grp, left, fold)
            sums.append((nqblk - a[1]) + (nqblk - b[1]))
        assert max(sums) - min(sums) <= 1, (nheads, nqblk, sums)
