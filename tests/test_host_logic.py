"""CPU: host-side behaviour of the API mirror (error contract of the reference's core.py)."""
import pytest
import torch

import util  # noqa: F401
import sageattention
import sageattention_amd as sa
from sageattention_amd import core, shard


def test_public_names_match_reference():
    # /root/reference/sageattention/__init__.py:1-5
    names = {"sageattn", "sageattn_varlen", "sageattn_qk_int8_pv_fp16_triton", "sageattn_qk_int8_pv_fp16_cuda",
             "sageattn_qk_int8_pv_fp8_cuda", "sageattn_qk_int8_pv_fp8_cuda_sm90"}
    assert names <= set(dir(sa)) and names <= set(dir(sageattention))
    assert sageattention.sageattn is sa.sageattn


def test_reference_module_paths_and_signatures():
    """The reference's module paths (sageattention/core.py, quant.py, triton/*.py) resolve to the gfx950 implementation with the reference's
    parameter names and defaults; the kernel-level entry points reject CPU tensors instead of emulating."""
    import inspect
    import sageattention.core as rc
    import sageattention.quant as rq
    from sageattention.triton import (attn_qk_int8_per_block, attn_qk_int8_per_block_causal, attn_qk_int8_block_varlen,
                                      attn_qk_int8_per_block_causal_varlen, quant_per_block, quant_per_block_varlen, quant_per_thread)
    assert rc.sageattn is sa.sageattn and rc.sageattn_qk_int8_pv_fp8_cuda_sm90 is sa.sageattn_qk_int8_pv_fp8_cuda_sm90
    par = lambda f: list(inspect.signature(f).parameters)
    assert par(attn_qk_int8_per_block.forward) == ["q", "k", "v", "q_scale", "k_scale", "tensor_layout", "attn_mask", "output_dtype", "return_lse"]   # attn_qk_int8_per_block.py:130
    assert par(attn_qk_int8_per_block_causal.forward) == ["q", "k", "v", "q_scale", "k_scale", "tensor_layout", "output_dtype", "return_lse"]       # _causal.py:124
    vl = ["q", "k", "v", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "q_scale", "k_scale", "cu_seqlens_q_scale", "cu_seqlens_k_scale", "output_dtype"]
    assert par(attn_qk_int8_block_varlen.forward) == vl and par(attn_qk_int8_per_block_causal_varlen.forward) == vl                                  # :123, :138
    assert par(quant_per_block.per_block_int8) == ["q", "k", "km", "BLKQ", "BLKK", "sm_scale", "tensor_layout"] == par(rq.per_block_int8)          # quant_per_block.py:49, quant.py:22
    assert par(quant_per_block_varlen.per_block_int8) == ["q", "k", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "BLKQ", "BLKK", "sm_scale"]
    assert par(quant_per_thread.per_thread_int8) == ["q", "k", "km", "BLKQ", "WARPQ", "BLKK", "WARPK", "sm_scale", "tensor_layout"]                 # quant_per_thread.py:154
    assert par(rq.per_warp_int8) == ["q", "k", "km", "BLKQ", "WARPQ", "BLKK", "tensor_layout"] and par(rq.sub_mean) == ["v", "tensor_layout"]
    assert inspect.signature(rq.per_channel_fp8).parameters["smooth_v"].default is True and par(rq.per_channel_fp8) == ["v", "tensor_layout", "scale_max", "smooth_v"]   # quant.py:224-229
    q8 = torch.zeros(1, 1, 8, 64, dtype=torch.int8)
    with pytest.raises(AssertionError):
        attn_qk_int8_per_block.forward(q8, q8, torch.zeros(1, 1, 8, 64, dtype=torch.float16), torch.ones(1, 1, 1), torch.ones(1, 1, 1))


def test_cpu_tensors_are_rejected_not_emulated():
    q = torch.zeros(1, 1, 8, 64, dtype=torch.float16)
    with pytest.raises(ValueError, match="Unsupported architecture"):
        sa.sageattn(q, q, q)
    for fn in (sa.sageattn_qk_int8_pv_fp8_cuda, sa.sageattn_qk_int8_pv_fp16_cuda, sa.sageattn_qk_int8_pv_fp16_triton):
        with pytest.raises(AssertionError, match="must be on cuda"):
            fn(q, q, q)


def test_pad_head_dim_rules():
    for d, want in ((32, 64), (64, 64), (96, 128), (128, 128)):
        q = torch.zeros(1, 1, 4, d, dtype=torch.float16)
        qq, kk, vv, og = core._pad_head_dim(q, q, q)
        assert qq.shape[-1] == want and og == d
    with pytest.raises(ValueError, match="Unsupported head_dim"):
        core._pad_head_dim(*(torch.zeros(1, 1, 4, 160, dtype=torch.float16),) * 3)


def test_lse_correction_shapes_and_values():
    q = torch.randn(2, 4, 10, 64).half(); k = torch.randn(2, 2, 12, 64).half()
    km = k.mean(dim=2, keepdim=True)
    corr = core._lse_correction(q, km, "HND")
    assert corr.shape == (2, 4, 10) and corr.dtype == torch.float32
    want = torch.einsum("bhld,bhd->bhl", q.float(), km[:, :, 0].float().repeat_interleave(2, 1))
    assert (corr - want).abs().max() < 2e-2
    corr_n = core._lse_correction(q.transpose(1, 2), km.transpose(1, 2), "NHD")
    assert corr_n.shape == (2, 4, 10) and torch.equal(corr_n, corr)
    assert core._smooth_k(q, k, "HND", False, True) == (None, None)


def test_shard_units_partition():
    for B, H, g in ((2, 32, 1), (2, 32, 4), (3, 8, 8), (1, 48, 1)):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard.shard_range(B * H // g, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(B * H // g))
    sizes = [shard.shard_range(64, r, 8) for r in range(8)]
    assert all(hi - lo == 8 for lo, hi in sizes)


def test_custom_ops_registered_with_fake_impl():
    """sm80/sm89/sm90_compile.py convention: custom_op + register_fake so torch.compile can trace (shape-only here)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from sageattention_amd import ops  # noqa: F401
    with FakeTensorMode():
        dev = "cuda"
        q = torch.empty(2, 300, 4, 128, dtype=torch.int8, device=dev)       # NHD
        k = torch.empty(2, 300, 2, 128, dtype=torch.int8, device=dev)
        v = torch.empty(2, 2, 5, 128, 64, dtype=torch.uint8, device=dev)
        o = torch.empty(2, 300, 4, 128, dtype=torch.bfloat16, device=dev)
        qs = torch.empty(2, 4, 96, device=dev); ks = torch.empty(2, 2, 20, device=dev); vs = torch.empty(2, 2, 128, device=dev)
        lse = torch.ops.sageattention_gfx950.qk_int8_sv_f8_attn(q, k, v, o, qs, ks, vs, None, 0, 1, 3, 32, 0.1, 1, 1)
        assert lse.shape == (2, 4, 300) and lse.dtype == torch.float32
        lse0 = torch.ops.sageattention_gfx950.qk_int8_sv_f16_attn(q, k, v, o, qs, ks, None, 0, 0, 2, 32, 0.1, 0, 0)
        assert lse0.numel() == 0


def test_split_kv_planner():
    """Host logic of the split-KV route (core._split_kv_plan): when a call is split and into how many chunks."""
    plan = core._split_kv_plan
    assert plan(1, 32, 128, 32768, False, None) == 16          # 32 workgroups, 512 key tiles -> 16 chunks of 32 tiles
    assert plan(1, 16, 1024, 16384, False, None) == 4          # 128 workgroups, 256 tiles: target 6 -> largest divisor of 256 below it
    assert plan(2, 32, 8192, 8192, False, None) == 0           # the grid already fills the chip
    assert plan(1, 32, 128, 32768, True, None) == 0            # causal with Lq != Lk is not split
    assert plan(1, 8, 8192, 8192, True, None) == 0             # causal: split only on request (measured slower, see core.py)
    assert plan(1, 8, 8192, 8192, True, 4) == 4
    assert plan(1, 8, 128, 32768 + 32, False, None) == 0       # ragged key range
    assert plan(1, 8, 128, 2048, False, None) == 0             # short key range
    assert plan(1, 8, 128, 4096, False, 0) == 0                # switched off
    assert plan(1, 8, 128, 4096, False, 8) == 8                # forced
    with pytest.raises(ValueError):
        plan(1, 8, 128, 4096, False, 7)                        # does not divide the 64 key tiles
    # FP8 PV (auto_default=False): no split unless asked for -- the default FP8 route has ONE schedule, the one its parity gate names
    assert plan(1, 32, 128, 32768, False, None, auto_default=False) == 0
    assert plan(1, 32, 128, 32768, False, "auto", auto_default=False) == 16
    assert plan(1, 32, 128, 32768, False, 8, auto_default=False) == 8
    assert plan(1, 32, 128, 32768, False, "auto") == 16
    with pytest.raises(ValueError):
        plan(1, 8, 128, 4096, False, "yes")


def test_bench_rank_units_partition_the_global_problem():
    """bench.py gives rank r the shard_bh slice of the global batch: the slices of all ranks are disjoint and cover it."""
    B, Hq, Hkv, L, D = 3, 4, 2, 5, 8
    q = torch.arange(B * Hq * L * D, dtype=torch.float32).reshape(B, Hq, L, D)
    k = torch.arange(B * Hkv * L * D, dtype=torch.float32).reshape(B, Hkv, L, D)
    for world in (1, 2, 4, 5):
        seen_q, seen_k = [], []
        for r in range(world):
            qs, ks, vs, (lo, hi) = shard.shard_bh(q, k, k, r, world)
            assert qs.shape == (1, (hi - lo) * (Hq // Hkv), L, D) and ks.shape == (1, hi - lo, L, D)
            seen_q.append(qs.reshape(-1, L, D)); seen_k.append(ks.reshape(-1, L, D))
        assert torch.equal(torch.cat(seen_q), q.reshape(-1, L, D)) and torch.equal(torch.cat(seen_k), k.reshape(-1, L, D))


def test_prepass_route_choice(monkeypatch):
    """Which pre-pass a dense call takes (core._fused_prepass_wanted): the one-launch kernel unless it cannot take the head
    (too long for the in-launch barrier / no device) or very many heads of <= 256 keys would leave its 512-row slabs half empty."""
    meta = lambda *s: torch.empty(*s, device="meta", dtype=torch.float16)
    monkeypatch.setattr(core, "prepass_fused_ok", lambda k, layout="HND": core._dims(k, layout)[2] <= 32768)
    want = core._fused_prepass_wanted
    assert want(meta(2, 32, 8192, 128), "HND", None)
    assert want(meta(2, 8192, 32, 128), "NHD", None)
    assert want(meta(1, 4, 200, 64), "HND", None)              # few heads: one launch instead of six
    assert want(meta(8, 32, 257, 64), "HND", None)
    assert not want(meta(64, 16, 256, 64), "HND", None)        # 1024 half-empty slabs
    assert not want(meta(1, 2, 40000, 64), "HND", None)        # beyond the barrier's reach ...
    assert not want(meta(1, 2, 40000, 64), "HND", True)        # ... whatever the caller asks for
    assert not want(meta(2, 32, 8192, 128), "HND", False)
    assert want(meta(64, 16, 256, 64), "HND", True)
    # without a GPU the C ABI reports a maximum head length of 0: never the fused route, never a crash
    monkeypatch.undo()
    assert not core.prepass_fused_ok(torch.empty(1, 1, 64, 64, dtype=torch.float16)) or torch.cuda.is_available()
