#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (runs only where
/root/reference exists, i.e. in the build container -- never on the GPU box).

The reference's Triton kernels are executed unmodified on CPU with TRITON_INTERPRET=1;
its modules are loaded by file path because `import sageattention` needs the compiled
CUDA `_fused` extension (sageattention/quant.py:20).  The host glue of
`sageattn_qk_int8_pv_fp16_triton` / `sageattn_varlen` (core.py:260-331, :399-448: pad,
K mean, v->fp16, LSE fix-up) asserts `q.is_cuda`, so those ~40 lines are re-stated here
line for line around the reference kernels.

    TRITON_INTERPRET=1 python tests/golden/gen_golden.py

writes tests/golden/*.npz (inputs + reference outputs, fp16/bf16 as uint16 bit patterns).
"""
import importlib.util
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
import numpy as np
import torch

REF = "/root/reference/sageattention/triton"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


quant_pb = _load("quant_per_block")
quant_pb_varlen = _load("quant_per_block_varlen")
quant_pt = _load("quant_per_thread")
attn_nc = _load("attn_qk_int8_per_block")
attn_c = _load("attn_qk_int8_per_block_causal")
attn_nc_varlen = _load("attn_qk_int8_block_varlen")
attn_c_varlen = _load("attn_qk_int8_per_block_causal_varlen")


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def ref_dense(q, k, v, is_causal, sm_scale=None, smooth_k=True, return_lse=True, attn_mask=None):
    """core.py:242-331 with tensor_layout="HND", quantization_backend="triton"."""
    dtype = q.dtype
    head_dim_og = q.size(-1)
    if head_dim_og < 64:
        q, k, v = (torch.nn.functional.pad(t, (0, 64 - head_dim_og)) for t in (q, k, v))
    elif 64 < head_dim_og < 128:
        q, k, v = (torch.nn.functional.pad(t, (0, 128 - head_dim_og)) for t in (q, k, v))
    km = k.mean(dim=2, keepdim=True) if smooth_k else None
    lse_correction = None
    if smooth_k and return_lse:
        g = q.size(1) // k.size(1)
        kmb = torch.repeat_interleave(km, g, dim=1) if g > 1 else km
        lse_correction = torch.matmul(q, kmb.transpose(2, 3)).squeeze(-1).to(torch.float32)
    if dtype == torch.bfloat16:
        v = v.to(torch.float16)
    if sm_scale is None:
        sm_scale = 1.0 / (head_dim_og ** 0.5)
    q_int8, q_scale, k_int8, k_scale = quant_pb.per_block_int8(q, k, km=km, sm_scale=sm_scale, tensor_layout="HND")
    if is_causal:
        o, lse = attn_c.forward(q_int8, k_int8, v, q_scale, k_scale, tensor_layout="HND", output_dtype=dtype, return_lse=return_lse)
    else:
        if attn_mask is not None:      # core.py:313-322
            attn_mask = attn_mask.expand((q.shape[0], q.shape[1], q.shape[2], k.shape[2]))
        o, lse = attn_nc.forward(q_int8, k_int8, v, q_scale, k_scale, tensor_layout="HND", output_dtype=dtype, attn_mask=attn_mask,
                                 return_lse=return_lse)
    o = o[..., :head_dim_og]
    if return_lse:
        lse = lse / 1.44269504 + lse_correction * sm_scale if smooth_k else lse / 1.44269504
    return o, lse, dict(q_int8=q_int8, q_scale=q_scale, k_int8=k_int8, k_scale=k_scale, km=km)


def ref_varlen(q, k, v, cu_q, cu_k, max_q, max_k, is_causal, sm_scale=None, smooth_k=True):
    """core.py:399-448."""
    dtype = q.dtype
    head_dim_og = q.size(-1)
    if dtype == torch.bfloat16:
        v = v.to(torch.float16)
    if smooth_k:
        km = k.mean(dim=0, keepdim=True)
        k = k - km
    if sm_scale is None:
        sm_scale = 1.0 / (head_dim_og ** 0.5)
    q_int8, q_scale, k_int8, k_scale, cu_qs, cu_ks = quant_pb_varlen.per_block_int8(q, k, cu_q, cu_k, max_q, max_k, sm_scale=sm_scale)
    fwd = attn_c_varlen.forward if is_causal else attn_nc_varlen.forward
    o = fwd(q_int8, k_int8, v, cu_q, cu_k, max_q, q_scale, k_scale, cu_qs, cu_ks, output_dtype=dtype)
    return o, dict(q_int8=q_int8, q_scale=q_scale, k_int8=k_int8, k_scale=k_scale, cu_qs=cu_qs, cu_ks=cu_ks)


def sdpa_f32(q, k, v, is_causal, sm_scale=None):
    qf, kf, vf = q.float(), k.float(), v.float()
    g = q.size(1) // k.size(1)
    if g > 1:
        kf, vf = kf.repeat_interleave(g, 1), vf.repeat_interleave(g, 1)
    return torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=is_causal, scale=sm_scale)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def dense_case(name, B, Hq, Hkv, Lq, Lk, D, dtype, causal, kbias=0.0, seed=0):
    torch.manual_seed(seed)
    q = torch.randn(B, Hq, Lq, D).to(dtype)
    k = (torch.randn(B, Hkv, Lk, D) + kbias * torch.randn(1, Hkv, 1, D)).to(dtype)
    v = torch.randn(B, Hkv, Lk, D).to(dtype)
    o, lse, aux = ref_dense(q, k, v, causal)
    truth = sdpa_f32(q, k, v, causal) if Lq == Lk or not causal else None
    arrs = dict(q=bits(q), k=bits(k), v=bits(v), o=bits(o), lse=lse.numpy(),
                q_int8=aux["q_int8"].numpy(), q_scale=aux["q_scale"].numpy(),
                k_int8=aux["k_int8"].numpy(), k_scale=aux["k_scale"].numpy(), km=bits(aux["km"]),
                meta=np.array([B, Hq, Hkv, Lq, Lk, D, 0 if dtype == torch.float16 else 1, int(causal)], dtype=np.int64))
    if truth is not None:
        cos = torch.nn.functional.cosine_similarity(o.float().flatten(), truth.flatten(), dim=0).item()
        rmse = (o.float() - truth).pow(2).mean().sqrt().item()
        print(f"  {name}: reference vs fp32 SDPA cos={cos:.6f} rmse={rmse:.2e}")
    save(name, **arrs)


def mask_case(name, B, Hq, Hkv, Lq, Lk, D, dtype, kind, seed=0):
    """Triton API with attn_mask (bool or additive in the dtype of q), non-causal."""
    torch.manual_seed(seed)
    q = torch.randn(B, Hq, Lq, D).to(dtype)
    k = (torch.randn(B, Hkv, Lk, D) + torch.randn(1, Hkv, 1, D)).to(dtype)
    v = torch.randn(B, Hkv, Lk, D).to(dtype)
    if kind == "bool":
        mask = torch.rand(B, 1, Lq, Lk) > 0.35
        mask[:, :, :128, 64:128] = False           # a whole 128x64 block that must be skipped
        mask[:, :, 130:140, :] = False              # fully masked rows
        mask[:, :, :, 0] |= torch.rand(B, 1, Lq) > 0.1
        mask[:, :, 130:140, :] = False
    elif kind == "bool_skipall":                    # the whole first query block sees no key: EVERY tile of it is skipped (o = 0 / l_i's initial 1.0, lse = -inf)
        mask = torch.rand(B, 1, Lq, Lk) > 0.4
        mask[:, :, :128, :] = False
        mask[:, :, 131, :] = False                  # and one fully masked row inside a block that does run
    else:
        mask = (2.0 * torch.randn(1, Hq, Lq, Lk)).to(dtype)
        mask[:, :, :, 5:9] = -30000.0
    o, lse, aux = ref_dense(q, k, v, False, attn_mask=mask)
    m = mask.numpy() if kind.startswith("bool") else bits(mask)
    save(name, q=bits(q), k=bits(k), v=bits(v), o=bits(o), lse=lse.numpy(), mask=m,
         meta=np.array([B, Hq, Hkv, Lq, Lk, D, 0 if dtype == torch.float16 else 1, 0], dtype=np.int64))


def varlen_case(name, lens, Hq, Hkv, D, dtype, causal, seed=0):
    torch.manual_seed(seed)
    total = sum(lens)
    q = torch.randn(total, Hq, D).to(dtype)
    k = (torch.randn(total, Hkv, D) + 1.5 * torch.randn(1, Hkv, D)).to(dtype)
    v = torch.randn(total, Hkv, D).to(dtype)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    o, aux = ref_varlen(q, k, v, cu, cu, max(lens), max(lens), causal)
    save(name, q=bits(q), k=bits(k), v=bits(v), o=bits(o), cu=cu.numpy(),
         q_int8=aux["q_int8"].numpy(), q_scale=aux["q_scale"].numpy(),
         k_int8=aux["k_int8"].numpy(), k_scale=aux["k_scale"].numpy(),
         cu_qs=aux["cu_qs"].numpy(), cu_ks=aux["cu_ks"].numpy(),
         meta=np.array([len(lens), Hq, Hkv, total, total, D, 0 if dtype == torch.float16 else 1, int(causal)], dtype=np.int64))


def varlen_cross_case(name, lens_q, lens_k, Hq, Hkv, D, dtype, causal, seed=0):
    """cu_seqlens_q != cu_seqlens_k (cross-attention-like packing; causal stays top-left aligned,
    attn_qk_int8_per_block_causal_varlen.py)."""
    torch.manual_seed(seed)
    tq, tk = sum(lens_q), sum(lens_k)
    q = torch.randn(tq, Hq, D).to(dtype)
    k = (torch.randn(tk, Hkv, D) + 1.5 * torch.randn(1, Hkv, D)).to(dtype)
    v = torch.randn(tk, Hkv, D).to(dtype)
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32)
    o, aux = ref_varlen(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal)
    save(name, q=bits(q), k=bits(k), v=bits(v), o=bits(o), cu_q=cu_q.numpy(), cu_k=cu_k.numpy(),
         q_int8=aux["q_int8"].numpy(), q_scale=aux["q_scale"].numpy(),
         k_int8=aux["k_int8"].numpy(), k_scale=aux["k_scale"].numpy(),
         cu_qs=aux["cu_qs"].numpy(), cu_ks=aux["cu_ks"].numpy(),
         meta=np.array([len(lens_q), Hq, Hkv, tq, tk, D, 0 if dtype == torch.float16 else 1, int(causal)], dtype=np.int64))


def per_thread_case(name, B, Hq, Hkv, Lq, Lk, D, dtype, seed=0, BLKQ=128, WARPQ=32, BLKK=64, WARPK=64):
    """quant_per_thread.py:154-203 with the groups of the caller: the fp8 / fp16 CUDA APIs' default (BLKQ 128, WARPQ 32, BLKK 64, WARPK 64:
    core.py:601-602,790-791), the fp16+fp32 API at D = 128 (WARPQ 16, core.py:604), the sm90 API (BLKQ 64, WARPQ 16, BLKK 128, WARPK 128: core.py:967)."""
    torch.manual_seed(seed)
    q = torch.randn(B, Hq, Lq, D).to(dtype)
    k = (torch.randn(B, Hkv, Lk, D) + 2.0).to(dtype)
    km = k.mean(dim=2, keepdim=True)
    q8, qs, k8, ks = quant_pt.per_thread_int8(q, k, km, BLKQ=BLKQ, WARPQ=WARPQ, BLKK=BLKK, WARPK=WARPK)
    extra = {} if (BLKQ, WARPQ, BLKK, WARPK) == (128, 32, 64, 64) else dict(groups=np.array([BLKQ, WARPQ, BLKK, WARPK], dtype=np.int64))
    save(name, q=bits(q), k=bits(k), km=bits(km), q_int8=q8.numpy(), q_scale=qs.numpy(), k_int8=k8.numpy(), k_scale=ks.numpy(),
         meta=np.array([B, Hq, Hkv, Lq, Lk, D, 0 if dtype == torch.float16 else 1, 0], dtype=np.int64), **extra)


def long_cases():
    """Round 6: key ranges long enough for several trips of the kernels' six-body pipelined loop and a remainder behind them
    (17 whole 64-key tiles + a ragged one; 1000 causal rows in 128-row blocks)."""
    dense_case("long_nc_lq256_lk1100_d128_f16", 1, 2, 1, 256, 1100, 128, torch.float16, False, kbias=1.0, seed=16)
    dense_case("long_c_n1000_d64_bf16", 1, 2, 2, 1000, 1000, 64, torch.bfloat16, True, kbias=0.5, seed=17)


def per_thread_group_cases():
    f16, bf16 = torch.float16, torch.bfloat16
    per_thread_case("per_thread_sm90_d128_f16", 1, 2, 1, 200, 300, 128, f16, seed=13, BLKQ=64, WARPQ=16, BLKK=128, WARPK=128)
    per_thread_case("per_thread_sm90_d64_bf16", 2, 2, 2, 130, 129, 64, bf16, seed=14, BLKQ=64, WARPQ=16, BLKK=128, WARPK=128)
    per_thread_case("per_thread_warpq16_d128_f16", 1, 2, 1, 200, 150, 128, f16, seed=15, BLKQ=128, WARPQ=16, BLKK=64, WARPK=64)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; golden vectors can only be generated in the build container")
    f16, bf16 = torch.float16, torch.bfloat16
    if len(sys.argv) > 1 and sys.argv[1] == "--masks-only":
        mask_case("mask_bool_lq300_lk333_d64_f16", 2, 4, 2, 300, 333, 64, f16, "bool", seed=8)
        mask_case("mask_add_lq200_lk256_d128_bf16", 1, 2, 2, 200, 256, 128, bf16, "add", seed=9)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--mask-skipall-only":
        mask_case("mask_bool_skipall_lq140_lk130_d64_f16", 1, 2, 1, 140, 130, 64, f16, "bool_skipall", seed=12)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--per-thread-groups-only":
        per_thread_group_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--long-only":
        long_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--varlen-cross-only":
        varlen_cross_case("varlenx_nc_d128_bf16", [100, 257, 64], [333, 64, 500], 4, 2, 128, bf16, False, seed=10)
        varlen_cross_case("varlenx_c_d64_f16", [200, 130, 70], [300, 130, 40], 4, 1, 64, f16, True, seed=11)
        sys.exit(0)
    dense_case("c1_b1h4n512d64_f16", 1, 4, 4, 512, 512, 64, f16, False)             # BASELINE.json configs[0]
    dense_case("gqa_causal_n300d128_bf16", 1, 4, 2, 300, 300, 128, bf16, True, kbias=2.0, seed=1)
    dense_case("cross_lq200_lk333_d64_f16", 2, 2, 2, 200, 333, 64, f16, False, kbias=1.0, seed=2)
    dense_case("causal_n384d128_f16", 1, 2, 2, 384, 384, 128, f16, True, seed=3)
    dense_case("pad_d96_n160_f16", 1, 2, 1, 160, 160, 96, f16, False, seed=4)
    varlen_case("varlen_nc_d64_f16", [100, 257, 64], 4, 2, 64, f16, False, seed=5)
    varlen_case("varlen_c_d64_f16", [100, 257, 64], 4, 2, 64, f16, True, seed=5)
    varlen_case("varlen_c_d128_bf16", [130, 64, 300], 4, 1, 128, bf16, True, seed=6)
    per_thread_case("per_thread_quant_d128_f16", 1, 2, 1, 200, 150, 128, f16, seed=7)
    mask_case("mask_bool_lq300_lk333_d64_f16", 2, 4, 2, 300, 333, 64, f16, "bool", seed=8)
    mask_case("mask_add_lq200_lk256_d128_bf16", 1, 2, 2, 200, 256, 128, bf16, "add", seed=9)
    varlen_cross_case("varlenx_nc_d128_bf16", [100, 257, 64], [333, 64, 500], 4, 2, 128, bf16, False, seed=10)
    varlen_cross_case("varlenx_c_d64_f16", [200, 130, 70], [300, 130, 40], 4, 1, 64, f16, True, seed=11)
    mask_case("mask_bool_skipall_lq140_lk130_d64_f16", 1, 2, 1, 140, 130, 64, f16, "bool_skipall", seed=12)
    per_thread_group_cases()
    long_cases()
