"""Host logic of the caller shims (sageattention_amd/processors.py) on CPU with an injected attention function,
and -- marked gpu -- the same module running the HIP kernels against fp32 SDPA."""
import pytest
import torch
import torch.nn.functional as F

import util
from sageattention_amd import processors


class FakeAttention(torch.nn.Module):
    """The attribute surface diffusers' Attention exposes to its processors."""

    def __init__(self, dim, ctx_dim, heads, norm=True):
        super().__init__()
        self.heads = heads
        self.to_q = torch.nn.Linear(dim, dim, bias=False)
        self.to_k = torch.nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = torch.nn.Linear(ctx_dim, dim, bias=False)
        self.norm_q = torch.nn.LayerNorm(dim) if norm else None
        self.norm_k = torch.nn.LayerNorm(dim) if norm else None
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(dim, dim), torch.nn.Identity()])
        self.processor = None

    def forward(self, x, ctx=None, mask=None):
        return self.processor(self, x, encoder_hidden_states=ctx, attention_mask=mask)


def reference_forward(attn, x, ctx=None, mask=None):
    c = x if ctx is None else ctx
    q, k, v = attn.to_q(x), attn.to_k(c), attn.to_v(c)
    if attn.norm_q is not None:
        q, k = attn.norm_q(q), attn.norm_k(k)
    q, k, v = (t.unflatten(2, (attn.heads, -1)).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=mask).to(x.dtype)
    return attn.to_out[1](attn.to_out[0](o.transpose(1, 2).flatten(2, 3)))


def sdpa_nhd(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, tensor_layout="NHD"):
    assert tensor_layout == "NHD" and dropout_p == 0.0
    o = F.scaled_dot_product_attention(*(t.transpose(1, 2).float() for t in (q, k, v)), attn_mask=attn_mask, is_causal=is_causal)
    return o.transpose(1, 2).to(q.dtype)


def test_processor_host_logic_self_and_cross_attention():
    torch.manual_seed(0)
    attn, attn_self = FakeAttention(64, 48, 4), FakeAttention(64, 64, 4)
    model = torch.nn.ModuleDict({"cross": attn, "self": attn_self, "other": torch.nn.Linear(4, 4)})
    assert processors.set_sage_attention(model, attn_func=sdpa_nhd) == 2
    x, ctx = torch.randn(2, 10, 64), torch.randn(2, 7, 48)
    assert torch.allclose(attn_self(x), reference_forward(attn_self, x), atol=1e-5)
    assert torch.allclose(attn(x, ctx), reference_forward(attn, x, ctx), atol=1e-5)          # Lq != Lk
    mask = torch.rand(2, 1, 10, 7) > 0.3
    mask[..., 0] = True
    assert torch.allclose(attn(x, ctx, mask), reference_forward(attn, x, ctx, mask), atol=1e-5)


def test_processor_rotary_callback_and_errors():
    attn = FakeAttention(32, 32, 2, norm=False)
    calls = []
    attn.processor = processors.SageAttnProcessor(sdpa_nhd, apply_rotary=lambda t, r: (calls.append(t.shape), t * r)[1])
    x = torch.randn(1, 5, 32)
    attn.processor(attn, x, rotary_emb=torch.ones(1, 5, 1, 1))
    assert calls == [torch.Size([1, 5, 2, 16])] * 2                                           # q and k, NHD
    attn.processor = processors.SageAttnProcessor(sdpa_nhd)
    with pytest.raises(ValueError):
        attn.processor(attn, x, rotary_emb=torch.ones(1))
    with pytest.raises(NotImplementedError):
        processors.sdpa(x, x, x, dropout_p=0.1)
    with pytest.raises(ValueError):
        processors.sdpa(x, x, x, attn_mask=torch.ones(1), is_causal=True)


@pytest.mark.gpu
@pytest.mark.parametrize("cross,masked", [(False, False), (True, False), (True, True)])
def test_processor_on_gpu_matches_sdpa(cross, masked):
    """Wan/Mochi/LTX-shaped use: bf16 module, [B, L, H*D] activations, head_dim 128, cross-attention Lq != Lk."""
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    attn = FakeAttention(512, 512, 4).to(dev).to(torch.bfloat16)
    assert processors.set_sage_attention(attn) == 1
    x = torch.randn(2, 300, 512, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(2, 77, 512, device=dev, dtype=torch.bfloat16) if cross else None
    mask = None
    if masked:
        mask = torch.ones(2, 1, 300, 77, dtype=torch.bool, device=dev)
        mask[1, :, :, 50:] = False                                                            # padded text tokens
    with torch.no_grad():
        got = attn(x, ctx, mask).float()
        want = reference_forward(attn, x, ctx, mask).float()
    rel = (got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()
    assert torch.isfinite(got).all() and rel.item() <= (0.03 if masked else 0.06), rel.item()
    cos = util.cos_sim(got.cpu().numpy(), want.cpu().numpy())
    assert cos >= 0.998, cos
