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


# ------------------------------------------------------------------------------------------------ model-family processors
class RMSNorm(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = torch.nn.Parameter(1.0 + 0.1 * torch.randn(dim))

    def forward(self, x):
        return (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)).type_as(x) * self.weight


class FakeWanAttention(torch.nn.Module):
    """Wan's attention surface: full-width RMS norms, optional image-context projections (I2V cross-attention)."""

    def __init__(self, dim, heads, i2v=False):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(dim, dim) for _ in range(3))
        self.norm_q, self.norm_k = RMSNorm(dim), RMSNorm(dim)
        self.add_k_proj = torch.nn.Linear(dim, dim) if i2v else None
        self.add_v_proj = torch.nn.Linear(dim, dim) if i2v else None
        self.norm_added_k = RMSNorm(dim) if i2v else None
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(dim, dim), torch.nn.Dropout(0.0)])
        self.processor = None


def wan_reference(attn, x, ctx=None, rotary=None, text_len=512):
    """Independent restatement of modify_wan.py:16-99 in the [B, H, L, D] layout with fp32 SDPA."""
    img = None
    if attn.add_k_proj is not None:
        n_img = ctx.shape[1] - text_len
        img, ctx = ctx[:, :n_img], ctx[:, n_img:]
    c = x if ctx is None else ctx
    q, k, v = attn.norm_q(attn.to_q(x)), attn.norm_k(attn.to_k(c)), attn.to_v(c)
    q, k, v = (t.unflatten(2, (attn.heads, -1)) for t in (q, k, v))
    if rotary is not None:
        def rot(t, fc, fs):
            t1, t2 = t.unflatten(-1, (-1, 2)).unbind(-1)
            out = torch.empty_like(t)
            out[..., 0::2] = t1 * fc[..., 0::2] - t2 * fs[..., 1::2]
            out[..., 1::2] = t1 * fs[..., 1::2] + t2 * fc[..., 0::2]
            return out
        q, k = rot(q, *rotary), rot(k, *rotary)
    sd = lambda a, b, c_: F.scaled_dot_product_attention(a.transpose(1, 2).float(), b.transpose(1, 2).float(), c_.transpose(1, 2).float()).transpose(1, 2).flatten(2, 3).type_as(x)
    o = sd(q, k, v)
    if img is not None:
        ki = attn.norm_added_k(attn.add_k_proj(img)).unflatten(2, (attn.heads, -1))
        vi = attn.add_v_proj(img).unflatten(2, (attn.heads, -1))
        o = o + sd(q, ki, vi)
    return attn.to_out[1](attn.to_out[0](o))


def _wan_rotary(L, D):
    ang = torch.arange(L)[:, None].float() * torch.exp(-torch.arange(0, D, 2).float() / D * 4.0)[None, :]
    fc = torch.cos(ang).repeat_interleave(2, dim=-1)[None, :, None, :]      # pair value repeated on both channels
    fs = torch.sin(ang).repeat_interleave(2, dim=-1)[None, :, None, :]
    return fc, fs


def test_wan_processor_host_logic_t2v_and_i2v():
    torch.manual_seed(0)
    self_attn, cross = FakeWanAttention(64, 4), FakeWanAttention(64, 4, i2v=True)
    self_attn.processor = processors.SageWanAttnProcessor(sdpa_nhd)
    cross.processor = processors.SageWanAttnProcessor(sdpa_nhd, text_context_length=6)
    x = torch.randn(2, 12, 64)
    rot = _wan_rotary(12, 16)
    assert torch.allclose(self_attn.processor(self_attn, x, rotary_emb=rot), wan_reference(self_attn, x, rotary=rot), atol=2e-5)
    ctx = torch.randn(2, 5 + 6, 64)                                          # 5 image tokens + 6 text tokens
    assert torch.allclose(cross.processor(cross, x, encoder_hidden_states=ctx), wan_reference(cross, x, ctx, text_len=6), atol=2e-5)

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn1, self.attn2 = FakeWanAttention(64, 4), FakeWanAttention(64, 4, i2v=True)
    model = torch.nn.Module()
    model.blocks = torch.nn.ModuleList([Blk(), Blk()])
    assert processors.set_sage_attn_wan(model, sdpa_nhd) == 2
    assert all(isinstance(b.attn1.processor, processors.SageWanAttnProcessor) and b.attn2.processor is None for b in model.blocks)


class FakeMochiAttention(torch.nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(dim, dim, bias=False) for _ in range(3))
        self.add_q_proj, self.add_k_proj, self.add_v_proj = (torch.nn.Linear(ctx_dim, dim, bias=False) for _ in range(3))
        hd = dim // heads
        self.norm_q, self.norm_k, self.norm_added_q, self.norm_added_k = (RMSNorm(hd) for _ in range(4))
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(dim, dim), torch.nn.Dropout(0.0)])
        self.to_add_out = torch.nn.Linear(dim, ctx_dim)
        self.processor = None


def mochi_reference(attn, x, ctx, mask, rotary=None):
    """Independent restatement of modify_mochi.py:15-110: per-sample gather of the valid prompt tokens, fp32 SDPA, pad back."""
    H = attn.heads
    q, k, v = (p(x).unflatten(2, (H, -1)) for p in (attn.to_q, attn.to_k, attn.to_v))
    q, k = attn.norm_q(q), attn.norm_k(k)
    eq, ek, ev = (p(ctx).unflatten(2, (H, -1)) for p in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj))
    eq, ek = attn.norm_added_q(eq), attn.norm_added_k(ek)
    if rotary is not None:
        def rot(t, fc, fs):
            te, to = t[..., 0::2].float(), t[..., 1::2].float()
            return torch.stack([(te * fc - to * fs).to(t.dtype), (te * fs + to * fc).to(t.dtype)], dim=-1).flatten(-2)
        q, k = rot(q, *rotary), rot(k, *rotary)
    B, Lv, Le = x.shape[0], x.shape[1], ctx.shape[1]
    outs = []
    for b in range(B):
        idx = torch.nonzero(mask[b].flatten(), as_tuple=False).flatten()
        qb, kb, vb = (torch.cat([t[b:b + 1], e[b:b + 1, idx]], dim=1).transpose(1, 2).float() for t, e in ((q, eq), (k, ek), (v, ev)))
        ob = F.scaled_dot_product_attention(qb, kb, vb).transpose(1, 2).type_as(x)
        outs.append(F.pad(ob, (0, 0, 0, 0, 0, Lv + Le - ob.shape[1])))
    o = torch.cat(outs, dim=0).flatten(2, 3)
    h, e = o.split_with_sizes((Lv, Le), dim=1)
    return attn.to_out[1](attn.to_out[0](h)), attn.to_add_out(e)


def test_mochi_processor_host_logic_custom_attn_func():
    torch.manual_seed(1)
    attn = FakeMochiAttention(64, 48, 4)
    attn.processor = processors.SageMochiAttnProcessor(sdpa_nhd)
    x, ctx = torch.randn(2, 9, 64), torch.randn(2, 6, 48)
    mask = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 0]])
    ang = torch.rand(9, 4, 8)
    rot = (torch.cos(ang), torch.sin(ang))
    h, e = attn.processor(attn, x, ctx, mask, image_rotary_emb=rot)
    hr, er = mochi_reference(attn, x, ctx, mask, rot)
    assert torch.allclose(h, hr, atol=2e-5) and torch.allclose(e, er, atol=2e-5)

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn1 = FakeMochiAttention(64, 48, 4)
    model = torch.nn.Module()
    model.transformer_blocks = torch.nn.ModuleList([Blk(), Blk(), Blk()])
    assert processors.set_sage_attn_mochi(model) == 2                         # the last block keeps its processor
    assert model.transformer_blocks[2].attn1.processor is None


class FakeLTXAttention(FakeAttention):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__(dim, ctx_dim, heads, norm=False)
        self.norm_q, self.norm_k = RMSNorm(dim), RMSNorm(dim)
        self._proc = "original"

    def prepare_attention_mask(self, mask, target_length, batch_size):
        return mask.repeat_interleave(self.heads, dim=0)                      # [B*H, 1, Lk] additive, as diffusers does

    def get_processor(self):
        return self._proc

    def set_processor(self, p):
        self._proc = p
        self.processor = p


def test_ltx_processor_host_logic_mask_and_rotary():
    torch.manual_seed(2)
    attn = FakeLTXAttention(64, 64, 4)

    class Blk(torch.nn.Module):
        def __init__(self, a):
            super().__init__()
            self.attn1 = a
    model = torch.nn.Module()
    model.transformer_blocks = torch.nn.ModuleList([Blk(attn)])
    rot = lambda t, r: t * r
    assert processors.set_sage_attn_ltx(model, sdpa_nhd, apply_rotary=rot) == 1
    assert attn.origin_processor == "original" and isinstance(attn.processor, processors.SageLTXAttnProcessor)
    x = torch.randn(2, 10, 64)
    add = torch.zeros(2, 1, 10)
    add[1, :, 7:] = -1e4
    r = 1.0 + 0.1 * torch.randn(1, 10, 64)
    got = attn.processor(attn, x, attention_mask=add, image_rotary_emb=r)
    q, k, v = attn.norm_q(attn.to_q(x)) * r, attn.norm_k(attn.to_k(x)) * r, attn.to_v(x)
    q, k, v = (t.unflatten(2, (4, -1)).transpose(1, 2) for t in (q, k, v))
    want = F.scaled_dot_product_attention(q, k, v, attn_mask=add[:, None].expand(2, 4, 1, 10))
    want = attn.to_out[1](attn.to_out[0](want.transpose(1, 2).flatten(2, 3)))
    assert torch.allclose(got, want, atol=2e-5)


@pytest.mark.gpu
def test_wan_i2v_processor_on_gpu():
    """Wan I2V cross-attention on the HIP kernels: 257 image tokens + 512 text tokens, two attention calls summed."""
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    attn = FakeWanAttention(512, 4, i2v=True).to(dev).to(torch.bfloat16)
    attn.processor = processors.SageWanAttnProcessor()
    x = torch.randn(2, 640, 512, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(2, 257 + 512, 512, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        got, want = attn.processor(attn, x, encoder_hidden_states=ctx).float(), wan_reference(attn, x, ctx).float()
    assert util.cos_sim(got.cpu().numpy(), want.cpu().numpy()) >= 0.998
    self_attn = FakeWanAttention(512, 4).to(dev).to(torch.bfloat16)
    self_attn.processor = processors.SageWanAttnProcessor()
    rot = tuple(t.to(dev) for t in _wan_rotary(640, 128))
    with torch.no_grad():
        got, want = self_attn.processor(self_attn, x, rotary_emb=rot).float(), wan_reference(self_attn, x, rotary=rot).float()
    assert util.cos_sim(got.cpu().numpy(), want.cpu().numpy()) >= 0.998


@pytest.mark.gpu
def test_mochi_processor_packs_the_ragged_batch_into_one_varlen_call():
    """Default Mochi processor: all samples' (video + valid prompt) tokens in ONE sageattn_varlen call; against the per-sample
    fp32 restatement, and against the per-sample route of the same processor."""
    torch.manual_seed(4)
    dev = torch.device("cuda:0")
    attn = FakeMochiAttention(512, 256, 4).to(dev).to(torch.bfloat16)
    x = torch.randn(3, 500, 512, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(3, 64, 256, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(3, 64, dtype=torch.int64, device=dev)
    mask[0, :10], mask[1, :64], mask[2, :33] = 1, 1, 1
    with torch.no_grad():
        h, e = processors.SageMochiAttnProcessor()(attn, x, ctx, mask)
        h2, e2 = processors.SageMochiAttnProcessor(processors.sdpa)(attn, x, ctx, mask)
        hr, er = mochi_reference(attn, x, ctx, mask)
    for got, want in ((h, hr), (e, er), (h2, hr), (e2, er)):
        assert torch.isfinite(got.float()).all()
        assert util.cos_sim(got.float().cpu().numpy(), want.float().cpu().numpy()) >= 0.998
    assert (e[0, 10:] == attn.to_add_out(torch.zeros(1, 512, device=dev, dtype=torch.bfloat16))).all()   # padded prompt slots: zero attention output


@pytest.mark.gpu
def test_ltx_processor_on_gpu_with_padding_mask():
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    attn = FakeLTXAttention(512, 512, 4).to(dev).to(torch.bfloat16)
    attn.processor = processors.SageLTXAttnProcessor()
    x = torch.randn(2, 384, 512, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(2, 128, 512, device=dev, dtype=torch.bfloat16)
    add = torch.zeros(2, 1, 128, device=dev, dtype=torch.bfloat16)
    add[1, :, 90:] = -10000.0
    with torch.no_grad():
        got = attn.processor(attn, x, encoder_hidden_states=ctx, attention_mask=add).float()
        q, k, v = attn.norm_q(attn.to_q(x)), attn.norm_k(attn.to_k(ctx)), attn.to_v(ctx)
        q, k, v = (t.unflatten(2, (4, -1)).transpose(1, 2).float() for t in (q, k, v))
        want = F.scaled_dot_product_attention(q, k, v, attn_mask=add[:, None].float().expand(2, 4, 1, 128))
        want = attn.to_out[0](want.transpose(1, 2).flatten(2, 3).to(torch.bfloat16)).float()
    assert util.cos_sim(got.cpu().numpy(), want.cpu().numpy()) >= 0.998
