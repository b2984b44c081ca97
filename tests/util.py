"""Shared helpers for the parity tests (torch <-> oracle plumbing, tile-image decoding)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = [int(x) for x in z["meta"]]
    return z, meta


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def bits(t: torch.Tensor) -> np.ndarray:
    """fp16/bf16 tensor -> uint16 bit patterns (numpy, contiguous, on host)."""
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16).copy()


def from_bits(a: np.ndarray, dtype_code: int, device="cpu") -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy())
    return t.view(torch.float16 if dtype_code == 0 else torch.bfloat16).to(device)


def f32(a: np.ndarray, dtype_code: int) -> np.ndarray:
    if dtype_code == 0:
        return a.view(np.float16).astype(np.float32)
    return (a.astype(np.uint32) << 16).view(np.float32)


def pv_token_of_position(p):
    g, c, j = p >> 5, (p >> 3) & 3, p & 7
    return 16 * c + 8 * (j >> 2) + 4 * g + (j & 3)


_TOK = np.array([pv_token_of_position(p) for p in range(64)])


def _image_cols(D: int, fp8: bool) -> np.ndarray:
    """cols[d, p] = element index inside image row d that holds position p (XOR-swizzled chunks:
    fp8 rows are 4 chunks of 16 with chunk ^= (d>>2)&3, fp16 rows 8 chunks of 8 with chunk ^= (d>>1)&7)."""
    d = np.arange(D)[:, None]
    p = np.arange(64)[None, :]
    per = 16 if fp8 else 8
    ch, i = p // per, p % per
    phys = (ch ^ ((d >> 2) & 3)) if fp8 else (ch ^ ((d >> 1) & 7))
    return phys * per + i


def encode_v_image(v: np.ndarray, fp8: bool) -> np.ndarray:
    """Logical [..., L, D] -> tile image [..., T, D, 64] exactly as csrc/sage_prep_v.hip lays it out."""
    *lead, L, D = v.shape
    T = (L + 63) // 64
    pad = np.zeros((*lead, T * 64, D), dtype=v.dtype)
    pad[..., :L, :] = v
    xp = pad.reshape(*lead, T, 64, D)[..., _TOK, :]            # [..., T, p, D]
    xp = np.swapaxes(xp, -1, -2)                               # [..., T, D, p]
    img = np.empty_like(xp)
    cols = np.broadcast_to(_image_cols(D, fp8), xp.shape)
    np.put_along_axis(img, cols, xp, axis=-1)
    return img


def decode_v_image(img: np.ndarray, L: int, fp8: bool) -> np.ndarray:
    """Inverse of encode_v_image: [..., T, D, 64] -> logical [..., L, D]."""
    *lead, T, D, W = img.shape
    assert W == 64
    cols = np.broadcast_to(_image_cols(D, fp8), img.shape)
    xp = np.swapaxes(np.take_along_axis(img, cols, axis=-1), -1, -2)   # [..., T, p, D]
    out = np.empty_like(xp)
    out[..., _TOK, :] = xp
    return out.reshape(*lead, T * 64, D)[..., :L, :]


def out_ulp(scale: float, dtype_code: int) -> float:
    """One ulp of the output dtype AT the value `scale` (= max|o|): 2^(floor(log2 scale) - 7) for bf16, - 10 for fp16.  (A fixed fraction of
    max|o| is wrong on one side: 2^-8 max|o| is half an ulp just above a power of two -- seed 425 of a 2000-seed run, one bf16 output one ulp
    off at 0.61 -- and 2^-7 max|o| is two ulps just below one.)"""
    import math
    if not scale > 0.0:
        return 0.0
    return max(2.0 ** (math.floor(math.log2(scale)) - (7 if dtype_code == 1 else 10)), 2.0 ** -24 if dtype_code == 0 else 0.0)   # (fp16 subnormals: 2^-24 apart)


def cos_sim(a: np.ndarray, b: np.ndarray) -> float:
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def rmse(a, b) -> float:
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def sdpa_f32(q, k, v, causal, sm_scale=None):
    """fp32 SDPA truth on whatever device the tensors live on (HND layout)."""
    qf, kf, vf = q.float(), k.float(), v.float()
    g = q.size(1) // k.size(1)
    if g > 1:
        kf, vf = kf.repeat_interleave(g, 1), vf.repeat_interleave(g, 1)
    scale = sm_scale if sm_scale is not None else q.size(-1) ** -0.5
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if causal:
        Lq, Lk = q.size(2), k.size(2)
        m = torch.ones(Lq, Lk, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~m, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), vf)


# ---- LSE merge reference (sequence-parallel callers; sageattention_amd/ring.py) -------------------------------------
def attn_with_lse_f32(q, k, v, tensor_layout="HND", is_causal=False, sm_scale=None, return_lse=True, **_):
    """Plain fp32 attention with the natural-log LSE, top-left causal mask (what sageattn(return_lse=True) returns)."""
    import torch
    if tensor_layout == "NHD":
        q, k, v = (x.transpose(1, 2) for x in (q, k, v))
    qf, kf, vf = q.float(), k.float(), v.float()
    g = qf.shape[1] // kf.shape[1]
    if g > 1:
        kf, vf = kf.repeat_interleave(g, 1), vf.repeat_interleave(g, 1)
    s = (qf @ kf.transpose(-1, -2)) * (sm_scale if sm_scale is not None else q.shape[-1] ** -0.5)
    if is_causal:
        Lq, Lk = s.shape[-2:]
        s = s.masked_fill(torch.ones(Lq, Lk, dtype=torch.bool).triu(1), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    o = (torch.softmax(s, dim=-1) @ vf).to(q.dtype)
    if tensor_layout == "NHD":
        o = o.transpose(1, 2)
    return o, lse


def merge_states_torch(o_acc, lse_acc, o_new, lse_new, tensor_layout="HND", first=False, out=None):
    """The formula of sage_merge_states in torch (fp32), in place on (o_acc [B,H,L,D], lse_acc [B,H,L])."""
    import torch
    on = o_new.float()
    if tensor_layout == "NHD":
        on = on.transpose(1, 2)
    if first:
        o_acc.copy_(on)
        lse_acc.copy_(lse_new)
    else:
        m = torch.maximum(lse_acc, lse_new)
        wa = torch.exp(lse_acc - m).unsqueeze(-1)
        wb = torch.exp(lse_new - m).unsqueeze(-1)
        dead = (m == float("-inf")).unsqueeze(-1)
        res = torch.where(dead, torch.zeros_like(on), (o_acc * wa + on * wb) / (wa + wb))
        lse = torch.where(dead.squeeze(-1), m, m + torch.log((wa + wb).squeeze(-1)))
        o_acc.copy_(res)
        lse_acc.copy_(lse)
    if out is not None:
        out.copy_((o_acc.transpose(1, 2) if tensor_layout == "NHD" else o_acc).to(out.dtype))
