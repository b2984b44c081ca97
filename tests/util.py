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
