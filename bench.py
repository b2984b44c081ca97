#!/usr/bin/env python3
"""bench.py -- attention TFLOPS of the gfx950 SageAttention hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c4|c5|...] [--no-configs] [--no-sweep] [--sweep]
                    [--no-cpu-baseline] [--replay] [--dry-run-ranks N]

Workload of the headline line (default `c3` = BASELINE.json configs[2], the configuration the north-star target is quoted on):
B=2, H=32, N=8192, D=128, causal, INT8 QK^T + FP8 PV with two-level FP32 accumulation.  FLOPs = 4*B*H*N*N*D / 2 (causal) -- the
reference's formula (bench/bench_qk_int8_pv_fp8_cuda_sm90.py:34).

A "step" is one launch of the fused attention kernel on pre-quantised operands resident in HBM (exactly what the reference's bench
scripts time and what its published TOPS mean: "attention kernel only, excluding quantization and smoothing", README.md:174).
`value` is that kernel-only throughput.  The same run also times the whole `sageattn()` call (K mean + Q/K INT8 quantisation + V FP8
pre-pass + attention) and reports it under "end_to_end", plus accuracy vs fp32 SDPA under "accuracy".

The default single-GPU run appends, under "configs", every other BASELINE.json configuration, each with its own `roofline` object
(HIP events around the attention launch on the launch stream; peaks as for the headline):
  c2   qk_int8_pv_fp16 B=2 H=32 N=4096 D=128 causal: kernel-only + whole call; and the Triton-named API
       (sageattn_qk_int8_pv_fp16_triton, bench/bench_qk_int8_pv_fp16_triton.py) at the same shape: whole call + its attention kernel
  c4   sageattn_varlen GQA Hq=32 Hkv=8 D=128, lengths 256..16384, causal and non-causal: attention kernel only + whole call
  c5   CogVideoX1.5-5B shape B=2 H=48 N=17776 D=64: kernel-only + whole call + a short drop-in replay (42 layers x 2 denoising steps)
  sweep_b4_per_warp   the reference bench script's own shape (bench_qk_int8_pv_fp8_cuda_sm90.py:7-11,33-50): batch 4, `per_warp` scales
       in the sm90 kernels' groups (q per 16 rows, k per 128 keys), N = 1k..32k, causal and non-causal, kernel-only

Multi-GPU (driver launches one rank per GPU with torch.distributed.run): the path shards by (batch, kv-head) units with no data-path
collective.  The GLOBAL problem is batch B*world; its (batch, kv-head) units are generated one by one from a counter-based seed
(`unit_inputs`), so a rank builds only the units `shard.shard_range` gives it -- tests/test_bench_dry_run.py checks on CPU that this is
`shard.shard_bh` of the global tensors -- and holds B*H units (weak scaling); the only communication is the barrier and the MAX-reduce of
the elapsed time.  `--dry-run-ranks N` walks that split for N ranks on the host (unit ranges, FLOP accounting, per-rank HBM) without a GPU.

`--replay` (config c5, BASELINE.json configs[4]): the reference's drop-in usage replayed without the model weights --
`F.scaled_dot_product_attention = sageattn` (example/cogvideox_infer.py:34-35), then `--replay-layers` x `--replay-steps` (42 x 50 for
CogVideoX1.5-5B) calls of F.scaled_dot_product_attention on the model's attention shape; with N ranks the (batch, head) units of the one
global call are split across ranks (strong scaling, as example/run_parallel.sh splits one video over 8 GPUs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (B, H, Hkv, N, D, causal, pv, dtype)
    "c2": dict(B=2, H=32, Hkv=32, N=4096, D=128, causal=True, pv="fp16", dtype="fp16",
               workload="qk_int8_pv_fp16 B=2 H=32 N=4096 D=128 causal (BASELINE.json configs[1])"),
    "c3": dict(B=2, H=32, Hkv=32, N=8192, D=128, causal=True, pv="fp8", dtype="bf16",
               workload="qk_int8_pv_fp8 two-level accum B=2 H=32 N=8192 D=128 causal (BASELINE.json configs[2])"),
    "c3nc": dict(B=2, H=32, Hkv=32, N=8192, D=128, causal=False, pv="fp8", dtype="bf16",
                 workload="qk_int8_pv_fp8 two-level accum B=2 H=32 N=8192 D=128 NON-causal (balance probe)"),
    "c5": dict(B=2, H=48, Hkv=48, N=17776, D=64, causal=False, pv="fp8", dtype="bf16",
               workload="CogVideoX1.5-5B shaped sageattn() B=2 H=48 N=17776 D=64 non-causal (BASELINE.json configs[4])"),
    # XCD balance probes (head counts that are not multiples of 8)
    "h28": dict(B=1, H=28, Hkv=4, N=8192, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe B=1 H=28 Hkv=4 N=8192 causal"),
    "h12": dict(B=1, H=12, Hkv=12, N=8192, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe B=1 H=12 N=8192 causal"),
    "d64f16": dict(B=2, H=32, Hkv=32, N=8192, D=64, causal=True, pv="fp16", dtype="fp16", workload="probe D=64 FP16 PV B=2 H=32 N=8192 causal"),
    "n32k": dict(B=1, H=16, Hkv=16, N=32768, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe steady state B=1 H=16 N=32768 D=128 causal"),
    "c2nc": dict(B=2, H=32, Hkv=32, N=4096, D=128, causal=False, pv="fp16", dtype="fp16", workload="probe C2 shape NON-causal (FP16 PV tails)"),
    "c2l": dict(B=1, H=16, Hkv=16, N=16384, D=128, causal=True, pv="fp16", dtype="fp16", workload="probe steady state FP16 PV B=1 H=16 N=16384 D=128 causal"),
    "d64f8": dict(B=2, H=32, Hkv=32, N=8192, D=64, causal=True, pv="fp8", dtype="bf16", workload="probe D=64 FP8 PV B=2 H=32 N=8192 causal"),
    "n1k": dict(B=2, H=32, Hkv=32, N=1024, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe C3 shape at N=1024"),
    "n2k": dict(B=2, H=32, Hkv=32, N=2048, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe C3 shape at N=2048"),
    "n4k": dict(B=2, H=32, Hkv=32, N=4096, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe C3 shape at N=4096"),
    "n16k": dict(B=2, H=32, Hkv=32, N=16384, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe C3 shape at N=16384"),
    "h4": dict(B=1, H=4, Hkv=4, N=16384, D=128, causal=True, pv="fp8", dtype="bf16", workload="probe B=1 H=4 N=16384 causal"),
}
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md: bf16/f16 2.5 PF, fp8 5.0 PF (the MX-scaled
# instruction the PV step issues; the non-scaled fp8 MFMA runs at the bf16 rate), int8 = 2x bf16 = 5.0 POPS.
# Half of the FLOPs are INT8 (QK^T), half FP8 or FP16 (PV) -> harmonic blend:
PEAK_I8, PEAK_F8, PEAK_F16 = 5000.0, 5000.0, 2500.0
# HBM-side bytes per launch of the dominant kernels: read from profiles/r6_pmc.json, the summary tools/pmc_collect.py wrote from rocprofv3 --pmc
# passes at the commit named inside it ((2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB; FETCH_SIZE and WRITE_SIZE in passes of their own; the gfx950
# correction per MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide streaming reads).  One entry per driver-run configuration;
# profiles/r6_pmc_<cfg>.txt holds the raw counters and the other derived figures (VALU-active, MFMA-busy, waves per SIMD, LDS conflicts).
# algorithmic bytes: INT8 q (kernel-only bench) + 16-bit o + INT8 k + FP8 / FP16 V image; c4 / c2t: 16-bit q (quantised in the prologue)
ALGO_BYTES = {"c3": 335.5e6, "c5": 546.1e6, "c2": 201.3e6, "c4": 651.9e6, "c4nc": 651.9e6, "c2t": 268.4e6}
_PMC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6_pmc.json")
try:
    PMC = json.load(open(_PMC_PATH))
except Exception:          # (a checkout without the profile: traffic is reported as null, never as a stale constant)
    PMC = {}


def pmc_traffic(config_name):
    """(bytes per launch | None, note) for a bench configuration from the committed PMC summary."""
    e = PMC.get(config_name)
    if not e:
        return None, None
    return float(e["traffic_bytes"]), ("rocprofv3 --pmc passes at commit %s, %s: (2 x FETCH_SIZE %.0f + WRITE_SIZE %.0f) KiB; algorithmic %.4g B (x %.2f)"
                                       % (e.get("commit", "?"), e.get("file", "profiles/r6_pmc.json"), e["fetch_kib"], e["write_kib"],
                                          e["algorithmic_bytes"], e["traffic_over_algorithmic"]))


def blended_peak(pv: str) -> float:
    p2 = PEAK_F8 if pv == "fp8" else PEAK_F16
    return 1.0 / (0.5 / PEAK_I8 + 0.5 / p2)


def peak_note(pv: str) -> str:
    return ("harmonic blend of the dense MFMA peaks of the two halves: INT8 5.0 POPS (QK^T) and " +
            ("FP8 5.0 PF (MX-scaled instruction)" if pv == "fp8" else "FP16 2.5 PF") + " (PV)")


def flops(cfg) -> float:
    f = 4.0 * cfg["B"] * cfg["H"] * cfg["N"] * cfg["N"] * cfg["D"]
    return f / 2 if cfg["causal"] else f


def _dtype(cfg):
    return torch.float16 if cfg["dtype"] == "fp16" else torch.bfloat16


def unit_inputs(cfg, device, seed, lo, hi):
    """(batch, kv-head) units lo .. hi-1 of the GLOBAL problem as `shard.shard_bh` lays them out: q [1, units*g, N, D], k / v
    [1, units, N, D].  Unit u (= batch * Hkv + kv-head) is drawn from its own generator, seeded by (seed, u): every rank can build exactly
    its units, on its own device, and the global problem is their concatenation by construction -- no rank ever holds the global tensors
    (8 ranks of C3 would otherwise each allocate 3 x 1.07 GB of bf16 and a 2.1 GB fp32 staging tensor to use one eighth of it)."""
    dt = _dtype(cfg)
    g = cfg["H"] // cfg["Hkv"]
    N, D = cfg["N"], cfg["D"]
    n = hi - lo
    q = torch.empty((1, n * g, N, D), dtype=dt, device=device)
    k = torch.empty((1, n, N, D), dtype=dt, device=device)
    v = torch.empty((1, n, N, D), dtype=dt, device=device)
    for i, u in enumerate(range(lo, hi)):
        gen = torch.Generator(device=device).manual_seed(int(seed) * 1000003 + u)
        q[0, i * g:(i + 1) * g] = torch.randn((g, N, D), generator=gen, device=device, dtype=torch.float32).to(dt)
        k[0, i] = torch.randn((N, D), generator=gen, device=device, dtype=torch.float32).to(dt)
        v[0, i] = torch.randn((N, D), generator=gen, device=device, dtype=torch.float32).to(dt)
    return q, k, v


def make_inputs(cfg, device, seed, batch_mult=1):
    """randn q, k, v of the whole (global) problem [B, H, N, D]: the concatenation of its units (`unit_inputs`)."""
    B = cfg["B"] * batch_mult
    q, k, v = unit_inputs(cfg, device, seed, 0, B * cfg["Hkv"])
    N, D = cfg["N"], cfg["D"]
    return q.view(B, cfg["H"], N, D), k.view(B, cfg["Hkv"], N, D), v.view(B, cfg["Hkv"], N, D)


def rank_inputs(cfg, device, seed, rank, world, weak=True):
    """This rank's (batch, kv-head) units of the global problem (the split of sageattention_amd/shard.py), as [1, units*g, N, D] /
    [1, units, N, D] tensors.  weak: the global batch is B*world (B*Hkv units per rank); strong: the global batch is B."""
    from sageattention_amd import shard
    units = cfg["B"] * (world if weak else 1) * cfg["Hkv"]
    lo, hi = shard.shard_range(units, rank, world)
    q, k, v = unit_inputs(cfg, device, seed, lo, hi)
    return q, k, v, (lo, hi)


def dry_run(cfg, world, weak=True):
    """The N-rank split walked on the host, no tensors: unit range, shapes, FLOPs and HBM bytes of every rank.  What `main` relies on:
    the ranges partition the units, weak scaling gives every rank the same unit count (so `value = per-rank FLOPs * world / time` is the
    global FLOP count over the slowest rank's time), a strong split's FLOPs add up to the one global call."""
    from sageattention_amd import shard
    units = cfg["B"] * (world if weak else 1) * cfg["Hkv"]
    g = cfg["H"] // cfg["Hkv"]
    esz = 2
    ranks, covered = [], 0
    for r in range(world):
        lo, hi = shard.shard_range(units, r, world)
        assert lo == covered and hi >= lo, (r, lo, hi, covered)
        covered = hi
        n = hi - lo
        c = dict(cfg, B=1, H=n * g, Hkv=n)
        elems_q, elems_kv = n * g * cfg["N"] * cfg["D"], n * cfg["N"] * cfg["D"]
        hbm = (elems_q + 2 * elems_kv) * esz            # q, k, v
        hbm += elems_q * esz                            # o
        hbm += elems_q + elems_kv + elems_kv * (1 if cfg["pv"] == "fp8" else 2)      # INT8 q (kernel-only bench), INT8 k, V image
        ranks.append({"rank": r, "units": [lo, hi], "q_shape": [1, n * g, cfg["N"], cfg["D"]], "kv_shape": [1, n, cfg["N"], cfg["D"]],
                      "flops": flops(c), "hbm_bytes": hbm, "fp32_staging_bytes": g * cfg["N"] * cfg["D"] * 4})
    assert covered == units
    total = sum(r["flops"] for r in ranks)
    glob = flops(dict(cfg, B=cfg["B"] * (world if weak else 1)))
    assert abs(total - glob) <= 1e-9 * glob, (total, glob)
    if weak:
        assert len({r["units"][1] - r["units"][0] for r in ranks}) == 1, "weak scaling: every rank holds B*Hkv units"
        assert abs(ranks[0]["flops"] * world - glob) <= 1e-9 * glob
    return {"world": world, "scaling": "weak" if weak else "strong", "units_total": units, "global_flops": glob, "ranks": ranks}


def prequantize(cfg, q, k, v):
    """Operands of the kernel-only benchmark, produced by the product's own pre-pass kernels.  cfg["gran"]: "per_thread" (default, the
    reference APIs' default granularity) or "per_warp_sm90" (bench_qk_int8_pv_fp8_cuda_sm90.py's default: per-warp scales in the sm90
    kernels' groups, q per 16 rows, k per 128 keys)."""
    from sageattention_amd import _cabi, core, quant as sq
    sm = cfg["D"] ** -0.5
    if cfg.get("gran") == "per_warp_sm90":
        assert cfg["pv"] == "fp8"
        q8, qs, gran, q_warp, sm_log2 = core._quant_q(q, "per_warp", "HND", 16, sm, blkk=128)
        _, _, k8, ks, vimg, vscale, _ = core._prepass_kv(q, k, v, "HND", "per_warp", 128, True, False, False, sq.prepass_fused_ok(k))
        return q8, qs, k8, ks, vimg, vscale, gran, q_warp, sm_log2
    km = sq.channel_mean(k)
    q8, qs, k8, ks = sq.per_thread_int8(q, k, km)
    if cfg["pv"] == "fp8":
        vimg, vscale, _ = sq.per_channel_fp8(v)
    else:
        vimg, vscale = sq.prep_v_fp16(v), None
    return q8, qs, k8, ks, vimg, vscale, _cabi.GRAN_PER_THREAD, 32, sm * 1.44269504


def kernel_only_step(cfg, ops, sm_scale=None, folded=None):
    """One launch of the attention kernel on pre-quantised operands.  folded: None = the process default FP8 score form (exact -- the reference's
    formula -- unless SAGE_FP8_SCORES says otherwise), True = the opt-in folded variant."""
    from sageattention_amd import core, ops as sa_ops
    q8, qs, k8, ks, vimg, vscale, gran, q_warp, sm_log2 = ops
    return core._attn_dense(cfg["pv"] == "fp8", q8, k8, vimg, vscale, qs, ks, _dtype(cfg), "HND", cfg["causal"],
                            gran, q_warp, sm_log2, cfg["pv"] == "fp8", False, folded_scores=sa_ops.fp8_folded(None) if folded is None else folded)[0]


def folded_variant(cfg, ops, fl, steps, warmup, ramp):
    """The opt-in folded FP8 score form (fp8_scores="folded", SAGE_ATTR_FP8_FOLDED_SCORES) timed beside the default: NOT the reference's
    arithmetic, reported so that what the default's exactness costs is on the record."""
    _, d = timed(lambda: kernel_only_step(cfg, ops, folded=True), steps, warmup, False, ramp)
    ms = sum(d) / len(d)
    r = roofline_obj(fl, ms, "fp8", "sage_attn_kernel, SFOLD = true (opt-in variant)", None)
    return {"what": "the same launch with fp8_scores=\"folded\": one FMA per score, m + bias c' rounded once per (row, tile, k scale); opt-in, not the default, "
                    "not the reference's formula", "ms_per_launch": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "roofline": r}


def e2e_step(cfg, q, k, v):
    import sageattention_amd as sa
    if cfg["pv"] == "fp8":
        return sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=cfg["causal"], pv_accum_dtype="fp32+fp32")
    return sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=cfg["causal"], pv_accum_dtype="fp32")


def prepass_roofline(cfg, k, v, config_name, between=None):
    """The second kernel of a sageattn() call: the one-launch K / V pre-pass (sage_prepass_kv), HBM-bound by its
    arithmetic (2 B/element read + 1 B/element written for K and for V).  Launch duration from HIP events around each
    launch, with `between` (the attention kernel of the same workload, as in a real call sequence) run before every
    launch: back-to-back pre-pass launches find part of K / V in the 256 MB Infinity Cache and time 15-20 % too fast."""
    from sageattention_amd import quant as sq
    if cfg["pv"] != "fp8" or not sq.prepass_fused_ok(k):
        return None
    for _ in range(3):
        if between is not None:
            between()
        sq.prepass_kv_fp8(k, v)
    torch.cuda.synchronize()
    reps = 20
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        if between is not None:
            between()
        a.record()
        sq.prepass_kv_fp8(k, v)
        b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / reps
    nbytes = 2 * 3 * k.numel()
    return {"kernel": "prepass_kv_kernel", "bound": "hbm", "avg_launch_ms": round(ms, 4), "achieved": round(nbytes / ms / 1e6, 1),
            "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / 8000.0, 4), "algorithmic_bytes": nbytes,
            "traffic": pmc_traffic("pp")[0] if config_name == "c3" else None,
            "traffic_note": pmc_traffic("pp")[1] if config_name == "c3" else None,
            "how": "HIP events around each launch, the workload's attention kernel launched in between (cold Infinity Cache, as inside sageattn())"}


def timed(fn, steps, warmup, dist_on, ramp_s=0.0):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides; also per-step HIP
    event durations (events recorded on the stream the kernels are launched on = torch's current).
    ramp_s: untimed pre-phase that keeps the device busy with the same step for that many seconds.  An idle
    MI355X needs ~40-50 ms of continuous work to reach its sustained clocks (tools/dvfs_probe.py,
    profiles/r1_run40_dvfs.txt: launch 0 = 1139 us, launches 50..400 = 808-816 us); without it a short run
    times the ramp instead of the steady state."""
    import torch.distributed as dist
    if ramp_s > 0:
        t_end = time.perf_counter() + ramp_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = [a.elapsed_time(b) for a, b in evs]
    return wall, dev_ms


def _median(xs):
    """Median launch duration of a sweep point (20 launches): one stray multi-millisecond launch -- seen once in this round's runs, N = 4096
    at 224 instead of ~1400 TFLOP/s by the mean -- must not decide a point of the curve.  The headline and the `roofline` objects keep the
    mean the contract asks for."""
    ys = sorted(xs)
    return ys[len(ys) // 2] if len(ys) % 2 else 0.5 * (ys[len(ys) // 2 - 1] + ys[len(ys) // 2])


def roofline_obj(fl, kern_ms, pv, kernel, config_name=None):
    achieved = fl / (kern_ms * 1e-3) / 1e12
    peak = blended_peak(pv)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": pmc_traffic(config_name)[0],
            "traffic_note": pmc_traffic(config_name)[1],
            "kernel": kernel, "avg_launch_ms": round(kern_ms, 4), "algorithmic_flops_per_launch": fl, "peak_note": peak_note(pv),
            "how": "HIP events around every launch on the launch stream, average of the timed launches"}


def cpu_baseline(cfg):
    """The oracle (a straight CPU port of the reference algorithm, OpenMP) timed on this box's host cores on a bounded
    sample of the same workload: same N, D, mask and precision, a few (batch, head) units -- about 2-3 s, so that the
    GPU phases are not a footnote of the run.  `reference_path`: the reference's OWN CPU-runnable path (its Triton kernels under
    TRITON_INTERPRET=1), which cannot run on the GPU box (the reference is not there): the committed measurement of
    tools/ref_cpu_time.py from the build container, cores stated, with fp32 SDPA and this port timed beside it on the same shapes."""
    import numpy as np
    import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    N, D = cfg["N"], cfg["D"]
    # one (batch, head) unit is ceil(N / 128) independent row blocks for the OpenMP loop; a few units keep every core busy
    H = max(1, min(cfg["B"] * cfg["H"], max(1, cores // 32)))
    rng = np.random.default_rng(0)
    q8 = rng.integers(-95, 95, (1, H, N, D), dtype=np.int8)
    k8 = rng.integers(-95, 95, (1, H, N, D), dtype=np.int8)
    gq, nq = oracle.group_index(N, "per_thread", "q", 128, 32)
    gk, nk = oracle.group_index(N, "per_thread", "k", 64, 64)
    qs = (0.5 + rng.random((1, H, nq))).astype(np.float32) * 0.02
    ks = (0.5 + rng.random((1, H, nk))).astype(np.float32) * 0.02
    if cfg["pv"] == "fp8":
        v = oracle.convert(rng.standard_normal((1, H, N, D)).astype(np.float32) * 100, "e4m3")
        vs, mode = np.ones((1, H, D), np.float32), oracle.PV_F8_TWO_LEVEL
    else:
        v = oracle.convert(rng.standard_normal((1, H, N, D)).astype(np.float32), "f16")
        vs, mode = None, oracle.PV_F16_F32ACC
    t0 = time.perf_counter()
    oracle.attn(q8, k8, v, qs, gq, ks, gk, causal=cfg["causal"], c=0.1275, pv_mode=mode, out_dtype=0, v_scale=vs)
    dt = time.perf_counter() - t0
    fl = 4.0 * H * N * N * D / (2 if cfg["causal"] else 1)
    out = {"value": round(fl / dt / 1e12, 6), "unit": "TFLOP/s", "cores": cores, "kind": "port",
           "sample": f"oracle/sage_oracle.c orc_attn (OpenMP, {cores} threads): B=1 H={H} of the workload's "
                     f"{cfg['B'] * cfg['H']} (batch,head) units, N={N} D={D} causal={cfg['causal']} pv={cfg['pv']}, {dt:.1f} s"}
    ref = os.path.join(ROOT, "profiles", "ref_triton_cpu.json")
    if os.path.exists(ref):
        try:
            with open(ref) as f:
                r = json.load(f)
            out["reference_path"] = {
                "what": r.get("what"), "where": "build container (the reference is not on the GPU box); tools/ref_cpu_time.py -> profiles/ref_triton_cpu.json",
                "cores": r.get("cores"), "cpu": r.get("cpu"),
                "cases": [{"case": c["case"], "shape": c["shape"], "gflops": c["reference_triton_interpreter_gflops"],
                           "seconds": c["reference_triton_interpreter_seconds"], "fp32_sdpa_cpu_gflops": c["fp32_sdpa_cpu_gflops"],
                           "openmp_port_gflops": c.get("openmp_port_gflops"), "timing": c.get("timing")}
                          for c in r.get("cases", [])]}
        except Exception as e:          # the artefact is informational
            out["reference_path"] = {"error": repr(e)}
    return out


def accuracy(cfg, q, k, v):
    """cos-sim / relative RMSE of sageattn() vs fp32 SDPA on two (batch, head) units."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import util
    o = e2e_step(cfg, q[:1, :2], k[:1, :2], v[:1, :2])
    truth = util.sdpa_f32(q[:1, :2], k[:1, :2], v[:1, :2], cfg["causal"]).cpu().numpy()
    got = o.float().cpu().numpy()
    return {"cos_sim_vs_fp32_sdpa": round(util.cos_sim(got, truth), 6),
            "rel_rmse_vs_fp32_sdpa": round(util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean())), 5)}


C4_LENS = [256, 512, 1000, 1024, 2048, 4096, 8192, 16384]     # SURVEY.md 8d, BASELINE.json configs[3]


def measure_c4(steps, warmup, ramp, device):
    """BASELINE.json configs[3]: sageattn_varlen, GQA Hq=32 Hkv=8 D=128 bf16, mixed lengths.  Two numbers per mask: the attention kernel
    alone (the launch of core._varlen_attend on the operands its pre-pass produced, HIP events around it -> roofline against the
    INT8 / FP16 blend) and the whole call (plan launch + one-launch K / V pre-pass + attention)."""
    import sageattention_amd as sa
    from sageattention_amd import core
    g = torch.Generator(device="cpu").manual_seed(4)
    total = sum(C4_LENS)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16).to(device)
    k = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(device)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(device)
    cu = torch.tensor([0] + list(torch.tensor(C4_LENS).cumsum(0)), dtype=torch.int32, device=device)
    out = {"workload": "sageattn_varlen GQA Hq=32 Hkv=8 D=128 bf16, lengths " + str(C4_LENS) + " (BASELINE.json configs[3])",
           "dtype": "int8 QK^T + fp16 PV, fp32 accumulate"}
    for causal in (True, False):
        fl = sum(4.0 * 32 * L * L * 128 for L in C4_LENS) / (2 if causal else 1)
        st = core._varlen_prepare(q, k, v, cu, cu, max(C4_LENS), max(C4_LENS), causal, None, True, {})
        _, dev_k = timed(lambda: core._varlen_attend(st), steps, warmup, False, ramp)
        kern_ms = sum(dev_k) / len(dev_k)
        fn = lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(C4_LENS), max(C4_LENS), is_causal=causal)
        wall, _ = timed(fn, steps, max(2, warmup // 2), False, ramp)
        out["causal" if causal else "non_causal"] = {
            "kernel_only": {"ms_per_launch": round(kern_ms, 4), "tflops": round(fl / kern_ms / 1e9, 2)},
            "roofline": roofline_obj(fl, kern_ms, "fp16", "sage_attn_kernel (packed / varlen launch over the device-built work list)",
                                     "c4" if causal else "c4nc"),
            "end_to_end": {"ms_per_call": round(wall / steps * 1e3, 4), "tflops": round(fl / (wall / steps) / 1e12, 2),
                           "what": "sageattn_varlen(): plan launch + one-launch K/V pre-pass + attention"}}
        del st
    return out


def _score_form():
    from sageattention_amd import ops as sa_ops
    return "folded" if sa_ops.fp8_folded(None) else "exact"


def measure_dense(name, cfg, device, steps, warmup, ramp, with_e2e=True):
    """kernel-only + roofline (+ whole call) of one dense configuration on this device."""
    q, k, v = make_inputs(cfg, device, 1234)
    ops = prequantize(cfg, q, k, v)
    fl = flops(cfg)
    _, dev_k = timed(lambda: kernel_only_step(cfg, ops), steps, warmup, False, ramp)
    kern_ms = sum(dev_k) / len(dev_k)
    out = {"workload": cfg["workload"], "dtype": "int8 QK^T + " + ("fp8(e4m3) PV" if cfg["pv"] == "fp8" else "fp16 PV") + ", fp32 accumulate",
           "kernel_only": {"ms_per_launch": round(kern_ms, 4), "tflops": round(fl / kern_ms / 1e9, 2)},
           "roofline": roofline_obj(fl, kern_ms, cfg["pv"], "sage_attn_kernel", name)}
    if cfg["pv"] == "fp8":
        out["fp8_score_form"] = _score_form()
        out["fp8_folded_variant"] = folded_variant(cfg, ops, fl, max(5, steps // 2), 3, ramp)
    if cfg["pv"] == "fp16" and q.dtype == torch.float16:
        # the attention launch of the DEFAULT route of sageattn_qk_int8_pv_fp16_cuda on fp16 inputs at this size: Q quantised in the prologue, V rows
        # read in place (no tile image; bit-identical outputs) -- beside the reference-style number above (INT8 operands + image)
        from sageattention_amd import core
        q8, qs, k8, ks, vimg, vscale, gran, q_warp, sm_log2 = ops
        rows = core._v_rows_wanted(q, k, v, "HND", cfg["causal"], None)
        _, d = timed(lambda: core._attn_fused_q(q, k8, v if rows else vimg, None, ks, "HND", cfg["causal"], sm_log2, False, v_rows=rows), max(5, steps // 2), 3, False, ramp)
        ms = sum(d) / len(d)
        out["kernel_only_default_route"] = {"ms_per_launch": round(ms, 4), "tflops": round(fl / ms / 1e9, 2),
                                            "what": "fused per-thread Q quantisation" + (" + V rows in place (sage_attn_fused_q_pv_f16_vrows)" if rows else " + V tile image")}
    if with_e2e:
        n = max(3, steps // 2)
        wall, _ = timed(lambda: e2e_step(cfg, q, k, v), n, 2, False, ramp)
        out["end_to_end"] = {"ms_per_call": round(wall / n * 1e3, 4), "tflops": round(fl / (wall / n) / 1e12, 2),
                             "what": "whole call: K mean + INT8 Q/K quant + V pre-pass + attention"}
        if cfg["pv"] == "fp16" and q.dtype == torch.float16:
            import sageattention_amd as sa
            wall, _ = timed(lambda: sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=cfg["causal"], pv_accum_dtype="fp32", v_in_place=False), n, 2, False, ramp)
            out["end_to_end_v_image_route"] = {"ms_per_call": round(wall / n * 1e3, 4), "tflops": round(fl / (wall / n) / 1e12, 2),
                                               "what": "the same call with v_in_place=False: K + V tile image pre-pass (what bf16 inputs take)"}
    return out, (q, k, v)


def measure_triton_api(cfg, q, k, v, steps, warmup, ramp):
    """The Triton-named API (sageattn_qk_int8_pv_fp16_triton; the reference times its kernels in bench/bench_qk_int8_pv_fp16_triton.py) at
    the C2 shape: the whole call, and its attention kernel alone (per-block Q quantised in the prologue, per-block K scales, FP16 PV in the
    Triton kernels' form) on the operands its pre-pass produced."""
    import sageattention_amd as sa
    from sageattention_amd import core, quant as sq
    fl = flops(cfg)
    sm = cfg["D"] ** -0.5
    _, k8, ks, vimg, _, _ = sq.prepass_kv_fp8(k, v, "HND", smooth_k=True, qk_quant_gran="per_block_triton", v_fp16=True)
    rows = core._v_rows_wanted(q, k, v, "HND", cfg["causal"], None)       # (fp16 inputs at this size: V rows in place, the call's default route)
    _, dev_k = timed(lambda: core._attn_fused_qblock(q, k8, v if rows else vimg, ks, "HND", cfg["causal"], sm * 1.44269504, False, v_rows=rows),
                     steps, warmup, False, ramp)
    kern_ms = sum(dev_k) / len(dev_k)
    n = max(3, steps // 2)
    wall, _ = timed(lambda: sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, is_causal=cfg["causal"]), n, 2, False, ramp)
    # the reference's kernel-level entry under the reference's module path (what its bench/bench_qk_int8_pv_fp16_triton.py imports and times):
    # INT8 q / k with their scales and the fp16 value tensor in -- V read in place (ABI 21), nothing in front of the kernel
    from sageattention.triton.attn_qk_int8_per_block import forward as ref_forward
    from sageattention.triton.attn_qk_int8_per_block_causal import forward as ref_forward_causal
    q8, qs, _, _ = sq.per_block_int8(q, k, sm_scale=sm, tensor_layout="HND", k_done=(k8, ks))      # (the Q half; K comes smoothed from the pre-pass above)
    fwd = ref_forward_causal if cfg["causal"] else ref_forward
    _, dev_f = timed(lambda: fwd(q8, k8, v, qs, ks, output_dtype=q.dtype), steps, warmup, False, ramp)
    fwd_ms = sum(dev_f) / len(dev_f)
    return {"workload": "sageattn_qk_int8_pv_fp16_triton at the C2 shape (B=2 H=32 N=4096 D=128 causal, fp16)",
            "kernel_only": {"ms_per_launch": round(kern_ms, 4), "tflops": round(fl / kern_ms / 1e9, 2)},
            "kernel_level_forward": {"ms_per_call": round(fwd_ms, 4), "tflops": round(fl / fwd_ms / 1e9, 2),
                                     "what": "sageattention.triton.attn_qk_int8_per_block_causal.forward(q_int8, k_int8, v, q_scale, k_scale): the reference's "
                                             "kernel-level API and signature, INT8 operands, fp16 value rows read in place (sage_attn_qk_int8_pv_f16_vrows)"},
            "roofline": roofline_obj(fl, kern_ms, "fp16", "sage_attn_kernel (per-block Q quantised in the prologue, Triton kernel form" + (", V rows in place)" if rows else ")"), "c2t"),
            "end_to_end": {"ms_per_call": round(wall / n * 1e3, 4), "tflops": round(fl / (wall / n) / 1e12, 2)}}


def replay(cfg, q, k, v, layers, steps, dist_on=False):
    """`F.scaled_dot_product_attention = sageattn` (example/cogvideox_infer.py:34-35), then layers x steps calls on (q, k, v)."""
    import torch.nn.functional as F
    import torch.distributed as dist
    import sageattention_amd as sa
    orig = F.scaled_dot_product_attention
    F.scaled_dot_product_attention = sa.sageattn
    try:
        for _ in range(5):
            F.scaled_dot_product_attention(q, k, v, is_causal=cfg["causal"])
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_step = []
        for _ in range(steps):
            ts = time.perf_counter()
            for _ in range(layers):
                o = F.scaled_dot_product_attention(q, k, v, is_causal=cfg["causal"])
            torch.cuda.synchronize()
            per_step.append(time.perf_counter() - ts)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        F.scaled_dot_product_attention = orig
    return wall, per_step, o


def other_configs(args, device):
    """The `configs` block of the default line: every BASELINE.json configuration besides the headline's, and the reference bench
    script's own sweep shape."""
    ramp = min(args.ramp_seconds, 0.2)
    steps, warmup = max(10, args.steps // 2), max(5, args.warmup // 2)
    out = {}
    c2, (q, k, v) = measure_dense("c2", CONFIGS["c2"], device, steps, warmup, ramp)
    c2["triton_api"] = measure_triton_api(CONFIGS["c2"], q, k, v, steps, warmup, ramp)
    out["c2"] = c2
    del q, k, v
    out["c4"] = measure_c4(max(8, steps // 2), max(3, warmup // 2), ramp, device)
    c5, (q, k, v) = measure_dense("c5", CONFIGS["c5"], device, max(8, steps // 2), max(3, warmup // 2), ramp)
    layers, dsteps = 42, 2
    wall, per_step, _ = replay(CONFIGS["c5"], q, k, v, layers, dsteps)
    c5["replay"] = {"what": "F.scaled_dot_product_attention = sageattn, 42 layers x 2 denoising steps of the model's attention call (bench.py "
                            "--config c5 --replay runs the full 42 x 50)", "calls": layers * dsteps, "total_seconds": round(wall, 3),
                    "ms_per_call": round(wall / (layers * dsteps) * 1e3, 4), "tflops": round(flops(CONFIGS["c5"]) * layers * dsteps / wall / 1e12, 2)}
    out["c5"] = c5
    del q, k, v
    # a decode / cross-attention-like call (few query rows, long KV): 32 workgroups on a 256-CU part unless the key range is split.  FP8 split-KV
    # rounds every P against a per-chunk maximum -- 2.8e-2 rel-RMS from the unsplit result, not the reference's arithmetic -- so it is opt-in
    # (split_kv="auto"); the FP16-PV entry point, whose split result meets the unsplit oracle at the usual bar, plans it by itself
    import sageattention_amd as sa
    g = torch.Generator(device="cpu").manual_seed(5)
    qd = torch.randn(1, 32, 128, 128, generator=g).to(torch.bfloat16).to(device)
    kd, vd = (torch.randn(1, 32, 32768, 128, generator=g).to(torch.bfloat16).to(device) for _ in range(2))
    dec = {"workload": "B=1 H=32 Lq=128 Lk=32768 D=128 non-causal, bf16, whole calls (pre-pass of 32768 keys included)"}
    for label, fn in (("fp8_default_unsplit", lambda: sa.sageattn(qd, kd, vd)), ("fp8_split_kv_auto_opt_in", lambda: sa.sageattn(qd, kd, vd, split_kv="auto")),
                      ("fp16_pv_default_auto_split", lambda: sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd))):
        wall, _ = timed(fn, 10, 3, False, 0.0)
        dec[label] = {"us_per_call": round(wall / 10 * 1e6, 1)}
    out["decode_like"] = dec
    del qd, kd, vd
    # the reference bench script's own shape: batch 4, per_warp (sm90 groups), N = 1k .. 32k, causal and non-causal, kernel-only
    sw = {"what": "kernel-only TFLOP/s, batch 4, H=32, D=128, qk_quant_gran per_warp in the sm90 kernels' groups (q per 16 rows, k per 128 keys), "
                  "fp32+fp32 -- bench/bench_qk_int8_pv_fp8_cuda_sm90.py:7-11,33-50; median of 20 launches each (10 at N = 32k), HIP events",
          "h100_published_causal": {"1024": 448, "2048": 624, "4096": 744, "8192": 795, "16384": 835, "32768": 858}}
    for causal in (True, False):
        row = {}
        for n in (1024, 2048, 4096, 8192, 16384, 32768):
            c = dict(CONFIGS["c3"], N=n, B=4, causal=causal, gran="per_warp_sm90")
            qq, kk, vv = make_inputs(c, device, 99)
            oo = prequantize(c, qq, kk, vv)
            del qq, kk, vv
            _, d = timed(lambda: kernel_only_step(c, oo), 20 if n <= 16384 else 10, 5, False, ramp)
            row[str(n)] = round(flops(c) / (_median(d) * 1e-3) / 1e12, 1)
            del oo
        sw["causal" if causal else "non_causal"] = row
    out["sweep_b4_per_warp"] = sw
    return out


def run_c4(args, device):
    out = measure_c4(args.steps, args.warmup, args.ramp_seconds, device)
    print(json.dumps({"metric": "sageattn_varlen TFLOPS (INT8 QK^T + FP16 PV): attention kernel only (value) and whole call", "unit": "TFLOP/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "value": out["causal"]["kernel_only"]["tflops"],
                      "higher_is_better": True, "config": {"workload": out["workload"]}, "roofline": out["causal"]["roofline"],
                      "detail": out, "data": "synthetic"}))


def run_replay(args, cfg, device, rank, world, dist_on):
    """BASELINE.json configs[4] / SURVEY 8d C5: the drop-in replay.  F.scaled_dot_product_attention is replaced by
    sageattn exactly as the reference's example does, then the model's attention calls are replayed: layers x steps
    calls on the model's shape.  One global problem (batch B), its (batch, head) units split over the ranks."""
    import torch.distributed as dist
    q, k, v, (lo, hi) = rank_inputs(cfg, device, 4321, rank, world, weak=False)
    calls = args.replay_layers * args.replay_steps
    wall, per_step, o = replay(cfg, q, k, v, args.replay_layers, args.replay_steps, dist_on)
    stats = torch.tensor([wall], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    wall = stats.item()
    fl = flops(cfg)                                                   # one global call
    if rank == 0:
        print(json.dumps({
            "metric": "drop-in replay: F.scaled_dot_product_attention = sageattn, whole-call TFLOPS", "unit": "TFLOP/s",
            "value": round(fl * calls / wall / 1e12, 2), "n_gpus": world, "higher_is_better": True, "scaling": "strong",
            "calls": calls, "layers": args.replay_layers, "denoise_steps": args.replay_steps,
            "total_seconds": round(wall, 3), "ms_per_call": round(wall / calls * 1e3, 4),
            "ms_per_denoise_step_attention": round(sum(per_step) / len(per_step) * 1e3, 3),
            "config": {"workload": cfg["workload"], "global_batch": cfg["B"], "heads": cfg["H"], "seq_len": cfg["N"],
                       "head_dim": cfg["D"], "units_this_rank": hi - lo,
                       "parallelism": f"(batch, head) units of one call split over {world} rank(s), no collective"},
            "dtype": "int8 QK^T + fp8(e4m3) PV, fp32 accumulate", "data": "synthetic (randn)",
            "o_shape_rank0": list(o.shape)}))


def rank_record(rank, local_rank, have_gpu=True):
    """What a rank reports about itself (gathered on rank 0 into `ranks_seen`, so that a scaling record proves N ranks on N GPUs)."""
    if not have_gpu:
        return {"rank": rank, "local_rank": local_rank, "device": None, "device_count": 0, "name": "cpu (dist check)", "pci_bus_id": None,
                "uuid": f"cpu-{rank}", "pid": os.getpid()}
    props = torch.cuda.get_device_properties(local_rank)
    return {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "device_count": torch.cuda.device_count(),
            "name": torch.cuda.get_device_name(local_rank), "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
            "pid": os.getpid()}


def check_ranks(gathered, world, backend):
    """`ranks_seen` of a run from the gathered rank records; over RCCL every rank must sit on a device of its own."""
    import torch.distributed as dist
    seen = {"world_size": dist.get_world_size() if dist.is_initialized() else world, "backend": dist.get_backend() if dist.is_initialized() else backend,
            "ranks": gathered}
    try:
        seen["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        seen["rccl_version"] = None
    assert len(gathered) == world and sorted(g["rank"] for g in gathered) == list(range(world)), "a rank is missing from the gathered records"
    if backend == "nccl":
        devs = [(g["uuid"] or g["device"]) for g in gathered]
        assert len(set(devs)) == world, f"ranks share a device: {devs}"
    return seen


def init_ranks(rank, world, local_rank, device, backend, have_gpu=True):
    """Process group of a --gpus N run (one rank per GPU, RCCL; `gloo` for the CPU checks) and the `ranks_seen` record."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    gathered = [None] * world
    dist.all_gather_object(gathered, rank_record(rank, local_rank, have_gpu))
    return check_ranks(gathered, world, backend)


def reduce_times(stats, world, steps):
    """MAX over ranks of the timed regions (the contract's clock) and every rank's own kernel-only time per step."""
    import torch.distributed as dist
    allw = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(allw, stats)
    per_rank_ms = [round(t[0].item() / steps * 1e3, 4) for t in allw]
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    return stats, per_rank_ms


def dist_check(args, rank, world, local_rank):
    """--dist-check: the distributed control flow of a --gpus N run on CPU ranks over gloo -- process group, rank records, the unit split of
    every rank (shapes only), barrier, time reduction, the JSON line -- with a fake step time instead of kernels.  It exists so that the first
    real 8-GPU run can only fail on RCCL, not on a typo in this bookkeeping (tests/test_bench_dry_run.py runs it with 8 ranks)."""
    import torch.distributed as dist
    seen = init_ranks(rank, world, local_rank, None, "gloo", have_gpu=False)
    cfg = CONFIGS[args.config]
    from sageattention_amd import shard
    lo, hi = shard.shard_range(cfg["B"] * cfg["Hkv"] * world, rank, world)
    dist.barrier()
    stats = torch.tensor([0.010 * args.steps * (1.0 + 0.01 * rank), 0.012], dtype=torch.float64)       # rank r is r % slower
    stats, per_rank_ms = reduce_times(stats, world, args.steps)
    wall_k = stats.tolist()[0]
    fl = flops(cfg)
    out = {"dist_check": True, "n_gpus": world, "steps": args.steps, "ms_per_step": round(wall_k / args.steps * 1e3, 4),
           "value": round(fl * world / (wall_k / args.steps) / 1e12, 2), "ranks_seen": seen, "ms_per_step_per_rank": per_rank_ms,
           "units_of_this_rank": [lo, hi]}
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed clock-ramp phase before the warmup steps (0 disables); reported in the JSON")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["c4"])
    ap.add_argument("--sweep", action="store_true", help="also batch 4 (per-thread scales) in the N=1k..32k kernel-only sweep")
    ap.add_argument("--no-sweep", action="store_true", help="skip the N=1k..32k kernel-only sweep that the default c3 run appends (BASELINE.json's metric)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (c2, c4, c5, the reference bench script's sweep) of the default c3 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replay", action="store_true", help="drop-in replay of the model's attention calls (use with --config c5)")
    ap.add_argument("--replay-layers", type=int, default=42)
    ap.add_argument("--replay-steps", type=int, default=50)
    ap.add_argument("--dry-run-ranks", type=int, default=0, help="walk the N-rank split of the configuration on the host (no GPU) and print it")
    ap.add_argument("--dist-check", action="store_true", help="run the distributed bookkeeping of a --gpus N launch on CPU ranks over gloo (no GPU, no kernels)")
    args = ap.parse_args()

    if args.dry_run_ranks:
        cfg = CONFIGS[args.config if args.config != "c4" else "c3"]
        print(json.dumps({"dry_run": dry_run(cfg, args.dry_run_ranks, weak=not args.replay), "config": args.config}))
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.dist_check:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
        dist_check(args, rank, world, local_rank)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # dry-run aid for a 1-GPU box: SAGE_BENCH_BACKEND=gloo puts every rank on cuda:0 and uses gloo for the
    # barrier / MAX-reduce, to exercise the multi-process control flow without RCCL (never used by the driver)
    backend = os.environ.get("SAGE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = None
    if dist_on:
        import torch.distributed as dist
        ranks_seen = init_ranks(rank, world, local_rank, device, backend)

    from sageattention_amd import _cabi
    _cabi.load()

    if args.config == "c4":
        run_c4(args, device)
        return
    cfg = CONFIGS[args.config]
    if args.replay:
        run_replay(args, cfg, device, rank, world, dist_on)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    # weak scaling: the global batch is B*world; every rank builds its own B*Hkv (batch, kv-head) units
    q, k, v, _units = rank_inputs(cfg, device, 1234, rank, world, weak=True)
    cfg = dict(cfg, B=1, H=q.shape[1], Hkv=k.shape[1], B_global=cfg["B"])          # batch folded into heads (the layout of shard_bh)
    sm_scale = cfg["D"] ** -0.5
    ops = prequantize(cfg, q, k, v)
    torch.cuda.synchronize()

    wall_k, dev_k = timed(lambda: kernel_only_step(cfg, ops, sm_scale), args.steps, args.warmup, dist_on, args.ramp_seconds)
    wall_e, dev_e = timed(lambda: e2e_step(cfg, q, k, v), max(3, args.steps // 2), 2, dist_on, args.ramp_seconds)
    e2e_steps = max(3, args.steps // 2)
    prepass = prepass_roofline(cfg, k, v, args.config, between=lambda: kernel_only_step(cfg, ops, sm_scale))

    stats = torch.tensor([wall_k, wall_e], dtype=torch.float64, device=device)
    per_rank_ms = None
    if dist_on:
        stats, per_rank_ms = reduce_times(stats, world, args.steps)
    wall_k, wall_e = stats.tolist()

    fl = flops(cfg)
    ms_per_step = wall_k / args.steps * 1e3
    value = fl * world / (wall_k / args.steps) / 1e12
    kern_ms = sum(dev_k) / len(dev_k)                     # average launch duration (HIP events)

    out = {
        "metric": "attention TFLOPS (%s, hd=%d), kernel-only, as published by the reference" % ("causal" if cfg["causal"] else "non-causal", cfg["D"]),
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "clock_ramp_seconds": args.ramp_seconds,
        "vs_baseline": round(value / world / 795.0, 4) if args.config == "c3" else None,
        "vs_baseline_note": "per-GPU kernel-only TFLOPS / 795 (SageAttn2-8b, H100, hd128 causal N=8k; BASELINE.md section 1)",
        "dtype": "int8 QK^T + " + ("fp8(e4m3) PV" if cfg["pv"] == "fp8" else "fp16 PV") + ", fp32 accumulate",
        "fp8_score_form": _score_form() if cfg["pv"] == "fp8" else None,
        "fp8_score_form_note": "exact = exp2(fma(s, c, -m)) as the reference's kernels evaluate it (attn_utils.cuh:445-449): the default of every entry point",
        "data": "synthetic (randn, quantised by the product's own pre-pass kernels)",
        "config": {"workload": cfg["workload"], "global_batch": cfg["B_global"] * world, "heads": CONFIGS[args.config]["H"], "seq_len": cfg["N"],
                   "head_dim": cfg["D"], "parallelism": f"batch*head shard x{world} (shard.shard_range units of the global batch), no collective"},
        "roofline": roofline_obj(fl, kern_ms, cfg["pv"], "sage_attn_kernel", args.config),
        "end_to_end": {"ms_per_call": round(wall_e / e2e_steps * 1e3, 4),
                       "tflops": round(fl * world / (wall_e / e2e_steps) / 1e12, 2),
                       "what": "sageattn(): K mean + INT8 Q/K quant + V pre-pass + attention",
                       "prepass": prepass},
    }
    if cfg["pv"] == "fp8" and world == 1:
        out["fp8_folded_variant"] = folded_variant(cfg, ops, fl, max(10, args.steps // 2), 5, min(args.ramp_seconds, 0.2))
    if ranks_seen is not None:
        out["ranks_seen"] = ranks_seen
        out["ms_per_step_per_rank"] = per_rank_ms
    if rank == 0:
        try:
            out["accuracy"] = accuracy(cfg, q, k, v)
        except Exception as e:   # accuracy is informational here; the gate is tests/ -m gpu
            out["accuracy"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        if (args.sweep or (args.config == "c3" and world == 1)) and not args.no_sweep:
            # BASELINE.json's metric is the whole curve "hd=128 causal at N=1k..32k", so the default run carries it: kernel-only,
            # the config's batch (2); --sweep adds batch 4 with the same per-thread scales (the reference bench script's own
            # per-warp batch-4 sweep is in the `configs` block)
            batches = [("sweep_kernel_only_tflops", cfg["B_global"])] + ([("sweep_kernel_only_tflops_batch4", 4)] if args.sweep else [])
            for key, bsz in batches:
                sweep = {}
                for n in (1024, 2048, 4096, 8192, 16384, 32768):
                    c = dict(CONFIGS[args.config], N=n, B=bsz)
                    qq, kk, vv = make_inputs(c, device, 99)
                    oo = prequantize(c, qq, kk, vv)
                    _, d = timed(lambda: kernel_only_step(c, oo, sm_scale), 20, 5, False, min(args.ramp_seconds, 0.2))
                    sweep[str(n)] = round(flops(c) / (_median(d) * 1e-3) / 1e12, 1)
                    del qq, kk, vv, oo
                out[key] = sweep
            # the reference's bench helper writes 256 MB between repetitions so that no launch finds its operands in a cache
            # (bench/utils.py:7-33); MI355X has a 256 MB Infinity Cache, which holds the operands of every sweep point up to N = 4k
            flush = torch.empty(int(256e6) // 4, dtype=torch.int32, device=device)
            cold = {}
            for n in (1024, 2048, 4096, 8192, 16384, 32768):
                c = dict(CONFIGS[args.config], N=n, B=cfg["B_global"])
                qq, kk, vv = make_inputs(c, device, 99)
                oo = prequantize(c, qq, kk, vv)
                timed(lambda: kernel_only_step(c, oo, sm_scale), 5, 5, False, min(args.ramp_seconds, 0.2))
                d = []
                for _ in range(12):
                    flush.zero_()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    kernel_only_step(c, oo, sm_scale)
                    b.record()
                    b.synchronize()
                    d.append(a.elapsed_time(b))
                cold[str(n)] = round(flops(c) / (_median(d) * 1e-3) / 1e12, 1)
                del qq, kk, vv, oo
            del flush
            out["sweep_kernel_only_tflops_cache_flushed"] = cold
            out["sweep_flushed_note"] = ("the same sweep with 256 MB written between launches (the reference's bench/utils.py:7-33 flush; MI355X's Infinity "
                                         "Cache is 256 MB): median of 12 single launches each, every launch timed on its own behind the flush")
            out["sweep_note"] = ("kernel-only TFLOP/s of the workload's kernel at N = 1k .. 32k (B=%d, H=%d, D=%d, %s): median of 20 launches each, HIP events"
                                 % (cfg["B_global"], CONFIGS[args.config]["H"], cfg["D"], "causal" if cfg["causal"] else "non-causal"))
        if args.config == "c3" and world == 1 and not args.no_configs:
            del q, k, v, ops
            torch.cuda.empty_cache()
            try:
                out["configs"] = other_configs(args, device)
            except Exception as e:            # the headline must not be lost to a secondary measurement
                out["configs"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
