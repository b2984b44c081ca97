#!/usr/bin/env python3
"""Time the REFERENCE's own CPU-runnable path -- its Triton kernels, unmodified, under TRITON_INTERPRET=1 -- in the build container
(the only place /root/reference exists; the GPU box has neither the reference nor a reason to run this).  SURVEY.md 8d asks for
this number beside the GPU's; bench.py reads the artefact written here (profiles/ref_triton_cpu.json) and reports it as
cpu_baseline.reference_path, next to the OpenMP port it times live on the GPU box's own cores.

    python tools/ref_cpu_time.py            # writes profiles/ref_triton_cpu.json

Cases: BASELINE.json configs[0] in full (B=1 H=4 N=512 D=64 non-causal, qk_int8_pv_fp16 Triton path incl. its quantisers), and one
(batch, head) unit of the hd128 causal sweep at N=1024 (the interpreter executes one program at a time on one core: N=4096 would
take most of an hour).  The fp32 SDPA of the same shapes on the same cores is timed beside it.
"""
import json
import os
import sys
import time

os.environ["TRITON_INTERPRET"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_golden as ref                         # loads the reference's Triton modules by path (see its header)


def flops(B, H, N, D, causal):
    return 4.0 * B * H * N * N * D / (2 if causal else 1)


def best_of(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    best = float("inf")
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def port_seconds(B, H, N, D, causal):
    """The OpenMP port (oracle/sage_oracle.c, what bench.py times live on the GPU box's cores) on the same shape and the same cores:
    per-block scales, FP16 PV in the Triton kernels' form -- the arithmetic of the reference path timed beside it."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import oracle
    oracle.build()
    rng = np.random.default_rng(0)
    q8 = rng.integers(-95, 95, (B, H, N, D), dtype=np.int8)
    k8 = rng.integers(-95, 95, (B, H, N, D), dtype=np.int8)
    gq, nq = oracle.group_index(N, "per_block", "q", 128, 128)
    gk, nk = oracle.group_index(N, "per_block", "k", 64, 64)
    qs = (0.5 + rng.random((B, H, nq))).astype(np.float32) * 0.02
    ks = (0.5 + rng.random((B, H, nk))).astype(np.float32) * 0.02
    v = oracle.convert(rng.standard_normal((B, H, N, D)).astype(np.float32), "f16")
    return best_of(lambda: oracle.attn(q8, k8, v, qs, gq, ks, gk, causal=causal, c=1.0, pv_mode=oracle.PV_F16_TRITON, out_dtype=0), n=3, warm=1)


def case(name, B, H, N, D, causal, dtype):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, H, N, D, generator=g).to(dtype)
    k = torch.randn(B, H, N, D, generator=g).to(dtype)
    v = torch.randn(B, H, N, D, generator=g).to(dtype)
    t0 = time.perf_counter()
    o, _, _ = ref.ref_dense(q, k, v, causal, return_lse=False)       # (the interpreter is deterministic and slow: one run)
    dt = time.perf_counter() - t0
    qf, kf, vf = q.float(), k.float(), v.float()
    truth = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=causal)
    dt_sdpa = best_of(lambda: torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=causal), n=3, warm=1)
    dt_port = port_seconds(B, H, N, D, causal)
    cos = torch.nn.functional.cosine_similarity(o.float().flatten(), truth.flatten(), dim=0).item()
    fl = flops(B, H, N, D, causal)
    return {"case": name, "shape": {"B": B, "H": H, "N": N, "D": D, "causal": causal, "dtype": str(dtype).split(".")[-1]},
            "reference_triton_interpreter_seconds": round(dt, 2), "reference_triton_interpreter_gflops": round(fl / dt / 1e9, 4),
            "fp32_sdpa_cpu_seconds": round(dt_sdpa, 4), "fp32_sdpa_cpu_gflops": round(fl / dt_sdpa / 1e9, 2),
            "openmp_port_seconds": round(dt_port, 4), "openmp_port_gflops": round(fl / dt_port / 1e9, 2),
            "timing": "reference: one run; fp32 SDPA and the OpenMP port: best of 3 after one warm-up call, same cores",
            "cos_sim_vs_fp32_sdpa": round(cos, 6)}


def main():
    out = {"what": "thu-ml/SageAttention sageattn_qk_int8_pv_fp16_triton (per-block INT8 quantisers + attention kernels, unmodified) "
                   "under TRITON_INTERPRET=1 on the build container's host cores; host glue restated in tests/golden/gen_golden.py",
           "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
           "cases": []}
    out["cases"].append(case("BASELINE.json configs[0]", 1, 4, 512, 64, False, torch.float16))
    out["cases"].append(case("one (batch, head) unit of the hd128 causal sweep at N=1024", 1, 1, 1024, 128, True, torch.float16))
    out["cases"].append(case("one (batch, head) unit of the hd128 causal sweep at N=4096", 1, 1, 4096, 128, True, torch.float16))
    path = os.path.join(ROOT, "profiles", "ref_triton_cpu.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
