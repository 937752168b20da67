#!/usr/bin/env python3
"""bench.py — decode tokens/s (+ p50 TTFT) of Llama-3-8B GPTQ-int4 (BASELINE.json configs[1]) on N
MI355X GPUs of one node, synthetic weights/prompts of that shape (SURVEY.md §8d recipe).

A "step" = one decode step of the running batch through the whole engine (scheduler, metadata upload,
hipGraph replay of the forward, argmax, token download).  Weights and KV cache are resident in HBM when
the timed region starts.  At N > 1 every rank is an independent replica (the 8B model fits one GPU:
north_star asks for TP only "where the model is too large") — no data-path collective, scaling "weak".

Prints ONE JSON line on rank 0.  Extra legs (not in the timed region): bs=32 throughput, p50 TTFT,
the per-kernel roofline of the dequant-GEMM family (HIP-event timed launches, rotating layers), and the
CPU oracle timed on the host cores (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="decode batch of the timed region (headline: 1)")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--model", default="llama3-8b-gptq")
    ap.add_argument("--no-extras", action="store_true", help="skip bs=32 / TTFT / roofline / cpu legs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (rocprofv3 kernel tracing crashes on hipGraph replay)")
    ap.add_argument("--blocks", type=int, default=8192, help="KV blocks (64 tokens each); 0 = kv_fraction of free HBM")
    return ap.parse_args()


def make_prompts(n, length, vocab, seed=42):
    import numpy as np
    r = np.random.default_rng(seed)
    return [r.integers(1000, vocab - 1000, size=length).astype("uint32") for _ in range(n)]


def run_decode(eng, prompts, warmup, steps, sync):
    """prefill the prompts (untimed), `warmup` decode steps (untimed), then time exactly `steps` steps."""
    rids = [eng.add_request(p, max_tokens=warmup + steps + 8, ignore_eos=True) for p in prompts]
    # all prompts through prefill: the scheduler alternates prefill and decode steps once something is running (A14),
    # so "the first decode step" is not enough for batches whose prompts exceed one 8192-token prefill step
    while True:
        n, is_prefill = eng.step()
        if not is_prefill and n == len(prompts):
            break
    done = 1
    while done < warmup:
        eng.step()
        done += 1
    sync()
    t0 = time.perf_counter()
    ms_events = eng.timed_decode(steps)
    sync()
    dt = time.perf_counter() - t0
    outs = [eng.output(r) for r in rids]
    while eng.has_unfinished():  # drain
        eng.step()
    return dt, ms_events, outs


def ttft_p50(eng, prompt_len, vocab, batch, reps=5):
    """TTFT = first-token time - creation time (engine.rs:1004-1012), `batch` requests submitted together"""
    import numpy as np
    vals = []
    for i in range(reps):
        prompts = make_prompts(batch, prompt_len, vocab, seed=100 + i)
        rids = [eng.add_request(p, max_tokens=2, ignore_eos=True) for p in prompts]
        while eng.has_unfinished():
            eng.step()
        for r in rids:
            t = eng.times(r)
            vals.append(t["first_token_ms"] - t["created_ms"])
    return float(np.median(vals))


def cpu_baseline(cfg):
    """the CPU oracle (kind "port": the reference has no int4 CPU path, src/utils/gptq.rs:212-222) timed
    on the host cores: one decode token through 2 of the 32 layers' seven int4 GEMMs, extrapolated."""
    import numpy as np
    from oracle import oracle as orc
    H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    Hq, Hkv, g = cfg["num_heads"], cfg["num_kv_heads"], cfg["group_size"]
    shapes = [(H, Hq * D), (H, Hkv * D), (H, Hkv * D), (Hq * D, H), (H, I), (H, I), (I, H)]
    n_layers_sample = 2
    mats = []
    for li in range(n_layers_sample):
        for si, (K, N) in enumerate(shapes):
            qw = orc.fill_hash_u32((K // 8) * N, 7 + li * 16 + si).reshape(K // 8, N)
            sc = orc.fill_uniform((K // g, N), 99 + si, 0.002, 0.02, 0)
            mats.append((K, N, qw, sc))
    xs = {K: orc.fill_normal((1, K), K, 0.0, 1.0, 0) for K in {s[0] for s in shapes}}
    orc.gptq_gemv_fast(xs[H], mats[0][2], mats[0][3], g, 0)  # warm the thread pool
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 10.0 or reps < 1:
        for K, N, qw, sc in mats:
            orc.gptq_gemv_fast(xs[K], qw, sc, g, 0)
        reps += 1
    per_layer = (time.perf_counter() - t0) / reps / n_layers_sample
    # lm_head (bf16 dense, 1 GB) is not in the sample; scale by its share of the per-token bytes
    quant_bytes = sum(K * N // 2 for K, N in shapes) * cfg["num_layers"]
    total = per_layer * cfg["num_layers"] * (1.0 + (cfg["vocab_size"] * H * 2) / quant_bytes)
    return dict(value=1.0 / total, unit="tokens/s", cores=int(os.environ.get("OMP_NUM_THREADS", "1")), kind="port",
                sample=f"bs=1 decode: {reps}x the 7 int4 GEMMs of {n_layers_sample}/32 Llama-3-8B layers via oracle/vra_oracle.c "
                       f"orc_gptq_gemv_fast (~{time.perf_counter() - t0:.0f}s), extrapolated to 32 layers + lm_head bytes")


def dist_init(world, local_rank, backend="nccl"):
    """one process per GPU (torchrun env); backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests."""
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist_mod
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist_mod.init_process_group(backend)
    return dist_mod


def max_over_ranks(dist, seconds):
    """the job's time is the slowest rank's time."""
    if dist is None:
        return seconds
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ncpu = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(ncpu // max(world, 1), 128))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    from vllm_rs_amd import _lib
    from vllm_rs_amd import engine as E
    L = _lib.load()
    dist = dist_init(world, local_rank)
    if L.vra_device_count() <= local_rank:
        raise SystemExit("bench.py needs a GPU: no HIP device for this rank (the product has no CPU fallback)")
    L.vra_set_device(local_rank)

    def sync():
        L.vra_device_sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    cfg = dict({"llama3-8b-gptq": E.LLAMA3_8B, "qwen2-7b-awq": E.QWEN2_7B, "llama3-70b-tp8-rank": E.LLAMA3_70B_TP8_RANK}[a.model])
    max_bs = max(32, a.batch)
    eng = E.Engine(cfg, max_num_seqs=max_bs, max_model_len=8192, num_gpu_blocks=a.blocks, use_graph=not a.no_graph, device=local_rank,
                   seed=1234 + rank).init_synthetic()
    V = cfg["vocab_size"]

    # ---------------- timed region: K decode steps at the headline batch
    dt, ms_events, outs = run_decode(eng, make_prompts(a.batch, a.prompt_len, V), a.warmup, a.steps, sync)
    dt = max_over_ranks(dist, dt)
    tokens = a.batch * a.steps * world  # weak scaling: every rank decodes its own batch, no data-path collective
    line = {
        "metric": "decode tokens/sec (+ p50 TTFT), Llama-3-8B int4, bs=1/32",
        "value": tokens / dt, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{a.model} shape (H{cfg['hidden_size']} L{cfg['num_layers']} Hq{cfg['num_heads']} Hkv{cfg['num_kv_heads']} "
                               f"D{cfg['head_dim']} I{cfg['intermediate_size']} V{cfg['vocab_size']}), int4 g128, TP=1 greedy decode, batch {a.batch} per GPU, "
                               f"prompt {a.prompt_len}, {a.steps} generated tokens, KV block 64, hipGraph replay; N>1 = independent replicas",
                   "batch_per_gpu": a.batch},
        "gpu_ms_per_step_events": ms_events / a.steps,
    }

    if rank == 0 and not a.no_extras:
        # ---------------- roofline of the dequant-GEMM family (bs = headline batch)
        fam = {0: "norm+qkv", 1: "o_proj+res", 2: "norm+gate_up+silu", 3: "down+res"}
        per = {}
        tot_b = tot_ms = 0.0
        for w, name in fam.items():
            ms = eng.bench_gemm(w, a.batch, 320)
            b = eng.gemm_bytes(w, a.batch)
            per[name] = {"ms": ms, "bytes": b, "GBps": b / ms / 1e6}
            tot_b += b
            tot_ms += ms
        dom = per["norm+gate_up+silu"]
        traffic = None  # HBM bytes per launch from the PMC pass (rocprofv3 --pmc FETCH_SIZE, corrected per the microarch guide)
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc.json")))
            traffic = pmc["kernels"]["gemv_q4_kernel<BF16,2,1,false>"]["hbm_bytes_per_launch"] if a.batch == 1 else None
        except Exception:
            pass
        line["roofline"] = {"bound": "hbm", "kernel": "gemv_q4_kernel<BF16,NBW=2,SPT=1,AWQ=false> (RMSNorm + gate/up int4 GEMV + SiLU*mul)", "achieved": dom["GBps"],
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                            "algorithmic_bytes_per_launch": dom["bytes"], "avg_launch_ms": dom["ms"],
                            "family": per, "family_GBps": tot_b / tot_ms / 1e6, "family_frac": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS,
                            "family_ms_per_token": tot_ms * cfg["num_layers"]}
        # ---------------- bs=32 decode + TTFT
        if a.batch != 32:
            dt32, _, _ = run_decode(eng, make_prompts(32, a.prompt_len, V, seed=43), 8, 64, lambda: L.vra_device_sync())
            line["bs32_tokens_per_s_per_gpu"] = 32 * 64 / dt32
            line["bs32_ms_per_step"] = dt32 * 1e3 / 64
        # ---------------- the KV term ("also ctx in {1k, 8k}"): decode at long contexts
        lc = {}
        for bs, ctx in ((1, 1024), (1, 8000), (32, 1024), (32, 4096)):
            dtl, _, _ = run_decode(eng, make_prompts(bs, ctx, V, seed=77 + ctx), 4, 16, lambda: L.vra_device_sync())
            kv = bs * (ctx + 10) * 131072  # bytes of K and V read per step (131 072 B per token and sequence)
            lc[f"bs{bs}_ctx{ctx}"] = {"tokens_per_s": bs * 16 / dtl, "ms_per_step": dtl * 1e3 / 16, "kv_GB_per_step": kv / 1e9}
        line["long_context_decode"] = lc
        line["ttft_p50_ms"] = {"bs1_prompt128": ttft_p50(eng, 128, V, 1), "bs32_prompt128": ttft_p50(eng, 128, V, 32, reps=2),
                               "bs1_prompt2048": ttft_p50(eng, 2048, V, 1, reps=3)}
        line["step_bytes_roofline"] = {"algorithmic_bytes_per_step": 3625975808 + 1050673152 + 532480,
                                       "frac_of_8TBps": (3625975808 + 1050673152 + 532480) / (dt / a.steps) / 8e12 if a.batch == 1 else None}
        if world == 1 and not a.no_cpu:
            line["cpu_baseline"] = cpu_baseline(cfg)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
