#!/usr/bin/env python3
"""bench.py — decode tokens/s (+ p50 TTFT) of Llama-3-8B GPTQ-int4 (BASELINE.json configs[1]) on N MI355X GPUs of one
node, synthetic weights/prompts of that shape (SURVEY.md §8d recipe).

A "step" = one decode step of the running batch through the whole engine (scheduler, metadata upload, hipGraph replay of
the forward, argmax, token download).  Weights and KV cache are resident in HBM when the timed region starts.  At N > 1
every rank is an independent replica (the 8B model fits one GPU: north_star asks for TP only "where the model is too
large") — no data-path collective, scaling "weak".  `--gpus N` run directly spawns its own N ranks (one process per
GPU); under `python -m torch.distributed.run` the launcher's RANK / LOCAL_RANK / WORLD_SIZE are used.  The barrier and the
max-over-ranks of the contract run over a Unix socket between the ranks of the node (NodeRendezvous): no PyTorch in the
harness of a no-PyTorch product.  `--tp N` instead runs ONE tensor-parallel engine over N ranks
(BASELINE config 4's mechanics: RCCL + one-shot all-reduce, vllm_rs_amd/runner.py) and reports its tokens/s.

Prints ONE JSON line on rank 0.  Extra legs (not in the timed region): the roofline of the dequant-GEMM family (HIP-event
timed launches, rotating layers), bs=32 throughput, long-context decode, p50 TTFT (128 / 2048 / 32768-token prompts, the
last one cold and with a prefix-cache hit: config 5), the reference-binding path (`ffi_path`: the GEMMs of a layer through
marlin_4bit_bf16 + vra_rms_norm + vra_silu_mul as the Rust layers would issue them), a Qwen2-7B-AWQ decode line (config 3)
and the CPU oracle timed on the host cores (rank 0, N == 1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300-6900 GB/s is the measured streaming ceiling
FAMILY = {0: "norm+qkv", 1: "o_proj+res", 2: "norm+gate_up+silu", 3: "down+res"}
KERNEL_OF = {"norm+qkv": "gemv_q4s_kernel<BF16,1,false>", "o_proj+res": "gemv_q4s_kernel<BF16,1,false>",
             "norm+gate_up+silu": "gemv_q4s_kernel<BF16,2,false>", "down+res": "gemv_q4s_kernel<BF16,1,false>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="decode batch of the timed region (headline: 1)")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--model", default="llama3-8b-gptq")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / bs=32 / TTFT / ffi / qwen / cpu legs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-depth parity leg (the oracle over all layers, ~20 s)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (rocprofv3 kernel tracing crashes on hipGraph replay)")
    ap.add_argument("--blocks", type=int, default=8192, help="KV blocks (64 tokens each); 0 = kv_fraction of free HBM")
    ap.add_argument("--tp", type=int, default=0, help="run ONE tensor-parallel engine over this many ranks (spawns its own runner processes)")
    return ap.parse_args()


def make_prompts(n, length, vocab, seed=42):
    import numpy as np
    r = np.random.default_rng(seed)
    return [r.integers(1000, vocab - 1000, size=length).astype("uint32") for _ in range(n)]


def run_decode(eng, prompts, warmup, steps, sync):
    """prefill the prompts (untimed), `warmup` decode steps (untimed), then time exactly `steps` steps."""
    rids = [eng.add_request(p, max_tokens=warmup + steps + 8, ignore_eos=True) for p in prompts]
    # all prompts through prefill: the scheduler alternates prefill and decode steps once something is running (A14),
    # so "the first decode step" is not enough for batches whose prompts exceed one prefill step
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


def ttft_p50(eng, prompt_len, vocab, batch, reps=5, seed0=100):
    """TTFT = first-token time - creation time (engine.rs:1004-1012), `batch` requests submitted together"""
    import numpy as np
    vals = []
    for i in range(reps):
        prompts = make_prompts(batch, prompt_len, vocab, seed=seed0 + i)
        rids = [eng.add_request(p, max_tokens=2, ignore_eos=True) for p in prompts]
        while eng.has_unfinished():
            eng.step()
        for r in rids:
            t = eng.times(r)
            vals.append(t["first_token_ms"] - t["created_ms"])
    return float(np.median(vals))


def lib_sha16():
    from vllm_rs_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]


def src_sha16():
    """hash of the sources the library is built from (csrc + host + the header): unlike the .so hash it survives a rebuild
    on another checkout, and any kernel edit changes it"""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "vllm_rs_amd", "csrc", "*.*")) + glob.glob(os.path.join(ROOT, "vllm_rs_amd", "host", "*.*")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if f.endswith((".hip", ".cuh", ".h", ".cpp")) or os.path.basename(f) == "Makefile":
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE pass — ONLY if that pass was taken
    with a library built from these very sources (the source hash, and the .so hash of that build, are stored next to the
    counters): a kernel change can never leave a stale number in the line."""
    try:
        pmc = load_pmc()
        if pmc is None:
            return None
        if kernel == "family":  # the four launches of a layer: norm+qkv, o_proj, down (<..,1,..>) and the gate/up pair (<..,2,..>)
            return 3 * pmc["kernels"]["gemv_q4s_kernel<BF16,1,false>"]["hbm_bytes_per_launch"] + pmc["kernels"]["gemv_q4s_kernel<BF16,2,false>"]["hbm_bytes_per_launch"]
        return pmc["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def load_pmc():
    """the committed counter / trace pass of this round (profiles/r06_pmc.json, tools/summarise_profiles.py) — only if it was taken
    with a library built from these very sources"""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc.json")))
    except Exception:
        return None
    if pmc.get("src_sha16") != src_sha16() and pmc.get("lib_sha16") != lib_sha16():
        return None
    return pmc


def in_situ_family():
    """the dequant-GEMV family INSIDE the decode step: average launch durations of the two kernel-E instantiations in
    the rocprofv3 kernel trace of the eager bs-1 step (profiles/r06_decode_bs1_kernel_trace.txt) — 3 launches of <BF16,1,false> and
    one of <BF16,2,false> per layer — next to the isolated-launch figure bench.py times itself"""
    pmc = load_pmc()
    try:
        t = pmc["in_situ_decode_bs1_kernel_trace"]
        us = 3 * t["gemv_q4s_kernel<BF16,1,false>"]["avg_us"] + t["gemv_q4s_kernel<BF16,2,false>"]["avg_us"]
        return {"us_per_layer": us, "GBps": 113475584 / us / 1e3, "frac": 113475584 / us / 1e3 / HBM_PEAK_GBS,
                "source": "profiles/r06_decode_bs1_kernel_trace.txt (rocprofv3 --kernel-trace of the eager step, same sources)"}
    except Exception:
        return None


def ffi_path(L, cfg, iters=6):
    """the reference's own binding path: what the Rust layers issue for ONE decode token through the seven
    symbols of src/utils/gptq.rs:3-6 — per layer NormX, marlin_4bit_bf16 x (q, k, v, o), +residual, NormX, marlin x (gate, up),
    silu*mul, marlin (down), +residual; Marlin-permuted scales, caller-owned workspace, no fusion across calls.  Attention
    is not on this path's GEMM count and is left out.  Distinct weights per layer (8 layer sets rotate: nothing is cache
    resident).  Returns ms per token (32 layers) measured with HIP events on one stream."""
    import ctypes as C
    H, I, D, Hq, Hkv, g = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"], cfg["num_heads"], cfg["num_kv_heads"], cfg["group_size"]
    P = C.c_void_p
    shapes = dict(q=(H, Hq * D), k=(H, Hkv * D), v=(H, Hkv * D), o=(Hq * D, H), gate=(H, I), up=(H, I), down=(I, H))
    nset = 8
    sets, allocs = [], []

    def dalloc(nbytes):
        p = L.vra_malloc(nbytes)
        allocs.append(p)
        return p
    for s in range(nset):
        lw = {}
        for i, (name, (K, N)) in enumerate(shapes.items()):
            w, sc = dalloc(K * N // 2), dalloc(K // g * N * 2)
            L.vra_fill_hash_u32(w, K * N // 8, 1000 + s * 16 + i, 0)
            L.vra_fill_uniform(sc, K // g * N, 2000 + s * 16 + i, 0.002, 0.02, 0, 0)
            lw[name] = (w, sc, K, N)
        lw["n1"], lw["n2"] = dalloc(H * 2), dalloc(H * 2)
        L.vra_fill_normal(lw["n1"], H, 7 + s, 1.0, 0.02, 0, 0)
        L.vra_fill_normal(lw["n2"], H, 9 + s, 1.0, 0.02, 0, 0)
        sets.append(lw)
    h, xn, q, k, v, att, t1, gt, up, act = (dalloc(n * 2) for n in (H, H, Hq * D, Hkv * D, Hkv * D, Hq * D, H, I, I, I))
    ws = dalloc(I * 4)
    L.vra_memset(ws, 0, I * 4, 0)
    L.vra_fill_normal(h, H, 3, 0.0, 1.0, 0, 0)
    L.vra_fill_normal(att, Hq * D, 4, 0.0, 1.0, 0, 0)

    def mm(x, lwt, out):
        w, sc, K, N = lwt
        L.marlin_4bit_bf16(x, w, sc, None, None, out, 1, K, N, ws, g, 0)

    def layer(lw):
        L.vra_rms_norm(h, lw["n1"], xn, 1, H, 1e-5, 0, 0)
        mm(xn, lw["q"], q), mm(xn, lw["k"], k), mm(xn, lw["v"], v)
        mm(att, lw["o"], t1)
        L.vra_add(t1, h, h, H, 0, 0)
        L.vra_rms_norm(h, lw["n2"], xn, 1, H, 1e-5, 0, 0)
        mm(xn, lw["gate"], gt), mm(xn, lw["up"], up)
        L.vra_silu_mul(gt, up, act, I, 0, 0)
        mm(act, lw["down"], t1)
        L.vra_add(t1, h, h, H, 0, 0)
    for lw in sets:
        layer(lw)
    L.vra_device_sync()
    e0, e1 = L.vra_event_create(), L.vra_event_create()
    L.vra_event_record(e0, 0)
    for _ in range(iters):
        for lw in sets:
            layer(lw)
    L.vra_event_record(e1, 0)
    ms = L.vra_event_elapsed_ms(e0, e1)
    err = L.vra_last_error().decode()
    for p in allocs:
        L.vra_free(p)
    L.vra_event_destroy(e0), L.vra_event_destroy(e1)
    if err:
        return {"error": err}
    per_layer = ms / (iters * nset)
    return {"ms_per_token_gemm_path": per_layer * cfg["num_layers"], "us_per_layer": per_layer * 1e3, "launches_per_layer": 12,
            "note": "7 marlin_4bit_bf16 + 2 vra_rms_norm + vra_silu_mul + 2 vra_add per layer, Marlin-permuted scales read in place (no conversion "
                    "launch), eager launches on the null stream; attention excluded"}


def roofline_prefill(eng, L, cfg, rows=4096):
    """the MFMA roofline of the prefill kernels — the int4 GEMMs of a layer at `rows` activation rows, per launch, timed with
    HIP events (vra_engine_bench_gemm: rotating layers), and the paged prefill attention (prefill_attn_kernel) on one sequence of `rows`
    tokens.  From 768 rows on a GEMM is the dequant pass (dequant_frag_kernel: w = rnd((q - z) * s), Marlin's weight, gptq.rs:116-178)
    plus the 256-row dense GEMM (gemm_dense_kernel, csrc/gemm_dense.cuh): `kernels` times BOTH launches of every GEMM, so `achieved` is
    the rate a prefill sees; `int4_fused_kernel_d` is the same layer with the path switched off (kernel D: gemm_q4_big_kernel, the exact
    product with the conversion inside every 64-row tile).  FLOPs: 2*M*K*N per GEMM; 4*D*Hq*(T*(T+1)/2) for the causal attention.
    Peak: 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)."""
    import numpy as np
    from vllm_rs_amd import ops
    H, I, D, Hq, Hkv = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"], cfg["num_heads"], cfg["num_kv_heads"]
    PEAK = 2500.0
    out = {"bound": "mfma", "peak": PEAK, "unit": "TFLOP/s", "rows": rows, "kernels": {}}
    shapes = {"norm+qkv": (0, H, (Hq + 2 * Hkv) * D), "o_proj+res": (1, Hq * D, H), "gate_up+silu": (2, H, 2 * I), "down+res": (3, I, H)}

    def layer(dst):
        tot_fl = tot_ms = 0.0
        for name, (w, K, N) in shapes.items():
            ms = eng.bench_gemm(w, rows, 6)
            fl = 2.0 * rows * K * N
            tot_fl += fl
            tot_ms += ms
            dst[name] = {"ms": ms, "TFLOPs": fl / ms / 1e9, "frac": fl / ms / 1e9 / PEAK}
        return tot_fl / tot_ms / 1e9

    dense_rows = L.vra_debug_dense_prefill_min_rows()
    out["dense_path_from_rows"] = dense_rows
    out["achieved"] = layer(out["kernels"])
    out["frac"] = out["achieved"] / PEAK
    if 0 < dense_rows <= rows:
        out["kernel"] = ("dequant_frag_kernel + gemm_dense_kernel<BF16,256|128,*> (both launches of each of a layer's four GEMMs; norm+qkv and "
                         "gate_up+silu include their rms_norm launch)")
        L.vra_debug_set_dense_prefill_min_rows(0)
        try:
            d = {}
            ach = layer(d)
            out["int4_fused_kernel_d"] = {"kernel": "gemm_q4_big_kernel<BF16,*,false,4>", "kernels": d, "achieved": ach, "frac": ach / PEAK}
        finally:
            L.vra_debug_set_dense_prefill_min_rows(dense_rows)
    else:
        out["kernel"] = "gemm_q4_big_kernel<BF16,*,false,4> (the four int4 GEMMs of a layer; norm+qkv includes its rms_norm launch)"
    # prefill attention
    BS = 64
    T = rows
    nb = (T + BS - 1) // BS
    r = np.random.default_rng(0)
    att = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, ops.BF16)
    # (random q / K / V: constant data clocks higher and collapses the softmax; ONE output buffer: ops.PagedAttention.forward_prefill
    # allocates and frees a buffer per call — a device synchronisation per launch that rounds 1-5 of this leg timed along with the kernel)
    q, kc, vc = ops.DevBuf(T * Hq * D * 2), ops.DevBuf(nb * Hkv * BS * D * 2), ops.DevBuf(nb * Hkv * BS * D * 2)
    L.vra_fill_normal(q.ptr, T * Hq * D, 31, 0.0, 1.0, 0, 0)
    L.vra_fill_normal(kc.ptr, nb * Hkv * BS * D, 32, 0.0, 1.0, 0, 0)
    L.vra_fill_normal(vc.ptr, nb * Hkv * BS * D, 33, 0.0, 1.0, 0, 0)
    o = ops.DevBuf(T * Hq * D * 2)
    bt, cl, cu = ops.dev(r.permutation(nb).astype(np.uint32)), ops.dev(np.array([T], np.uint32)), ops.dev(np.array([0, T], np.uint32))
    e0, e1 = L.vra_event_create(), L.vra_event_create()

    def launch():
        L.vra_paged_attention_prefill_sw(o.ptr, q.ptr, None, None, kc.ptr, vc.ptr, bt.ptr, cl.ptr, cu.ptr, None, 1, T, T, Hq, Hkv, D, BS, nb, att.scale,
                                         0.0, 0, ops.BF16, ops.BF16, 0)

    for _ in range(2):
        launch()
    ops.check_error()
    L.vra_device_sync()
    n = 10
    L.vra_event_record(e0, 0)
    for _ in range(n):
        launch()
    L.vra_event_record(e1, 0)
    ms = L.vra_event_elapsed_ms(e0, e1) / n
    L.vra_event_destroy(e0), L.vra_event_destroy(e1)
    fl = 4.0 * D * Hq * (T * (T + 1) / 2)
    out["prefill_attn"] = {"kernel": "prefill_attn_kernel<BF16,128,false,2>", "tokens": T, "ms": ms, "TFLOPs": fl / ms / 1e9, "frac": fl / ms / 1e9 / PEAK}
    out["note"] = ("why kernel D stops where it does: profiles/r06_kernel_d_probes.txt (the same tiling without the int4 -> bf16 conversion runs "
                   "1.08 PFLOP/s; the conversion and the per-group scale fix-up are VALU work the wave has no free issue slots for) — hence the "
                   "conversion once per GEMM instead of once per 64-row tile (profiles/r06_gemm_dense_microbench.txt)")
    return out


def ffi_step(L, cfg, ctx=150, replays=24):
    """A WHOLE decode step (bs 1, context `ctx`) issued call by call the way the reference's Rust layers issue it (VERDICT r5 #7a):
    embedding; per layer NormX, marlin_4bit_bf16 x (q, k, v) (gptq.rs:116-178), the rotary embedding (rotary_emb.rs:88-103),
    reshape_and_cache + paged attention (attention.rs:745-820), marlin (o), + residual, NormX, marlin x (gate, up), silu * mul,
    marlin (down), + residual; final norm, lm_head, argmax (llama.rs:269-321) — 17 launches per layer, nothing fused across calls,
    Marlin-permuted scales, 32 layers with their own weights.  Timed twice: eager from this (ctypes) host, and captured ONCE into a
    hipGraph by the host and replayed, as the reference replays its decode graphs (runner.rs graph capture).  HIP events on the stream."""
    import ctypes as C
    import numpy as np
    from vllm_rs_amd import ops
    H, I, D, Hq, Hkv, g, V, NL = (cfg[k] for k in ("hidden_size", "intermediate_size", "head_dim", "num_heads", "num_kv_heads", "group_size", "vocab_size", "num_layers"))
    BS, nblk = 64, (ctx + 63) // 64
    shapes = dict(q=(H, Hq * D), k=(H, Hkv * D), v=(H, Hkv * D), o=(Hq * D, H), gate=(H, I), up=(H, I), down=(I, H))
    allocs = []

    def dalloc(nbytes):
        ptr = L.vra_malloc(nbytes)
        allocs.append(ptr)
        return ptr
    layers = []
    for l in range(NL):
        lw = {}
        for i, (name, (K, N)) in enumerate(shapes.items()):
            w, sc = dalloc(K * N // 2), dalloc(K // g * N * 2)
            L.vra_fill_hash_u32(w, K * N // 8, 5000 + l * 16 + i, 0)
            L.vra_fill_uniform(sc, K // g * N, 6000 + l * 16 + i, 0.002, 0.02, 0, 0)
            lw[name] = (w, sc, K, N)
        lw["n1"], lw["n2"] = dalloc(H * 2), dalloc(H * 2)
        L.vra_fill_normal(lw["n1"], H, 7 + l, 1.0, 0.02, 0, 0)
        L.vra_fill_normal(lw["n2"], H, 90 + l, 1.0, 0.02, 0, 0)
        cache = nblk * Hkv * BS * D
        lw["kc"], lw["vc"] = dalloc(cache * 2), dalloc(cache * 2)
        L.vra_fill_normal(lw["kc"], cache, 11 + l, 0.0, 1.0, 0, 0)
        L.vra_fill_normal(lw["vc"], cache, 12 + l, 0.0, 1.0, 0, 0)
        layers.append(lw)
    embed, lm_head, fnorm = dalloc(V * H * 2), dalloc(V * H * 2), dalloc(H * 2)
    L.vra_fill_normal(embed, V * H, 1, 0.0, 0.02, 0, 0)
    L.vra_fill_normal(lm_head, V * H, 2, 0.0, 0.02, 0, 0)
    L.vra_fill_normal(fnorm, H, 3, 1.0, 0.02, 0, 0)
    h, xn, q, k, v, att, t1, gt, up, act = (dalloc(n * 2) for n in (H, H, Hq * D, Hkv * D, Hkv * D, Hq * D, H, I, I, I))
    logits, tok = dalloc(V * 4), dalloc(64)
    ws = dalloc(I * 4)
    L.vra_memset(ws, 0, I * 4, 0)
    am_ws = dalloc(L.vra_dense_gemm_argmax_workspace_bytes())
    L.vra_memset(am_ws, 0, L.vra_dense_gemm_argmax_workspace_bytes(), 0)
    cos, sin = dalloc(8192 * (D // 2) * 2), dalloc(8192 * (D // 2) * 2)
    L.vra_fill_uniform(cos, 8192 * (D // 2), 21, -1.0, 1.0, 0, 0)
    L.vra_fill_uniform(sin, 8192 * (D // 2), 22, -1.0, 1.0, 0, 0)
    ids = ops.dev(np.array([1234], np.uint32))
    pos = ops.dev(np.array([ctx - 1], np.int64))
    slot = ops.dev(np.array([((ctx - 1) // BS) * BS + (ctx - 1) % BS], np.int64))
    bt = ops.dev(np.arange(nblk, dtype=np.uint32)[None])
    cl = ops.dev(np.array([ctx], np.uint32))
    attn_ws = dalloc(L.vra_paged_attention_decode_workspace_bytes(1, Hq, D, ctx))
    scale = 1.0 / float(np.sqrt(D))

    def step(st):
        L.vra_embedding(ids.ptr, embed, h, 1, H, V, 0, st)
        for lw in layers:
            L.vra_rms_norm(h, lw["n1"], xn, 1, H, 1e-5, 0, st)
            for name, out in (("q", q), ("k", k), ("v", v)):
                w, sc, K, N = lw[name]
                L.marlin_4bit_bf16(xn, w, sc, None, None, out, 1, K, N, ws, g, st)
            L.vra_fused_rope(q, k, cos, sin, pos.ptr, 1, Hq, Hkv, D, D, 0, 0, 0, st)
            L.vra_reshape_and_cache(k, v, lw["kc"], lw["vc"], slot.ptr, 1, Hkv, D, BS, 0, 0, st)
            L.vra_paged_attention_decode(att, q, lw["kc"], lw["vc"], bt.ptr, cl.ptr, 1, Hq, Hkv, D, BS, nblk, ctx, scale, 0.0, attn_ws, 0, 0, st)
            w, sc, K, N = lw["o"]
            L.marlin_4bit_bf16(att, w, sc, None, None, t1, 1, K, N, ws, g, st)
            L.vra_add(t1, h, h, H, 0, st)
            L.vra_rms_norm(h, lw["n2"], xn, 1, H, 1e-5, 0, st)
            for name, out in (("gate", gt), ("up", up)):
                w, sc, K, N = lw[name]
                L.marlin_4bit_bf16(xn, w, sc, None, None, out, 1, K, N, ws, g, st)
            L.vra_silu_mul(gt, up, act, I, 0, st)
            w, sc, K, N = lw["down"]
            L.marlin_4bit_bf16(act, w, sc, None, None, t1, 1, K, N, ws, g, st)
            L.vra_add(t1, h, h, H, 0, st)
        L.vra_rms_norm(h, fnorm, xn, 1, H, 1e-5, 0, st)
        L.vra_dense_gemm_argmax(xn, lm_head, None, logits, tok, am_ws, 1, H, V, 0, st)

    out = {"launches_per_step": 17 * NL + 3, "context": ctx}
    try:
        st = L.vra_stream_create()
        e0, e1 = L.vra_event_create(), L.vra_event_create()
        for _ in range(2):
            step(st)
        L.vra_device_sync()
        L.vra_event_record(e0, st)
        for _ in range(4):
            step(st)
        L.vra_event_record(e1, st)
        out["eager_ms_per_step"] = L.vra_event_elapsed_ms(e0, e1) / 4
        # the host's own graph capture (plain HIP runtime calls, as a Rust host would make them through hip-sys)
        hip = C.CDLL("libamdhip64.so")
        graph, gexec = C.c_void_p(), C.c_void_p()
        sp = C.c_void_p(st)
        assert hip.hipStreamBeginCapture(sp, 0) == 0
        step(st)
        assert hip.hipStreamEndCapture(sp, C.byref(graph)) == 0
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
        for _ in range(3):
            assert hip.hipGraphLaunch(gexec, sp) == 0
        L.vra_device_sync()
        L.vra_event_record(e0, st)
        for _ in range(replays):
            hip.hipGraphLaunch(gexec, sp)
        L.vra_event_record(e1, st)
        out["graph_ms_per_step"] = L.vra_event_elapsed_ms(e0, e1) / replays
        hip.hipGraphExecDestroy(gexec), hip.hipGraphDestroy(graph)
        L.vra_event_destroy(e0), L.vra_event_destroy(e1)
        err = L.vra_last_error().decode()
        if err:
            out["error"] = err
    finally:
        L.vra_device_sync()
        for ptr in allocs:
            L.vra_free(ptr)
    out["note"] = ("one decode token through the seven FFI symbols + the section-B ops, attention included, nothing fused across calls; "
                   "graph = the same sequence captured once by the host and replayed; compare ms_per_step of the native engine")
    return out


def runner_ipc_step(cfg, steps=96, warmup=8, prompt_len=128):
    """A decode step through the literal drop-in path (VERDICT r5 #7b): this process plays the reference's ENGINE (src/core/engine.rs:
    300-378 handshake, :844-892 RunPrefill / RunDecode) over the reference's framing (src/runner/mod.rs:246-295: abstract-namespace
    Unix socket, `ready`, JSON Init, bincode frames, 1-byte acks) against the native `vra_runner` process with synthetic weights of the
    named shape.  One bincode round trip per step, as engine.rs:849-885 does; wall-clock per step on the engine side."""
    import socket
    import subprocess
    from vllm_rs_amd import wire
    runner = os.path.join(ROOT, "vllm_rs_amd", "vra_runner")
    if not os.path.exists(runner):
        return {"error": "vra_runner not built"}
    uuid = f"bench{os.getpid()}"
    name = f"vra-bench-{os.getpid()}"
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind("\0" + name)
    srv.listen(1)
    srv.settimeout(600)
    hb_srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    hb_srv.bind("\0command_" + uuid + "@vllm-rs-runner-heartbeat.sock")
    hb_srv.listen(1)
    hb_srv.settimeout(60)
    hf = dict(architectures=["LlamaForCausalLM"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
              num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"], num_key_value_heads=cfg["num_kv_heads"],
              head_dim=cfg["head_dim"], vocab_size=cfg["vocab_size"], max_position_embeddings=cfg["max_position_embeddings"],
              rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"], torch_dtype="bfloat16", tie_word_embeddings=False,
              quantization_config=dict(quant_method="gptq", bits=4, group_size=cfg["group_size"], desc_act=False, sym=True))
    proc = subprocess.Popen([runner, "--sock", name, "--uuid", uuid], cwd=ROOT)
    conn = hb = None
    try:
        conn, _ = srv.accept()
        conn.settimeout(600)
        assert wire._recv_exact(conn, 6) == b"ready\n"
        hb, _ = hb_srv.accept()
        hb.settimeout(60)
        assert wire._recv_exact(hb, 6) == b"ready\n"
        init = dict(rank=0, dev_id=0, num_shards=1, model_type="LLaMa", dtype="BF16", is_gguf=False, is_rope_i=False, config=hf,
                    econfig=dict(block_size=64, max_num_seqs=8, num_blocks=128, max_model_len=2048, seed=5),
                    model_pathes=dict(config_filename="/nonexistent/config.json", filenames=[]))  # no checkpoint: synthetic weights (bench mode)
        wire.send_frame(conn, wire.encode_init_json(init))
        assert wire.decode(wire.recv_frame(conn)) == ("InitAck", True)
        nblocks = 64
        ecfg = dict(model_id=None, weight_path="/nonexistent", weight_file=None, enforce_parser=None, hf_token=None, hf_token_path=None, num_blocks=nblocks,
                    kv_fraction=0.5, mamba_fraction=None, cpu_mem_fold=0.0, kvcache_memory_bytes=0, mamba_memory_bytes=0, mamba_slot_bytes=0,
                    mamba_cache_capacity=None, block_size=64, max_num_seqs=8, max_num_batched_tokens=2048, config_model_len=2048, max_model_len=2048,
                    max_tokens=None, isq=None, num_shards=1, device_ids=[0], generation_cfg=None, seed=5, prefix_cache=False, prefix_cache_max_tokens=None,
                    fp8_kvcache=False, server_mode=False, pd_config=None, mcp_command=None, mcp_config=None, mcp_args=None, tool_prompt_template=None,
                    pd_server_prefix_cache_ratio=None, pd_client_prefix_cache_ratio=None, yarn_scaling_factor=None, disable_reasoning=False)
        wire.send_frame(conn, wire.encode_usable_memory_left_json(ecfg))
        assert wire.decode(wire.recv_frame(conn)) == ("InitAck", True)
        greedy = dict(temperature=0.0)
        toks = [int(t) for t in make_prompts(1, prompt_len, cfg["vocab_size"], seed=3)[0]]
        table = list(range((prompt_len + 63) // 64))
        seq = dict(id=1, token_ids=toks, block_table=table, num_cached_tokens=0, sampling_params=greedy, status="Running")
        wire.send_frame(conn, wire.encode(("RunPrefill", ([seq], True))))
        _, out = wire.decode(wire.recv_frame(conn))
        if not out:
            return {"error": "the runner answered the prefill with an empty RunResponse"}
        toks.append(int(out[0]))
        t0 = 0.0
        for i in range(warmup + steps):
            if i == warmup:
                t0 = time.perf_counter()
            if (len(toks) + 63) // 64 > len(table):
                table.append(len(table))
            d = dict(id=1, last_token=toks[-1], len=len(toks), last_block_tokens=len(toks) - (len(table) - 1) * 64, block_table_last=table[-1],
                     block_tables=table, sampling_params=greedy)
            wire.send_frame(conn, wire.encode(("RunDecode", ([d], False))))
            _, out = wire.decode(wire.recv_frame(conn))
            if not out:
                return {"error": f"empty RunResponse at decode step {i}"}
            toks.append(int(out[0]))
        dt = time.perf_counter() - t0
        wire.send_frame(conn, wire.encode(("Shutdown", None)))
        return {"ms_per_step": dt * 1e3 / steps, "steps": steps, "tokens_per_s": steps / dt,
                "note": "bs 1 greedy decode, one RunDecode/RunResponse bincode round trip per step over the abstract Unix socket, engine side in "
                        "Python; the runner replays its captured hipGraph and returns token ids (vra_engine_forward_tokens)"}
    except Exception as ex:
        return {"error": repr(ex)}
    finally:
        for c in (conn, hb, srv, hb_srv):
            try:
                if c is not None:
                    c.close()
            except OSError:
                pass
        try:
            proc.wait(timeout=20)
        except Exception:
            proc.kill()


def oracle_decode_tokens_per_s(cfg, layers_sample, n_tokens, prompt_len, label):
    """CPU baseline: a REAL greedy decode through oracle/model.py (all ops: norm, GEMMs, rope, paged attention, lm_head,
    argmax) on a model of the named shape with `layers_sample` of its layers; the per-layer time comes from the difference to
    the same run with 0 layers and is scaled to the full depth."""
    import numpy as np
    from oracle import model as om
    from oracle import oracle as orc
    H, I, V, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["head_dim"]
    Hq, Hkv, g, dt = cfg["num_heads"], cfg["num_kv_heads"], cfg.get("group_size", 128), cfg["dtype"]
    quant = cfg.get("quant_method") == "gptq"
    w = {"model.embed_tokens.weight": orc.fill_normal((V, H), 1, 0.0, 0.02, dt), "model.norm.weight": orc.fill_normal((H,), 2, 1.0, 0.02, dt),
         "lm_head.weight": orc.fill_normal((V, H), 3, 0.0, 0.02, dt)}
    for i in range(layers_sample):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = orc.fill_normal((H,), 10 + i, 1.0, 0.02, dt)
        w[p + "post_attention_layernorm.weight"] = orc.fill_normal((H,), 20 + i, 1.0, 0.02, dt)
        for j, (name, K, N) in enumerate((("self_attn.q_proj", H, Hq * D), ("self_attn.k_proj", H, Hkv * D), ("self_attn.v_proj", H, Hkv * D),
                                          ("self_attn.o_proj", Hq * D, H), ("mlp.gate_proj", H, I), ("mlp.up_proj", H, I), ("mlp.down_proj", I, H))):
            if quant:
                w[p + name + ".qweight"] = orc.fill_hash_u32((K // 8) * N, 100 + i * 16 + j).reshape(K // 8, N)
                w[p + name + ".scales"] = orc.fill_uniform((K // g, N), 300 + i * 16 + j, 0.0005, 0.002, dt)
            else:
                w[p + name + ".weight"] = orc.fill_normal((N, K), 100 + i * 16 + j, 0.0, 1.0 / np.sqrt(K), dt)

    def run(nl):
        c = dict(cfg, num_layers=nl, max_position_embeddings=min(cfg["max_position_embeddings"], 2048))
        m = om.OracleModel(c, w, num_blocks=8)
        ids = np.arange(1000, 1000 + prompt_len, dtype=np.uint32)
        pos = np.arange(prompt_len, dtype=np.int64)
        bt = np.arange(8, dtype=np.uint32)[None]
        logits = m.forward(ids, pos, pos.copy(), bt, [prompt_len], [0, prompt_len])  # prefill (untimed)
        t0 = time.perf_counter()
        n = prompt_len
        for _ in range(n_tokens):
            tok = orc.argmax_f32(logits).astype(np.uint32)
            logits = m.forward(tok, np.array([n], np.int64), np.array([n], np.int64), bt, [n + 1])
            n += 1
        return (time.perf_counter() - t0) / n_tokens
    t0 = time.perf_counter()
    t_head = run(0)
    t_part = run(layers_sample)
    per_layer = max(t_part - t_head, 0.0) / layers_sample
    total = t_head + per_layer * cfg["num_layers"]
    return dict(value=1.0 / total, unit="tokens/s", cores=int(os.environ.get("OMP_NUM_THREADS", "1")), kind="port",
                sample=f"{label}: {n_tokens} greedy tokens through oracle/model.py + vra_oracle.c (double-precision accumulation, OpenMP) with "
                       f"{layers_sample}/{cfg['num_layers']} layers + embedding/lm_head, prompt {prompt_len}; per-layer time scaled to "
                       f"{cfg['num_layers']} layers ({time.perf_counter() - t0:.0f} s of CPU work incl. weight generation)")


def cpu_baseline(cfg_int4, cfg_tiny):
    """`cpu_baseline` of the JSON line = the workload of the headline metric (Llama-3-8B int4 decode, bs 1; kind "port": the
    reference has no int4 CPU path, src/utils/gptq.rs:212-222, and cannot be built here); `cpu_baseline_config1` = BASELINE
    config 1, the shape the reference COULD run on its CPU backend (TinyLlama-1.1B bf16 greedy decode)."""
    b = oracle_decode_tokens_per_s(cfg_int4, 8, 16, 32, "Llama-3-8B-shape GPTQ int4 g128 bs=1 decode")   # ~10 s on the box's 128 host threads
    a = oracle_decode_tokens_per_s(cfg_tiny, 11, 16, 32, "TinyLlama-1.1B-shape bf16 bs=1 decode (BASELINE config 1)")  # ~3 s
    return b, a


class NodeRendezvous:
    """barrier + max-over-ranks for the ranks of ONE node over an abstract-namespace Unix socket: rank 0 listens on
    "\\0vra-bench-<key>", the others connect (retrying until it is there).  key = VRA_BENCH_RDZV (self-spawned ranks) or
    MASTER_PORT (torch.distributed.run).  world 1: no socket at all."""

    def __init__(self, rank, world, key=None, timeout=600.0):
        import socket
        import struct
        self.rank, self.world, self._struct = rank, world, struct
        self.peers, self.sock = [], None
        if world <= 1:
            return
        key = key or os.environ.get("VRA_BENCH_RDZV") or os.environ.get("MASTER_PORT") or "default"
        addr = "\0vra-bench-" + str(key)
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(addr)
            srv.listen(world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                r = struct.unpack("<i", self._recv(c, 4))[0]
                got[r] = c
            self.peers = [got[r] for r in sorted(got)]
            srv.close()
        else:
            t0 = time.time()
            while True:
                try:
                    self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                    self.sock.connect(addr)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    self.sock.close()
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rank 0 of the bench never opened the rendezvous socket")
                    time.sleep(0.05)
            self.sock.settimeout(timeout)
            self.sock.sendall(struct.pack("<i", rank))

    @staticmethod
    def _recv(c, n):
        b = b""
        while len(b) < n:
            k = c.recv(n - len(b))
            if not k:
                raise ConnectionError("a rank of the bench went away")
            b += k
        return b

    def max(self, x):
        """the maximum of x over all ranks, on every rank (doubles as the barrier)"""
        if self.world <= 1:
            return float(x)
        st = self._struct
        if self.rank == 0:
            m = max([float(x)] + [st.unpack("<d", self._recv(c, 8))[0] for c in self.peers])
            for c in self.peers:
                c.sendall(st.pack("<d", m))
            return m
        self.sock.sendall(st.pack("<d", float(x)))
        return st.unpack("<d", self._recv(self.sock, 8))[0]

    def barrier(self):
        self.max(0.0)

    def close(self):
        for c in self.peers + ([self.sock] if self.sock else []):
            c.close()
        self.peers, self.sock = [], None


def max_over_ranks(rdzv, seconds):
    """the job's time is the slowest rank's time."""
    return float(seconds) if rdzv is None else rdzv.max(seconds)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: one process per GPU, this script again with the launcher's environment
    variables; rank 0 prints the JSON line.  Returns the worst exit code."""
    import subprocess
    import uuid
    key = uuid.uuid4().hex[:12]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), VRA_BENCH_RDZV=key, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(abs(p.wait()) for p in procs)


def tp_rank_algorithmic_bytes(cfg, world, M):
    """SURVEY §8(d) per GEMM — K*N/2 (packed int4) + (K/g)*N*2 (scales) + (K/g)*N/2 (zeros) + M*K*2 + M*N*2, as
    Model::gemm_algorithmic_bytes counts them — over ONE rank's shard (column-parallel q/k/v/gate/up: N / world, kv heads
    replicated below one per rank; row-parallel o/down: K / world) + the unsharded bf16 lm_head (llama.rs:226-245)"""
    H, I, D, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"], cfg["num_layers"]
    hq, g = cfg["num_heads"] // world, cfg.get("group_size", 128)
    hkv = cfg["num_kv_heads"] // world if cfg["num_kv_heads"] >= world else 1

    def one(K, N):
        return K * N // 2 + (K // g) * N * 2 + (K // g) * N // 2 + M * K * 2 + M * N * 2

    layer = one(H, hq * D) + 2 * one(H, hkv * D) + one(hq * D, H) + 2 * one(H, I // world) + one(I // world, H)
    return L * layer + cfg["vocab_size"] * H * 2 + M * H * 2 + M * cfg["vocab_size"] * 4


def tp_main(a):
    """ONE tensor-parallel engine over a.tp ranks (runner processes; shared GPU => one-shot IPC transport, own GPUs => RCCL +
    one-shot): decode tokens/s of the TP engine on the Llama-3-70B shape scaled to the rank count, or of --model."""
    from vllm_rs_amd import engine as E
    from vllm_rs_amd.runner import TPEngine
    cfg = dict({"llama3-8b-gptq": E.LLAMA3_8B, "qwen2-7b-awq": E.QWEN2_7B, "llama3-70b": E.LLAMA3_70B}[a.model])
    t0 = time.perf_counter()
    with TPEngine(cfg, a.tp, tensors=None, num_gpu_blocks=a.blocks // a.tp, max_num_seqs=max(32, a.batch), max_model_len=8192,
                  use_graph=not a.no_graph, seed=1234) as tp:
        load_s = time.perf_counter() - t0
        res = tp.timed_decode(make_prompts(a.batch, a.prompt_len, cfg["vocab_size"]), a.warmup, a.steps)
        transport = tp.transport
    ms = max(r[0] for r in res)
    # A21: at temperature 0 every rank must have produced the same tokens (rank-ordered f32 sums in the one-shot all-reduce, RCCL's
    # own determinism above the one-shot size) — asserted, not assumed
    identical = all(r[1] == res[0][1] for r in res)
    if not identical:
        raise SystemExit(f"bench.py --tp {a.tp}: the ranks disagree on the generated tokens (A21): " + json.dumps([r[1][0][:8] for r in res]))
    from vllm_rs_amd import _lib
    ngpu = min(a.tp, max(1, _lib.load().vra_device_count()))  # fewer devices than ranks: the ranks SHARE device 0 (functional run, time-sliced)
    rank_bytes = tp_rank_algorithmic_bytes(cfg, a.tp, a.batch)
    own_gpus = ngpu == a.tp
    step_s = ms / a.steps / 1e3
    print(json.dumps({"metric": f"decode tokens/sec, {a.model} TP={a.tp}", "value": a.batch * a.steps / (ms / 1e3), "unit": "tokens/s",
                      "n_gpus": ngpu if own_gpus else 1, "ranks": a.tp,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms / a.steps, "higher_is_better": True, "scaling": "strong",
                      "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                      "tokens_identical_across_ranks": identical,
                      # per RANK: the bytes one rank's GEMMs + lm_head must read per decode step, over the step time (valid as a
                      # roofline fraction only when every rank has its own GPU; time-shared ranks are marked)
                      "roofline": {"bound": "hbm", "kernel": "one rank's decode step (int4 GEMV family of its shard + the unsharded lm_head)",
                                   "achieved": rank_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": rank_bytes / step_s / 1e9 / HBM_PEAK_GBS if own_gpus else None, "traffic": None,
                                   "algorithmic_bytes_per_rank_per_step": rank_bytes,
                                   "note": None if own_gpus else f"{a.tp} ranks time-share one GPU: achieved is per rank of the shared device, frac not meaningful"},
                      "config": {"workload": f"{a.model} shape, int4 g128, tensor parallel over {a.tp} ranks ({transport} transport: one-shot all-reduce up to 1 MiB "
                                             f"per message, RCCL above — chosen by message size), batch {a.batch}, "
                                             f"prompt {a.prompt_len}, {a.steps} generated tokens" + ("" if own_gpus else "; all ranks time-share ONE GPU"), "load_s": load_s}}))


def main():
    a = parse()
    if a.tp > 1:
        return tp_main(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s): the line would not be the one asked for")
    ncpu = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(ncpu // max(world, 1), 128))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    from vllm_rs_amd import _lib
    from vllm_rs_amd import engine as E
    L = _lib.load()
    # the ranks meet BEFORE anything touches a device: a node that is short of GPUs is reported by every rank after the whole
    # job has assembled (and the CPU test of the 8-rank launch exercises spawn + rendezvous without a GPU)
    dist = NodeRendezvous(rank, world) if world > 1 else None
    if dist is not None:
        dist.barrier()
        if rank == 0:
            print(f"bench.py: rendezvous of {world} ranks complete", file=sys.stderr, flush=True)
    if L.vra_device_count() <= local_rank:
        raise SystemExit(f"bench.py needs {world} GPU(s): no HIP device for rank {local_rank} (the product has no CPU fallback)")
    L.vra_set_device(local_rank)

    def sync():  # barrier + device synchronisation on both sides of the timed region
        L.vra_device_sync()
        if dist is not None:
            dist.barrier()
            L.vra_device_sync()

    models = {"llama3-8b-gptq": E.LLAMA3_8B, "qwen2-7b-awq": E.QWEN2_7B, "llama3-70b-tp8-rank": E.LLAMA3_70B_TP8_RANK}
    cfg = dict(models[a.model])
    max_bs = max(32, a.batch)
    # cpu_mem_fold = 0 in every engine of this file: the reference's default 0.2 (the package default too) would pin 13 GB of host
    # memory per 8192-block engine before the first step; no request of this benchmark is ever preempted
    eng = E.Engine(cfg, max_num_seqs=max_bs, max_model_len=8192, num_gpu_blocks=a.blocks, use_graph=not a.no_graph, device=local_rank,
                   seed=1234 + rank, cpu_mem_fold=0.0).init_synthetic()
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
    if not a.no_graph:
        # what the GPU alone needs for this step: the same decode graph replayed back to back, no upload / download / host work
        # in between (vra_engine_bench_replay); ms_per_step minus this is what the host loop adds per step
        rp = eng.bench_replay(64)
        line["step_overhead"] = {"gpu_ms_per_step_graph_replay_only": rp, "host_loop_ms_per_step": dt * 1e3 / a.steps - rp,
                                 "note": "replay-only = the step's hipGraph launched 64x back to back with frozen metadata; the timed region above runs the full loop "
                                         "(schedule, one upload, graph launch, downloads of tokens and error word, wait, commit) per step"}

    if rank == 0 and not a.no_extras:
        # ---------------- roofline of the dequant-GEMM family at the headline batch: every launch of the four GEMV shapes of a
        # layer timed with HIP events on the engine stream (320 launches each, rotating over all layers' weights).  `roofline`
        # is quoted on the TIME-DOMINANT launch (the largest share of the family's time per token); the family figure is the
        # quantity north_star grades (3 625 975 808 algorithmic bytes per token / the family's time per token).
        per = {}
        tot_b = tot_ms = 0.0
        for w, name in FAMILY.items():
            ms = eng.bench_gemm(w, a.batch, 320)
            b = eng.gemm_bytes(w, a.batch)
            per[name] = {"ms": ms, "bytes": b, "GBps": b / ms / 1e6, "kernel": KERNEL_OF[name]}
            tot_b += b
            tot_ms += ms
        dom_name = max(per, key=lambda k: per[k]["ms"])
        dom = per[dom_name]
        # `achieved` / `frac` are the FAMILY's (the four dequant-GEMV launches of a layer = what north_star grades): algorithmic
        # bytes of the four launches / the sum of their average durations.  The time-dominant single launch is listed beside it.
        line["roofline"] = {"bound": "hbm", "kernel": "dequant-GEMV family: gemv_q4s_kernel<BF16,1,false> x3 (norm+qkv, o_proj, down) + "
                                                       "gemv_q4s_kernel<BF16,2,false> (norm+gate/up+SiLU*mul) per layer",
                            "achieved": tot_b / tot_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS,
                            "traffic": pmc_traffic("family") if a.batch == 1 else None,
                            "algorithmic_bytes_per_launch": tot_b, "avg_launch_ms": tot_ms,
                            "note": "per 'launch' = the four launches of one layer (113 475 584 B); traffic = their FETCH_SIZE x 2 from the committed counter pass",
                            "time_dominant_launch": {"name": dom_name, "kernel": dom["kernel"], "GBps": dom["GBps"], "frac": dom["GBps"] / HBM_PEAK_GBS,
                                                     "avg_launch_ms": dom["ms"], "algorithmic_bytes_per_launch": dom["bytes"],
                                                     "traffic": pmc_traffic(dom["kernel"]) if a.batch == 1 else None},
                            "in_situ_family": in_situ_family() if a.batch == 1 else None,
                            "family": per, "family_GBps": tot_b / tot_ms / 1e6, "family_frac": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS,
                            "family_ms_per_token": tot_ms * cfg["num_layers"], "lib_sha16": lib_sha16(), "src_sha16": src_sha16()}
        # ---------------- bs=32 decode
        if a.batch != 32:
            dt32, _, _ = run_decode(eng, make_prompts(32, a.prompt_len, V, seed=43), 8, 64, lambda: L.vra_device_sync())
            line["bs32_tokens_per_s_per_gpu"] = 32 * 64 / dt32
            line["bs32_ms_per_step"] = dt32 * 1e3 / 64
            per32, b32, ms32 = {}, 0.0, 0.0
            for w, name in FAMILY.items():
                ms = eng.bench_gemm(w, 32, 160)
                b = eng.gemm_bytes(w, 32)
                per32[name] = {"ms": ms, "bytes": b, "GBps": b / ms / 1e6}
                b32 += b
                ms32 += ms
            line["roofline_bs32"] = {"bound": "hbm", "kernel": "dequant-GEMM family at 32 rows (gemv_q4w_kernel: norm+qkv, o_proj, norm+gate/up in its sequential pair form, down_proj in K slices; x in fragment order; launched as the step launches them: o_proj / down_proj also write the next norm's ready-made operands, the two fused-norm launches consume them)",
                                     "achieved": b32 / ms32 / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b32 / ms32 / 1e6 / HBM_PEAK_GBS,
                                     "family": per32, "family_ms_per_step": ms32 * cfg["num_layers"], "step_ms": dt32 * 1e3 / 64}
        # ---------------- the KV term ("also ctx in {1k, 8k}"): decode at long contexts
        lc = {}
        for bs, ctx in ((1, 1024), (1, 8000), (32, 1024), (32, 4096)):
            dtl, _, _ = run_decode(eng, make_prompts(bs, ctx, V, seed=77 + ctx), 4, 16, lambda: L.vra_device_sync())
            kv = bs * (ctx + 10) * 131072  # bytes of K and V read per step (131 072 B per token and sequence)
            lc[f"bs{bs}_ctx{ctx}"] = {"tokens_per_s": bs * 16 / dtl, "ms_per_step": dtl * 1e3 / 16, "kv_GB_per_step": kv / 1e9}
        line["long_context_decode"] = lc
        line["ttft_p50_ms"] = {"bs1_prompt128": ttft_p50(eng, 128, V, 1), "bs32_prompt128": ttft_p50(eng, 128, V, 32, reps=2),
                               "bs1_prompt2048": ttft_p50(eng, 2048, V, 1, reps=3),
                               # (round 5: the cost-model fix of vra_gemm_q4_big_fits — a 200-token prompt must not be slower than a 256-token one)
                               "bs1_prompt200": ttft_p50(eng, 200, V, 1, reps=3), "bs1_prompt256": ttft_p50(eng, 256, V, 1, reps=3),
                               "bs1_prompt1024": ttft_p50(eng, 1024, V, 1, reps=3), "bs1_prompt4096": ttft_p50(eng, 4096, V, 1, reps=3)}
        try:  # the MFMA roofline of the prefill kernels (half of the headline metric is TTFT)
            line["roofline_prefill"] = roofline_prefill(eng, L, cfg)
        except Exception as ex:
            line["roofline_prefill"] = {"error": repr(ex)}
        line["step_bytes_roofline"] = {"algorithmic_bytes_per_step": 3625975808 + 1050673152 + 532480,
                                       "frac_of_8TBps": (3625975808 + 1050673152 + 532480) / (dt / a.steps) / 8e12 if a.batch == 1 else None}
        eng.close()
        eng = None
        if a.model == "llama3-8b-gptq":
            # ---------------- opt-in engine mode: the dequantised weights of the dense prefill path kept RESIDENT (VRA_DENSE_PREFILL_RESIDENT=1:
            # 14 GB next to the 4.5 GB of int4 tensors, made once at engine creation; bit-identical logits, tests/test_gpu_engine.py) — what a
            # long prefill costs without its dequant passes.  The default line above keeps int4 as the only resident format.
            try:
                os.environ["VRA_DENSE_PREFILL_RESIDENT"] = "1"
                er = E.Engine(cfg, max_num_seqs=8, max_model_len=8192, num_gpu_blocks=a.blocks, use_graph=not a.no_graph, device=local_rank,
                              seed=1234, cpu_mem_fold=0.0).init_synthetic()
                ttft_p50(er, 128, V, 1, reps=2)
                line["ttft_p50_ms_resident_dense_weights"] = {"bs1_prompt2048": ttft_p50(er, 2048, V, 1, reps=3),
                                                              "bs1_prompt4096": ttft_p50(er, 4096, V, 1, reps=3), "extra_resident_GB": 13.96}
                er.close()
            except Exception as ex:
                line["ttft_p50_ms_resident_dense_weights"] = {"error": repr(ex)}
            finally:
                os.environ.pop("VRA_DENSE_PREFILL_RESIDENT", None)
            # ---------------- config 5: 32 768-token prompt (4 chunks of 8192), then the same prompt again (511-block prefix hit)
            e5 = E.Engine(E.LLAMA31_8B, max_num_seqs=8, max_model_len=40960, num_gpu_blocks=2048, enable_prefix_cache=True,
                          use_graph=not a.no_graph, device=local_rank, seed=1234, cpu_mem_fold=0.0).init_synthetic()
            p32 = make_prompts(1, 32768, V, seed=5)[0]
            res = []
            for _ in range(2):
                rid = e5.add_request(p32, max_tokens=2, ignore_eos=True)
                while e5.has_unfinished():
                    e5.step()
                t = e5.times(rid)
                res.append(t["first_token_ms"] - t["created_ms"])
            line["ttft_p50_ms"]["bs1_prompt32768"] = res[0]
            line["ttft_p50_ms"]["bs1_prompt32768_prefix_hit_511_blocks"] = res[1]
            # ... then 8 prompts that share its first 16 384 tokens (256 cached blocks) and differ in their last 1024
            import numpy as np
            tails = make_prompts(8, 1024, V, seed=6)
            rids = [e5.add_request(np.concatenate([p32[:16384], t]), max_tokens=2, ignore_eos=True) for t in tails]
            while e5.has_unfinished():
                e5.step()
            tt = sorted(e5.times(r)["first_token_ms"] - e5.times(r)["created_ms"] for r in rids)
            line["ttft_p50_ms"]["bs8_prompt17408_shared_16k_prefix"] = 0.5 * (tt[3] + tt[4])
            e5.close()
            # ---------------- FP8 (E4M3) KV cache: the KV term of long-context decode at half the bytes
            e8 = E.Engine(cfg, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=a.blocks, use_graph=not a.no_graph, device=local_rank,
                          seed=1234, fp8_kvcache=True, cpu_mem_fold=0.0).init_synthetic()
            lc8 = {}
            for bs, ctx in ((1, 8000), (32, 4096)):
                dtl, _, _ = run_decode(e8, make_prompts(bs, ctx, V, seed=77 + ctx), 4, 16, lambda: L.vra_device_sync())
                lc8[f"bs{bs}_ctx{ctx}"] = {"tokens_per_s": bs * 16 / dtl, "ms_per_step": dtl * 1e3 / 16, "kv_GB_per_step": bs * (ctx + 10) * 65536 / 1e9}
            line["long_context_decode_fp8_kv"] = lc8
            e8.close()
            # ---------------- the reference's binding path
            line["ffi_path"] = ffi_path(L, cfg)
            line["ffi_path"]["native_family_ms_per_token"] = line["roofline"]["family_ms_per_token"]
            try:
                line["ffi_step"] = ffi_step(L, cfg)
                line["ffi_step_ms"] = line["ffi_step"].get("graph_ms_per_step")
            except Exception as ex:  # a leg of the extras never takes the line down
                line["ffi_step"] = {"error": repr(ex)}
            line["runner_ipc_step"] = runner_ipc_step(cfg)
            line["runner_ipc_step_ms"] = line["runner_ipc_step"].get("ms_per_step")
            # ---------------- config 3: Qwen2-7B AWQ
            eq = E.Engine(E.QWEN2_7B, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=2048, use_graph=not a.no_graph, device=local_rank,
                          seed=99, cpu_mem_fold=0.0).init_synthetic()
            Vq = E.QWEN2_7B["vocab_size"]
            d1, _, _ = run_decode(eq, make_prompts(1, a.prompt_len, Vq, seed=11), 8, 64, lambda: L.vra_device_sync())
            d32, _, _ = run_decode(eq, make_prompts(32, a.prompt_len, Vq, seed=12), 8, 32, lambda: L.vra_device_sync())
            tq = {"bs1_prompt128": ttft_p50(eq, 128, Vq, 1, reps=3), "bs1_prompt2048": ttft_p50(eq, 2048, Vq, 1, reps=3)}  # (2048: the AWQ dequant pass + dense GEMM)
            line["qwen2_7b_awq"] = {"ttft_p50_ms": tq, "bs1_tokens_per_s": 64 / d1, "bs1_ms_per_step": d1 * 1e3 / 64, "bs32_tokens_per_s": 32 * 32 / d32,
                                    "bs32_ms_per_step": d32 * 1e3 / 32, "algorithmic_bytes_per_step": 3390091264 + 1089994752,
                                    "frac_of_8TBps_bs1": (3390091264 + 1089994752) / (d1 / 64) / 8e12}
            eq.close()
        if world == 1 and not a.no_cpu:
            line["cpu_baseline"], line["cpu_baseline_config1"] = cpu_baseline(E.LLAMA3_8B, E.TINYLLAMA)
            if a.model == "llama3-8b-gptq" and not a.no_parity:
                # ---------------- full-depth parity of the configuration that was just timed (graph path, L = 32, V = 128256): the
                # oracle as the checker (tests/full_depth.py), ~20 s of host time
                from tests import full_depth
                rep = full_depth.run(dict(cfg), log=lambda *_: None)
                keys = ("workload", "tokens_equal", "first_divergence", "n_steps", "max_abs", "mean_abs", "min_frac_within_1e3_abs",
                        "min_frac_within_1e3_of_scale", "max_ulp_of_row_scale", "logit_scale", "per_step_max_abs", "seconds")
                line["parity_full_depth"] = {k: rep[k] for k in keys}
                # ... and against the REFERENCE's arithmetic: its norm order everywhere (others.rs:11-29) with the exact int4 product, and
                # the same with Marlin's 16-bit weight rounding (gptq.rs:116-178) — not only against the order the engine chose
                vkeys = ("arithmetic", "tokens_equal", "first_divergence", "n_steps", "max_abs", "mean_abs", "min_frac_within_1e3_of_scale",
                         "max_ulp_of_row_scale", "mean_ulp_of_row_scale", "per_step_max_abs", "seconds")
                line["parity_full_depth_reference_order"] = {k: rep["reference_order"][k] for k in vkeys}
                line["parity_full_depth_marlin_rounded"] = {k: rep["marlin_rounded"][k] for k in vkeys}
                if not a.no_extras:
                    # config 3 (Qwen2-7B AWQ, L = 28, V = 152064) the same way; zero points drawn as in real AWQ checkpoints
                    rep = full_depth.run(dict(E.QWEN2_7B), log=lambda *_: None)
                    line["parity_full_depth_qwen2"] = {k: rep[k] for k in keys}
                    line["parity_full_depth_qwen2_reference_order"] = {k: rep["reference_order"][k] for k in vkeys}
                    line["parity_full_depth_qwen2_marlin_rounded"] = {k: rep["marlin_rounded"][k] for k in vkeys}
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.close()
    if rank == 0:
        assert line["n_gpus"] == a.gpus
        print(json.dumps(line))


if __name__ == "__main__":
    main()
