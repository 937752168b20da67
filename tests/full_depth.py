"""Full-depth parity of a BASELINE config (north_star: "outputs match ... token-for-token at temperature=0
(logits within 1e-3 for bf16)"): the ENGINE as it is timed — synthetic weights of the named shape, every layer, the full
vocabulary, hipGraph replay — generates greedily from a prompt; the CPU oracle (oracle/model.py over the SAME synthetic
checkpoint, rebuilt from the counter hash) is fed the engine's tokens and its logits are compared step by step.

Test infrastructure: imported by tests/ and by bench.py's parity leg only (the product never imports the oracle).

Reported per run (not only asserted): tokens equal / first divergence index and the oracle's top-2 gap there, per-step max and
mean |dlogit|, the fraction of logits within 1e-3 absolute and within 1e-3 of the row's logit scale, and the deviation in
storage ulps of the row scale (the unit the per-layer tests use)."""
import time

import numpy as np

from oracle import model as om
from oracle import oracle as orc


def synthetic_checkpoint(cfg, seed=1234):
    """the checkpoint Model::init_synthetic draws on the device (host/model.cpp:100-145, qlinear_synth :52-83), in checkpoint
    format: tiled hash words -> nibble indices -> GPTQ / AWQ packing"""
    H, I, V, L, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_layers"], cfg["head_dim"]
    Hq, Hkv, g, dt = cfg["num_heads"], cfg["num_kv_heads"], cfg.get("group_size", 128), cfg["dtype"]
    awq = cfg.get("quant_method") == "awq"
    bias = bool(cfg.get("attention_bias")) or cfg.get("arch") == "qwen2"
    w = {"model.embed_tokens.weight": orc.fill_normal((V, H), seed + 7, 0.0, 0.02, dt), "model.norm.weight": orc.fill_normal((H,), seed + 8, 1.0, 0.02, dt)}
    if not cfg.get("tie_word_embeddings"):
        w["lm_head.weight"] = orc.fill_normal((V, H), seed + 9, 0.0, 0.02, dt)

    def lin(prefix, K, N, s, with_bias):
        tiled = orc.fill_hash_u32((K // 8) * N, s).reshape(K // 16, N * 2)
        idx = orc.tile_to_indices(tiled, K, N)
        w[prefix + ".qweight"] = orc.awq_pack(idx) if awq else orc.gptq_pack(idx)
        w[prefix + ".scales"] = orc.fill_uniform((K // g, N), s + 1, 0.002, 0.02, dt)
        if awq:
            w[prefix + ".qzeros"] = orc.fill_awq_zeros((K // g) * (N // 8), s + 2).reshape(K // g, N // 8)
        if with_bias:
            w[prefix + ".bias"] = orc.fill_normal((N,), s + 3, 0.0, 0.02, dt)

    for l in range(L):
        p = f"model.layers.{l}."
        w[p + "input_layernorm.weight"] = orc.fill_normal((H,), seed + 100 + 2 * l, 1.0, 0.02, dt)
        w[p + "post_attention_layernorm.weight"] = orc.fill_normal((H,), seed + 101 + 2 * l, 1.0, 0.02, dt)
        s = seed + 1234 + l * 64
        lin(p + "self_attn.q_proj", H, Hq * D, s + 0, bias)
        lin(p + "self_attn.k_proj", H, Hkv * D, s + 4, bias)
        lin(p + "self_attn.v_proj", H, Hkv * D, s + 8, bias)
        lin(p + "self_attn.o_proj", Hq * D, H, s + 12, False)
        lin(p + "mlp.gate_proj", H, I, s + 16, False)
        lin(p + "mlp.up_proj", H, I, s + 20, False)
        lin(p + "mlp.down_proj", I, H, s + 24, False)
    return w


def _stats(got, ref, dt):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = float(np.abs(ref).max())
    bits = 8 if dt == 0 else 11
    ulp = 2.0 ** (np.floor(np.log2(max(scale, 1.0))) - (bits - 1))
    srt = np.sort(ref)
    return dict(max_abs=float(d.max()), mean_abs=float(d.mean()), frac_within_1e3_abs=float((d <= 1e-3).mean()),
                frac_within_1e3_of_scale=float((d <= 1e-3 * scale).mean()), max_ulp_of_row_scale=float(d.max() / ulp), mean_ulp_of_row_scale=float(d.mean() / ulp), logit_scale=scale,
                oracle_top2_gap=float(srt[-1] - srt[-2]))


def run(cfg, *, n_prompt=32, n_gen=16, seed=1234, use_graph=True, blocks=64, engine_kw=None, log=print, variants=True):
    """-> report dict.  The engine generates; the oracle is teacher-forced with the engine's tokens (a divergence does not cascade)."""
    from vllm_rs_amd.engine import Engine
    t0 = time.perf_counter()
    eng = Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=blocks, use_graph=use_graph, seed=seed, **(engine_kw or {})).init_synthetic()
    r = np.random.default_rng(42)
    prompt = r.integers(1000, cfg["vocab_size"] - 1000, size=n_prompt).astype(np.uint32)
    rid = eng.add_request(prompt, max_tokens=n_gen, ignore_eos=True)
    eng_logits = []
    while eng.has_unfinished():
        eng.step()
        eng_logits.append(eng.last_logits(1)[0].copy())
    toks = [int(t) for t in eng.output(rid)]
    # (the last sampled token is not appended on finish, SURVEY Appendix A2: the sampled token of step i is argmax(logits[i]))
    eng_tokens = [int(np.argmax(l)) for l in eng_logits]
    eng.close()
    t_eng = time.perf_counter() - t0
    w = synthetic_checkpoint(cfg, seed)
    t_w = time.perf_counter() - t0 - t_eng
    oracle = om.OracleModel(dict(cfg, max_position_embeddings=min(cfg["max_position_embeddings"], 2048)), w, num_blocks=blocks)
    del w
    bt = np.arange(blocks, dtype=np.uint32)[None]
    pos = np.arange(n_prompt, dtype=np.int64)

    def teacher_forced(rule, rounding):
        """one pass of the oracle over the engine's request under one arithmetic variant -> (per-step stats, first divergence)"""
        om.ENGINE_RULE, om.WEIGHT_ROUNDING = rule, rounding
        try:
            oracle.reset_cache()
            ref = oracle.forward(prompt, pos, pos.copy(), bt, [n_prompt], [0, n_prompt])
            steps, first_div = [], None
            n = n_prompt
            for i, el in enumerate(eng_logits):
                st = _stats(el, ref[0], cfg["dtype"])
                ot = int(orc.argmax_f32(ref)[0])
                st["engine_token"], st["oracle_token"] = eng_tokens[i], ot
                if ot != eng_tokens[i] and first_div is None:
                    first_div = dict(step=i, oracle_top2_gap=st["oracle_top2_gap"], engine_token=eng_tokens[i], oracle_token=ot)
                steps.append(st)
                if i + 1 < len(eng_logits):
                    tok = np.array([eng_tokens[i]], np.uint32)  # teacher forcing with the ENGINE's token
                    ref = oracle.forward(tok, np.array([n], np.int64), np.array([n], np.int64), bt, [n + 1])
                    n += 1
            return steps, first_div
        finally:
            om.WEIGHT_ROUNDING = "exact"

    def summary(steps, first_div):
        return dict(tokens_equal=first_div is None, first_divergence=first_div, n_steps=len(steps),
                    max_abs=max(s["max_abs"] for s in steps), mean_abs=float(np.mean([s["mean_abs"] for s in steps])),
                    min_frac_within_1e3_abs=min(s["frac_within_1e3_abs"] for s in steps),
                    min_frac_within_1e3_of_scale=min(s["frac_within_1e3_of_scale"] for s in steps),
                    max_ulp_of_row_scale=max(s["max_ulp_of_row_scale"] for s in steps),
                    mean_ulp_of_row_scale=float(np.mean([s["mean_ulp_of_row_scale"] for s in steps])),
                    logit_scale=max(s["logit_scale"] for s in steps))

    # (1) the mirror: the oracle restates the order of the engine's fused-norm launches (oracle/model.py) — kernel faults show here
    steps, first_div = teacher_forced(eng.L, "exact")
    rep = dict(workload=f"H{cfg['hidden_size']} L{cfg['num_layers']} V{cfg['vocab_size']} {cfg.get('quant_method')} "
                        f"{'graph' if use_graph else 'eager'}: prompt {n_prompt} + {len(eng_logits)} greedy steps, oracle teacher-forced with the engine's tokens",
               **summary(steps, first_div),
               per_step_max_abs=[round(s["max_abs"], 5) for s in steps], per_step_mean_abs=[round(s["mean_abs"], 6) for s in steps],
               engine_tokens=eng_tokens, generated=toks)
    # (2), (3) the REFERENCE's arithmetic: its norm order everywhere (round(round(x·rstd)·γ) ahead of the GEMM,
    # others.rs:11-29) with the exact int4 product, and the same with Marlin's 16-bit weight rounding (gptq.rs:116-178) — the
    # distance a maintainer linking this library into vllm.rs would see against the CUDA build
    if variants:
        for key, rounding in (("reference_order", "exact"), ("marlin_rounded", "marlin")):
            t1 = time.perf_counter()
            st2, fd2 = teacher_forced(None, rounding)
            rep[key] = dict(summary(st2, fd2), per_step_max_abs=[round(s["max_abs"], 5) for s in st2],
                            arithmetic=("reference norm order (others.rs:11-29), " +
                                        ("weights rounded to the model dtype before the product (Marlin, gptq.rs:116-178)" if rounding == "marlin"
                                         else "exact int4 product, one rounding at the output")),
                            seconds=round(time.perf_counter() - t1, 1))
            log(f"[full-depth parity vs {key}] tokens_equal={rep[key]['tokens_equal']} first_divergence={fd2} max|d|={rep[key]['max_abs']:.4f} "
                f"mean|d|={rep[key]['mean_abs']:.5f} max {rep[key]['max_ulp_of_row_scale']:.2f} / mean {rep[key]['mean_ulp_of_row_scale']:.3f} ulp of the row scale")
    om.ENGINE_RULE = eng.L
    rep["seconds"] = dict(engine=round(t_eng, 1), oracle_weights=round(t_w, 1), total=round(time.perf_counter() - t0, 1))
    log(f"[full-depth parity] {rep['workload']}: tokens_equal={rep['tokens_equal']} first_divergence={first_div} max|d|={rep['max_abs']:.4f} "
        f"mean|d|={rep['mean_abs']:.5f} within 1e-3 abs >= {rep['min_frac_within_1e3_abs']:.3f}, within 1e-3 of scale >= "
        f"{rep['min_frac_within_1e3_of_scale']:.3f}, max {rep['max_ulp_of_row_scale']:.2f} ulp of the row scale ({rep['seconds']})")
    return rep
