#!/usr/bin/env python3
"""Generates tests/golden/hf_{llama,qwen2,qwen3}_tiny.npz — run in the BUILD container only (needs `transformers`
and torch on CPU; nothing here reads /root/reference, and the GPU box never runs this script).

Independent pin of the oracle's MODEL STRUCTURE (pre-norm residual blocks, RoPE convention = rotate-half,
GQA head mapping, SwiGLU, qkv bias for Qwen2, the per-head q_norm / k_norm of Qwen3 (round 5; attention.rs:713-735), last-token
lm_head): a tiny random LlamaForCausalLM / Qwen2ForCausalLM / Qwen3ForCausalLM from HuggingFace transformers, weights rounded to values exactly representable in bf16
AND f16, evaluated in float32.  The fixture holds inputs and expected outputs only:
  cfg_json, prompt ids, every weight tensor (uint16 bf16 bit patterns, HF names), f32 logits of the prompt's
  last position and of 8 greedy decode steps, and the greedy tokens.
Usage: python tests/golden/make_hf_golden.py
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def to_bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16).copy()


def build(kind, seed):
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM, Qwen3Config, Qwen3ForCausalLM
    common = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                  vocab_size=128, max_position_embeddings=128, rms_norm_eps=1e-5, tie_word_embeddings=False,
                  attn_implementation="eager")
    torch.manual_seed(seed)
    if kind == "llama":
        cfg = LlamaConfig(rope_theta=10000.0, attention_bias=False, mlp_bias=False, **common)
        model = LlamaForCausalLM(cfg)
    elif kind == "qwen3":
        cfg = Qwen3Config(rope_theta=1000000.0, use_sliding_window=False, head_dim=64, attention_bias=False, **common)
        model = Qwen3ForCausalLM(cfg)
    else:
        cfg = Qwen2Config(rope_theta=1000000.0, use_sliding_window=False, **common)
        model = Qwen2ForCausalLM(cfg)
    model = model.float().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            scale = 4.0 if ("proj" in n or "lm_head" in n) else 1.0   # livelier logits than the default init gives
            if "bias" in n:
                p.copy_(torch.randn_like(p) * 0.1)
            if "q_norm" in n or "k_norm" in n:  # (HF initialises them to 1: a norm without weights would not pin the weight's use)
                p.copy_(1.0 + 0.25 * torch.randn_like(p))
            v = (p * scale).to(torch.bfloat16).float()
            v = torch.where(v.abs() < 2.0 ** -14, torch.zeros_like(v), v)  # exactly representable in f16 as well
            p.copy_(v)
    return cfg, model


def run(kind, seed):
    cfg, model = build(kind, seed)
    g = torch.Generator().manual_seed(seed + 1)
    prompt = torch.randint(1, cfg.vocab_size, (1, 12), generator=g)
    ids = prompt.clone()
    logits, toks = [], []
    with torch.no_grad():
        for _ in range(9):  # prompt + 8 decode steps, full recompute in f32 (no KV cache involved on the HF side)
            out = model(ids).logits[0, -1].float()
            logits.append(out.numpy().copy())
            nxt = int(out.argmax())
            toks.append(nxt)
            ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
    weights = {n: to_bf16_bits(p) for n, p in model.state_dict().items() if "rotary" not in n and "inv_freq" not in n}
    meta = dict(arch=kind if kind in ("qwen2", "qwen3") else "llama", hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads,
                head_dim=getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads, vocab_size=cfg.vocab_size,
                max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                rope_theta=float(cfg.rope_parameters["rope_theta"]), attention_bias=(kind == "qwen2"), quant_method=None,
                **({"qk_norm": "head"} if kind == "qwen3" else {}))
    path = os.path.join(HERE, f"hf_{kind}_tiny.npz")
    np.savez_compressed(path, cfg_json=np.frombuffer(json.dumps(meta).encode(), np.uint8), prompt=prompt[0].numpy().astype(np.uint32),
                        logits=np.stack(logits).astype(np.float32), tokens=np.array(toks, np.uint32),
                        **{"w:" + k: v for k, v in weights.items()})
    print(path, os.path.getsize(path), "bytes; tokens", toks)


if __name__ == "__main__":
    import sys
    kinds = sys.argv[1:] or ["llama", "qwen2", "qwen3"]
    for kind, seed in (("llama", 11), ("qwen2", 23), ("qwen3", 37)):
        if kind in kinds:
            run(kind, seed)
