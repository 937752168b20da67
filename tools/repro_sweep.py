#!/usr/bin/env python3
"""Bit-reproducibility sweep (engine only, no oracle): every forward pass of a shape x step-size grid is run REPS times on the
same inputs and the logits are compared bit for bit.  A data race in a kernel shows up here as run-to-run differences long
before it shows up against a tolerance (round 3: kernel C's multi-item mode).

    python tools/repro_sweep.py [config ...]      # default: all"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests.test_gpu_engine import F16, prefill_inputs, simple_tables, small_cfg  # noqa: E402
from vllm_rs_amd.engine import Engine  # noqa: E402

L8 = dict(hidden_size=4096, intermediate_size=14336, num_heads=32, num_kv_heads=8, head_dim=128, rope_theta=500000.0)
CFGS = {
    "llama3_8b": small_cfg(num_layers=2, vocab_size=2048, max_position_embeddings=8192, **L8),
    "llama3_8b_awq": small_cfg(num_layers=1, vocab_size=2048, max_position_embeddings=8192, quant_method="awq", **L8),
    "llama3_8b_f16": small_cfg(num_layers=1, vocab_size=2048, max_position_embeddings=8192, dtype=F16, **L8),
    "tinyllama_q": small_cfg(hidden_size=2048, intermediate_size=5632, num_layers=2, num_heads=32, num_kv_heads=4, head_dim=64, vocab_size=2048,
                             max_position_embeddings=8192),
    "tinyllama_dense": small_cfg(hidden_size=2048, intermediate_size=5632, num_layers=2, num_heads=32, num_kv_heads=4, head_dim=64, vocab_size=2048,
                                 max_position_embeddings=8192, quant_method=None),
    "qwen2_7b": small_cfg(arch="qwen2", attention_bias=True, hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28, num_kv_heads=4,
                          head_dim=128, vocab_size=2048, quant_method="awq", rope_theta=1e6, rms_norm_eps=1e-6, max_position_embeddings=8192),
    "llama3_70b": small_cfg(hidden_size=8192, intermediate_size=28672, num_layers=1, num_heads=64, num_kv_heads=8, head_dim=128, vocab_size=2048,
                            max_position_embeddings=8192, rope_theta=500000.0),
    "llama3_70b_tp8_rank": small_cfg(hidden_size=8192, intermediate_size=3584, num_layers=1, num_heads=8, num_kv_heads=1, head_dim=128, vocab_size=2048,
                                     max_position_embeddings=8192, rope_theta=500000.0),
    "small_g32": small_cfg(group_size=32),
    "small_g64_awq": small_cfg(group_size=64, quant_method="awq"),
    "small_channelwise": small_cfg(group_size=-1),
    "small_f16": small_cfg(dtype=F16),
}
# (768+ tokens: the dense prefill path — kernel X with its split-K / tail-split exchanges through memory, csrc/gemm_dense.cuh)
PREFILL = [(1,), (3,), (7,), (20,), (33,), (64,), (100,), (128,), (200, 57), (512,), (1000, 24), (2048,), (3072,), (4096,), (5,) * 8, (17,) * 32, (300, 1, 64, 129)]
DECODE = [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 24, 32]
REPS = 6


def main(names=None):
    names = names or sys.argv[1:] or list(CFGS)
    bad = 0
    n_fwd = 0
    for name in names:
        cfg = CFGS[name]
        mp = cfg["max_position_embeddings"]
        for fp8 in (False, True):
            t0 = time.time()
            eng = Engine(cfg, max_num_seqs=32, max_model_len=mp, num_gpu_blocks=256, use_graph=False, seed=7, fp8_kvcache=fp8).init_synthetic()
            r = np.random.default_rng(5)
            for lens in PREFILL:
                if sum(lens) + 8 > mp or (fp8 and sum(lens) > 600):
                    continue
                prompts = [r.integers(0, cfg["vocab_size"], size=n).tolist() for n in lens]
                bt = simple_tables([len(p) + 8 for p in prompts])
                ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
                outs = [eng.forward_raw(ids, pos, slots, bt, ctx, cu) for _ in range(REPS)]
                n_fwd += REPS
                nd = sum(not np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs[1:])
                if nd or not np.isfinite(outs[0]).all():
                    bad += 1
                    print(f"NONDET {name} fp8={fp8} prefill {lens}: {nd}/{REPS - 1} runs differ, max |d| {max(float(np.abs(outs[0] - o).max()) for o in outs[1:]):.4f}", flush=True)
            for B in DECODE:
                for clen in ((30, 70), (1500, 2500)):  # short contexts; long ones (split-KV attention + merge)
                    if clen[1] + 8 > mp or B * ((clen[1] + 72) // 64) > 256 or (fp8 and B not in (1, 5, 32)):
                        continue
                    lens = r.integers(clen[0], clen[1], size=B)
                    prompts = [r.integers(0, cfg["vocab_size"], size=int(n)).tolist() for n in lens]
                    bt = simple_tables([len(p) + 8 for p in prompts])
                    for b0 in range(0, B, 4):  # prefill in groups (bounded step size)
                        pi = prefill_inputs(prompts[b0:b0 + 4], bt[b0:b0 + 4])
                        eng.forward_raw(pi[0], pi[1], pi[2], bt[b0:b0 + 4], pi[3], pi[4])
                    ids = r.integers(0, cfg["vocab_size"], size=B).astype(np.uint32)
                    pos = np.array([len(p) for p in prompts], np.int64)
                    slots = np.array([int(bt[b, pos[b] // 64]) * 64 + pos[b] % 64 for b in range(B)], np.int64)
                    ctx = (pos + 1).astype(np.uint32)
                    outs = [eng.forward_raw(ids, pos, slots, bt, ctx, None) for _ in range(REPS)]
                    n_fwd += REPS
                    nd = sum(not np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs[1:])
                    if nd or not np.isfinite(outs[0]).all():
                        bad += 1
                        print(f"NONDET {name} fp8={fp8} decode B={B} ctx~{clen}: {nd}/{REPS - 1} runs differ, max |d| {max(float(np.abs(outs[0] - o).max()) for o in outs[1:]):.4f}", flush=True)
            eng.close()
            print(f"{name} fp8={fp8}: swept in {time.time() - t0:.1f} s ({n_fwd} forwards so far)", flush=True)
    print("non-reproducible cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
