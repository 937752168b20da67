"""GPU parity of the fused decode launch "RMSNorm + q/k/v GEMV + RoPE + KV-cache write + paged attention" (csrc/qkv_attn.hip): decode
steps of 1..4 sequences run it when `vra_debug_set_fused_qkv_attn(1)` / VRA_FUSED_QKV_ATTN=1 is set (opt-in: measured at parity
with the two-launch path — kernel E, then decode_attn_fused_kernel + merge —).

  * against the oracle: the tolerance of tests/test_gpu_engine.py;
  * against the two-launch path of the SAME engine: the q/k/v values are bit-identical (same kernel-E arithmetic), the attention
    deals its 32-token tiles to 16 waves instead of 4 (another merge order): single ulps; and the KV cache both paths leave
    behind must be IDENTICAL (the next steps read it);
  * contexts that span one tile per wave, several tiles per wave (> 512 tokens), ragged tails, 1..4 sequences, shuffled block
    tables, both head dims, AWQ + bias, f16, graph replay == eager bitwise."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.test_gpu_engine import BF16, F16, build, check_logits, prefill_inputs, simple_tables, small_cfg
from vllm_rs_amd import _lib

pytestmark = pytest.mark.gpu

CFGS = {
    "llama3_8b_widths": small_cfg(hidden_size=4096, intermediate_size=14336, num_layers=2, num_heads=32, num_kv_heads=8, head_dim=128,
                                  vocab_size=2048, rope_theta=500000.0, max_position_embeddings=4096),
    "small_gptq": small_cfg(max_position_embeddings=2048),
    "small_f16": small_cfg(dtype=F16, max_position_embeddings=2048),
    "small_awq_bias": small_cfg(arch="qwen2", quant_method="awq", attention_bias=True, num_heads=8, num_kv_heads=2, head_dim=64, hidden_size=512,
                                max_position_embeddings=2048),
    "tinyllama_widths_q": small_cfg(hidden_size=2048, intermediate_size=5632, num_layers=2, num_heads=32, num_kv_heads=4, head_dim=64, vocab_size=2048,
                                    max_position_embeddings=2048),
    "qwen2_7b_widths": small_cfg(arch="qwen2", attention_bias=True, hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28,
                                 num_kv_heads=4, head_dim=128, vocab_size=2048, quant_method="awq", rope_theta=1e6, rms_norm_eps=1e-6,
                                 max_position_embeddings=2048),
}


def _decode_inputs(seqs, bt):
    ids = np.array([s[-1] for s in seqs], np.uint32)
    pos = np.array([len(s) - 1 for s in seqs], np.int64)
    slots = np.array([int(bt[b, (len(s) - 1) // 64]) * 64 + (len(s) - 1) % 64 for b, s in enumerate(seqs)], np.int64)
    ctx = np.array([len(s) for s in seqs], np.uint32)
    return ids, pos, slots, ctx


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("lens", [(37,), (70, 5), (129, 64, 31, 200), (700,), (1030, 3)])
def test_fused_qkv_attention_matches_oracle_and_the_two_launch_path(name, lens):
    cfg = CFGS[name]
    lib = _lib.load()
    nblk = sum((n + 8 + 63) // 64 for n in lens) + 2
    eng, oracle = build(cfg, seed=11, max_num_seqs=8, num_gpu_blocks=nblk)
    try:
        r = np.random.default_rng(2)
        prompts = [r.integers(0, cfg["vocab_size"], size=n).tolist() for n in lens]
        bt = simple_tables([len(p) + 8 for p in prompts])
        # shuffled physical blocks: the paged walk must follow the table
        perm = r.permutation(nblk).astype(np.uint32)
        bt = perm[bt]
        ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
        eng.forward_raw(ids, pos, slots, bt, ctx, cu)
        ref = oracle.forward(ids, pos, slots, bt, ctx, cu)
        seqs = [list(p) for p in prompts]
        tok = orc.argmax_f32(ref)
        for step in range(3):
            for s, t in zip(seqs, tok):
                s.append(int(t))
            ids, pos, slots, ctx = _decode_inputs(seqs, bt)
            lib.vra_debug_set_fused_qkv_attn(0)
            two = eng.forward_raw(ids, pos, slots, bt, ctx, None)
            lib.vra_debug_set_fused_qkv_attn(1)
            one = eng.forward_raw(ids, pos, slots, bt, ctx, None)
            again = eng.forward_raw(ids, pos, slots, bt, ctx, None)
            ref = oracle.forward(ids, pos, slots, bt, ctx, None)
            assert np.array_equal(one.view(np.uint32), again.view(np.uint32)), f"{name} {lens} step {step}: not reproducible"
            check_logits(one, ref, f"{name} {lens} fused qkv+attention step {step}", cfg["dtype"], max_ulps=5.0)
            check_logits(one, two, f"{name} {lens} fused vs two launches step {step}", cfg["dtype"], max_ulps=2.0)
            tok = orc.argmax_f32(ref)
    finally:
        lib.vra_debug_set_fused_qkv_attn(0)
        eng.close()


def test_fused_launch_leaves_the_same_kv_cache_as_the_two_launch_path():
    """two engines over the same weights, one per path, same steps: every later step reads the cache the earlier ones wrote, so the
    generated tokens agree over 40 steps only if the fused launch writes the new K row / V column where and as the other one does"""
    cfg = CFGS["llama3_8b_widths"]
    lib = _lib.load()
    outs = []
    try:
        for fused in (0, 1):
            lib.vra_debug_set_fused_qkv_attn(fused)
            eng, _ = build(cfg, seed=5, max_num_seqs=4, use_graph=False)
            outs.append([o.tolist() for o in eng.generate([list(range(10, 70)), list(range(300, 331)), list(range(40, 45))], max_tokens=40, ignore_eos=True)])
            eng.close()
    finally:
        lib.vra_debug_set_fused_qkv_attn(0)
    same = sum(a == b for a, b in zip(outs[0], outs[1]))
    # (greedy generation from random weights: a single-ulp difference in one logit can flip a near-tie and then the sequences
    # diverge; at least two of the three must agree token for token over all 40 steps, all three on the first 8)
    assert same >= 2, outs
    assert all(a[:8] == b[:8] for a, b in zip(outs[0], outs[1])), outs


def test_fused_launch_graph_replay_equals_eager_bitwise_and_replays_fresh_tags():
    cfg = CFGS["llama3_8b_widths"]
    lib = _lib.load()
    outs = []
    lib.vra_debug_set_fused_qkv_attn(1)
    try:
        for use_graph in (False, True):
            eng, _ = build(cfg, seed=5, max_num_seqs=4, use_graph=use_graph)
            outs.append(eng.generate([list(range(10, 60)), list(range(100, 131))], max_tokens=48, ignore_eos=True))
            eng.close()
    finally:
        lib.vra_debug_set_fused_qkv_attn(0)
    for a, b in zip(outs[0], outs[1]):
        assert a.tolist() == b.tolist(), (outs[0], outs[1])
