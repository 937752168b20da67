"""BASELINE config 5 on the GPU: 32k-token prompts through chunked prefill (4 x 8192, scheduler.rs:203,718-785), the paged
KV cache (512 blocks per sequence) and the prefix cache (511-block hit on resubmission, block_manager.rs:291-299; eight
prompts sharing a 16k prefix) — SURVEY §8(d) "Config 5".

 * parity: a 1-layer model at the full 32 768-token length, every chunk's last-row logits against the CPU oracle (the
   oracle's attention runs the 32k context in double precision in seconds, oracle/vra_oracle.c);
 * properties at full size: the Llama-3.1-8B shape (32 layers, synthetic int4 weights) — step counts, TTFT ordering,
   identical tokens from cached and recomputed prefixes, block accounting."""
import time

import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import check_logits, small_cfg
from vllm_rs_amd import engine as E
from vllm_rs_amd.engine import Engine

pytestmark = pytest.mark.gpu
BF16 = 0
CTX, CHUNK, BS = 32768, 8192, 64


def test_32k_chunked_prefill_logits_match_oracle_one_layer():
    cfg = small_cfg(num_layers=1, quant_method="gptq", max_position_embeddings=40960, rope_theta=500000.0,
                    rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                      original_max_position_embeddings=8192))
    w = om.make_random_checkpoint(cfg, 21)
    nblk = CTX // BS + 2
    eng = Engine(cfg, num_gpu_blocks=nblk, max_num_seqs=4, max_model_len=40960, use_graph=False).load_weights(w)
    oracle = om.OracleModel(cfg, w, num_blocks=nblk)
    r = np.random.default_rng(21)
    prompt = r.integers(1, cfg["vocab_size"] - 1, size=CTX).astype(np.uint32)
    bt = r.permutation(nblk)[: CTX // BS + 1].astype(np.uint32)[None]       # shuffled physical blocks
    for c in range(CTX // CHUNK):
        pos = np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.int64)
        slots = bt[0, pos // BS].astype(np.int64) * BS + pos % BS
        ctx = np.array([(c + 1) * CHUNK], np.uint32)
        cu = np.array([0, CHUNK], np.uint32)
        got = eng.forward_raw(prompt[pos], pos, slots, bt, ctx, cu)
        ref = oracle.forward(prompt[pos], pos, slots, bt, ctx, cu)
        check_logits(got, ref, f"32k prefill chunk {c} (context {ctx[0]})")
    # one decode step on top of the 32k context
    tok = np.array([int(orc.argmax_f32(ref)[0])], np.uint32)
    pos = np.array([CTX], np.int64)
    slots = bt[0, pos // BS].astype(np.int64) * BS + pos % BS
    got = eng.forward_raw(tok, pos, slots, bt, np.array([CTX + 1], np.uint32))
    ref = oracle.forward(tok, pos, slots, bt, np.array([CTX + 1], np.uint32))
    check_logits(got, ref, "decode at context 32769")
    eng.close()


def _drive(eng, rids):
    """run to completion; returns the (n_seqs, is_prefill) trace"""
    trace = []
    while eng.has_unfinished():
        trace.append(eng.step())
    assert all(eng.finished(r) for r in rids)
    return trace


def test_32k_prompt_four_chunks_then_prefix_hit_full_size():
    """Llama-3.1-8B shape, 32 layers: one 32 768-token prompt = exactly 4 prefill steps; resubmitted = ONE prefill step over
    the last block (511 blocks hit) with the same greedy tokens; eight prompts behind a cached 16k prefix prefill only their
    tails; every block returns to the pool / the prefix cache afterwards."""
    cfg = dict(E.LLAMA31_8B)
    nblocks = 3072
    eng = Engine(cfg, num_gpu_blocks=nblocks, max_num_seqs=8, max_model_len=40960, enable_prefix_cache=True, use_graph=True).init_synthetic()
    assert eng.num_gpu_blocks == nblocks
    r = np.random.default_rng(42)
    prompt = r.integers(1000, cfg["vocab_size"] - 1000, size=CTX).astype(np.uint32)
    a = eng.add_request(prompt, max_tokens=4, ignore_eos=True)
    t0 = time.perf_counter()
    tr = _drive(eng, [a])
    cold_s = time.perf_counter() - t0
    assert [p for _, p in tr][:5] == [True, True, True, True, False] and sum(p for _, p in tr) == 4
    ta = eng.times(a)
    b = eng.add_request(prompt, max_tokens=4, ignore_eos=True)
    tr = _drive(eng, [b])
    assert sum(p for _, p in tr) == 1                                  # the 511-block hit leaves 64 tokens to prefill
    tb = eng.times(b)
    ttft_cold, ttft_hit = ta["first_token_ms"] - ta["created_ms"], tb["first_token_ms"] - tb["created_ms"]
    print(f"[config5] TTFT 32768 tokens: cold {ttft_cold:.0f} ms, prefix hit {ttft_hit:.1f} ms; cold run {cold_s:.2f} s")
    assert ttft_hit < ttft_cold / 8
    assert eng.output(a).tolist() == eng.output(b).tolist()            # KV read back from the cached blocks == recomputed
    # ---- eight prompts sharing a 16k prefix
    prefix = r.integers(1000, cfg["vocab_size"] - 1000, size=16384).astype(np.uint32)
    w = eng.add_request(np.concatenate([prefix, prompt[:100]]), max_tokens=2, ignore_eos=True)
    _drive(eng, [w])
    tails = [r.integers(1000, cfg["vocab_size"] - 1000, size=1024).astype(np.uint32) for _ in range(8)]
    rids = [eng.add_request(np.concatenate([prefix, t]), max_tokens=6, ignore_eos=True) for t in tails]
    tr = _drive(eng, rids)
    assert sum(p for _, p in tr) <= 2                                   # 8 x 1024 tail tokens fit one or two prefill steps
    dup = eng.add_request(np.concatenate([prefix, tails[3]]), max_tokens=6, ignore_eos=True)
    _drive(eng, [dup])
    assert eng.output(dup).tolist() == eng.output(rids[3]).tolist()
    eng.close()


def test_kv_plan_from_free_memory_runs_on_hardware():
    """a16 on the GPU: num_gpu_blocks = 0 => the block count comes from free HBM x kv_fraction
    (KVCacheAllocator::plan_allocation, kvcache_allocator.rs:564-707) — the '288 GB plan'"""
    cfg = dict(E.LLAMA3_8B, num_layers=4)
    eng = Engine(cfg, num_gpu_blocks=0, kv_fraction=0.25, max_num_seqs=8, max_model_len=8192, use_graph=False).init_synthetic()
    L = eng.L
    import ctypes as C
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    L.vra_mem_info(C.byref(free_b), C.byref(total_b))
    nb = eng.num_gpu_blocks
    per_block = 64 * 8 * 128 * 2 * 2 * 4                               # BS x Hkv x D x 2 bytes x (K, V) x layers
    assert nb >= 2 and per_block * nb <= 0.30 * total_b.value
    assert per_block * nb >= 0.15 * total_b.value, (nb, total_b.value)   # about a quarter of a (mostly free) 288 GB part
    out = eng.generate([list(range(5, 70))], max_tokens=5, ignore_eos=True)[0]
    assert len(out) == 5
    print(f"[a16] planned {nb} blocks = {per_block * nb / 2**30:.1f} GiB of {total_b.value / 2**30:.0f} GiB")
    eng.close()
