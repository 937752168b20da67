"""GPU parity tests of the FP8 (OCP E4M3) KV cache — SURVEY §8f-4; the reference's `fp8_kvcache` option
(EngineConfig.fp8_kvcache, src/utils/config.rs:316; KVCacheAllocator dtype_size 1 / cache dtype U8,
src/utils/kvcache_allocator.rs:188-193,776; PagedAttention::new(.., fp8_kvcache), src/models/layers/attention.rs:607-616).

Byte work is bit-exact: every cache byte written by reshape_and_cache / the fused decode kernel equals the oracle's
round-to-nearest-even, saturating E4M3 encoding.  Attention over the FP8 cache widens the bytes exactly (E4M3 is a subset
of bf16 and f16) and is compared with the oracle's attention over the same bytes at the tolerance of the 16-bit cache tests."""
import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import check_logits, prefill_inputs, simple_tables, small_cfg
from tests.util import BF16, F16, assert_close_dt, rand_dt, rng
from vllm_rs_amd import _lib, ops
from vllm_rs_amd.engine import Engine, model_config

pytestmark = pytest.mark.gpu


def _setup(r, ctxs, Hkv, D, BS, dt, NB=64):
    B = len(ctxs)
    mb = max((c + BS - 1) // BS for c in ctxs)
    perm = r.permutation(NB)
    bt = np.zeros((B, mb), np.uint32)
    nxt, ks, vs, slots = 0, [], [], []
    for b, c in enumerate(ctxs):
        nb = (c + BS - 1) // BS
        bt[b, :nb] = perm[nxt:nxt + nb]
        nxt += nb
        ks.append(rand_dt(r, (c, Hkv, D), dt, 2.0))
        vs.append(rand_dt(r, (c, Hkv, D), dt, 2.0))
        slots.append(np.array([int(bt[b, j // BS]) * BS + j % BS for j in range(c)], np.int64))
    return bt, mb, np.concatenate(ks), np.concatenate(vs), np.concatenate(slots)


@pytest.mark.parametrize("dt", [BF16, F16])
def test_cache_write_is_rne_saturating_e4m3(dt):
    Hkv, D, BS, NB = 4, 128, 64, 8
    r = rng(1 + dt)
    T = 200
    k = rand_dt(r, (T, Hkv, D), dt, 3.0)
    v = rand_dt(r, (T, Hkv, D), dt, 3.0)
    # the interesting values: beyond +-448 (saturate), ties between codes, subnormals of E4M3, zeros, tiny values
    special = orc.to_dt(np.array([500.0, -1e4, 448.0, 464.0, 17.0, 19.0, 0.0009765625, 0.0029296875, 1e-4, -0.0, 0.3, -7.5], np.float32), dt)
    k[0, 0, :12], v[1, 2, :12] = special, special
    slots = r.permutation(NB * BS)[:T].astype(np.int64)
    slots[5] = -1                                        # padded lane: nothing written
    kc_ref, vc_ref = np.zeros((NB, Hkv, BS, D), np.uint8), np.zeros((NB, Hkv, D, BS), np.uint8)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, slots, BS, dt, orc.FP8)
    kc, vc = ops.DevBuf(kc_ref.nbytes).zero(), ops.DevBuf(vc_ref.nbytes).zero()
    pa = ops.PagedAttention(8, D, D ** -0.5, Hkv, BS, dt, fp8_kvcache=True)
    pa.reshape_and_cache(ops.dev(k), ops.dev(v), kc, vc, ops.dev(slots), T)
    assert np.array_equal(kc.numpy(np.uint8, kc_ref.shape), kc_ref)
    assert np.array_equal(vc.numpy(np.uint8, vc_ref.shape), vc_ref)


@pytest.mark.parametrize("Hq,Hkv,D,dt", [(32, 8, 128, BF16), (8, 1, 128, F16), (4, 2, 64, BF16)])
@pytest.mark.parametrize("ctxs", [[1], [31, 32, 33], [64, 65, 200, 7], [3000, 2049]])
def test_decode_attention_over_fp8_cache(Hq, Hkv, D, dt, ctxs):
    BS, NB = 64, 128
    r = rng(Hq + D + sum(ctxs))
    bt, mb, k, v, slots = _setup(r, ctxs, Hkv, D, BS, dt, NB)
    kc_ref, vc_ref = np.zeros((NB, Hkv, BS, D), np.uint8), np.zeros((NB, Hkv, D, BS), np.uint8)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, slots, BS, dt, orc.FP8)
    kc, vc = ops.DevBuf(kc_ref.nbytes).fill_bytes(0x7F), ops.DevBuf(vc_ref.nbytes).fill_bytes(0x7F)   # E4M3 NaN poison
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt, fp8_kvcache=True)
    pa.reshape_and_cache(ops.dev(k), ops.dev(v), kc, vc, ops.dev(slots), len(slots))
    B = len(ctxs)
    q = rand_dt(r, (B, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    ws = ops.DevBuf(ops.lib().vra_paged_attention_decode_workspace_bytes(B, Hq, D, max(ctxs))) if max(ctxs) > 2000 else None
    out = pa.forward_decode(ops.dev(q), kc, vc, ops.dev(bt), ops.dev(cl), B, mb, max(ctxs), ws)
    ref = orc.paged_attention(q, kc_ref, vc_ref, bt, cl, None, Hkv, BS, D ** -0.5, dt, kv_dt=orc.FP8)
    # K / V drawn at twice the scale of the 16-bit cache tests: the absolute floor (P rounded to the storage dtype x |V|) doubles too
    assert_close_dt(out.numpy(np.uint16, (B, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name="decode over fp8 cache",
                    abs_floor=8e-3 if dt == BF16 else 1.5e-3)


def test_prefill_attention_over_fp8_cache_with_prefix():
    Hq, Hkv, D, BS, dt, NB = 8, 2, 128, 64, BF16, 64
    r = rng(9)
    lens_q, prefix = [5, 70, 1, 33], [0, 64, 130, 0]
    ctxs = [a + b for a, b in zip(lens_q, prefix)]
    bt, mb, k, v, slots = _setup(r, ctxs, Hkv, D, BS, dt, NB)
    kc_ref, vc_ref = np.zeros((NB, Hkv, BS, D), np.uint8), np.zeros((NB, Hkv, D, BS), np.uint8)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, slots, BS, dt, orc.FP8)
    kc, vc = ops.DevBuf(kc_ref.nbytes).fill_bytes(0x7F), ops.DevBuf(vc_ref.nbytes).fill_bytes(0x7F)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt, fp8_kvcache=True)
    pa.reshape_and_cache(ops.dev(k), ops.dev(v), kc, vc, ops.dev(slots), len(slots))
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.uint32)
    Tq = int(cu_q[-1])
    q = rand_dt(r, (Tq, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    out = pa.forward_prefill(ops.dev(q), Tq, max(lens_q), ops.dev(cu_q), len(ctxs), k_cache=kc, v_cache=vc, block_tables=ops.dev(bt),
                             context_lens=ops.dev(cl), max_blocks=mb)
    ref = orc.paged_attention(q, kc_ref, vc_ref, bt, cl, cu_q, Hkv, BS, D ** -0.5, dt, kv_dt=orc.FP8)
    assert_close_dt(out.numpy(np.uint16, (Tq, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name="prefill over fp8 cache", abs_floor=8e-3)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("ctxs", [[1, 40, 65, 0], [300, 129], [511, 512, 513, 1030], [5000, 2100]])
def test_fused_decode_with_fp8_cache(dt, ctxs):
    """RoPE + cache write + attention in one launch over an FP8 cache: the new token's K row / V column reach the cache as
    E4M3 bytes (bit-exact) and enter THIS step's attention as the values a later read returns"""
    # (round 5: contexts at the split boundaries of the small-batch rule — 512 tokens = 16 tiles: 2 splits, 1024: 4 — and beyond
    # block 63, where a wave of the latency form re-bases its block-id vector: attention.hip `decode_nsplit`, LAT)
    Hq, Hkv, D, BS, NB = 8, 2, 128, 64, 128 if max(ctxs) > 400 else 32
    r = rng(sum(ctxs) + dt)
    B = len(ctxs)
    mb = max(max((c + BS - 1) // BS for c in ctxs), 1)
    perm = r.permutation(NB)
    bt = np.zeros((B, mb), np.uint32)
    nxt, hk, hv, hs = 0, [], [], []
    for b, c in enumerate(ctxs):
        nb = (c + BS - 1) // BS
        bt[b, :nb] = perm[nxt:nxt + nb]
        nxt += nb
        if c > 1:
            hk.append(rand_dt(r, (c - 1, Hkv, D), dt, 2.0))
            hv.append(rand_dt(r, (c - 1, Hkv, D), dt, 2.0))
            hs.append(np.array([int(bt[b, j // BS]) * BS + j % BS for j in range(c - 1)], np.int64))
    kc_ref, vc_ref = np.zeros((NB, Hkv, BS, D), np.uint8), np.zeros((NB, Hkv, D, BS), np.uint8)
    kc, vc = ops.DevBuf(kc_ref.nbytes).fill_bytes(0x7F), ops.DevBuf(vc_ref.nbytes).fill_bytes(0x7F)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt, fp8_kvcache=True)
    hk, hv, hs = np.concatenate(hk), np.concatenate(hv), np.concatenate(hs)
    orc.reshape_and_cache(hk, hv, kc_ref, vc_ref, hs, BS, dt, orc.FP8)
    pa.reshape_and_cache(ops.dev(hk), ops.dev(hv), kc, vc, ops.dev(hs), len(hs))
    q, k, v = rand_dt(r, (B, Hq, D), dt), rand_dt(r, (B, Hkv, D), dt, 2.0), rand_dt(r, (B, Hkv, D), dt, 2.0)
    pos = np.array([max(c - 1, 0) for c in ctxs], np.int64)
    slots = np.array([int(bt[b, (c - 1) // BS]) * BS + (c - 1) % BS if c > 0 else -1 for b, c in enumerate(ctxs)], np.int64)
    cos, sin = orc.rope_tables(D, 10000.0, 8192)
    cos, sin = orc.to_dt(cos, dt), orc.to_dt(sin, dt)
    cl = np.array(ctxs, np.uint32)
    qr, kr = orc.rope(q, cos, sin, pos, False, dt, dt), orc.rope(k, cos, sin, pos, False, dt, dt)
    orc.reshape_and_cache(kr, v, kc_ref, vc_ref, slots, BS, dt, orc.FP8)
    ref = orc.paged_attention(qr, kc_ref, vc_ref, bt, cl, None, Hkv, BS, D ** -0.5, dt, kv_dt=orc.FP8)
    wsb = ops.DevBuf(ops.lib().vra_paged_attention_decode_workspace_bytes(B, Hq, D, max(ctxs))) if max(ctxs) > 400 else None
    out = pa.rope_cache_decode(ops.dev(q), ops.dev(k), ops.dev(v), kc, vc, ops.dev(cos), ops.dev(sin), ops.dev(pos), ops.dev(slots),
                               ops.dev(bt), ops.dev(cl), B, mb, max(ctxs), wsb)
    got = out.numpy(np.uint16, (B, Hq, D))
    live = [b for b, c in enumerate(ctxs) if c > 0]
    for b, c in enumerate(ctxs):
        if c == 0:
            assert not got[b].any()
    assert_close_dt(got[live], ref[live], dt, max_ulp=2.0, max_mismatch_frac=0.5, name="fused decode fp8", abs_floor=8e-3 if dt == BF16 else 1.5e-3)
    kc_got, vc_got = kc.numpy(np.uint8, kc_ref.shape), vc.numpy(np.uint8, vc_ref.shape)
    for b in live:
        blk, off = int(slots[b]) // BS, int(slots[b]) % BS
        assert np.array_equal(kc_got[blk, :, off, :], kc_ref[blk, :, off, :]) and np.array_equal(vc_got[blk, :, :, off], vc_ref[blk, :, :, off])


@pytest.mark.parametrize("quant,arch", [("gptq", "llama"), ("awq", "qwen2")])
def test_engine_with_fp8_kvcache_matches_oracle(quant, arch):
    """whole engine (chunked prefill, decode with the fused kernel, hipGraph) over an FP8 cache against the oracle model
    with the same cache format; the KV plan counts one byte per element (twice the blocks for the same memory)"""
    cfg = small_cfg(quant_method=quant, arch=arch, attention_bias=(arch == "qwen2"))
    w = om.make_random_checkpoint(cfg, 8)
    eng = Engine(cfg, num_gpu_blocks=32, max_num_seqs=8, max_model_len=512, use_graph=False, fp8_kvcache=True).load_weights(w)
    oracle = om.OracleModel(cfg, w, num_blocks=32, fp8_kvcache=True)
    r = np.random.default_rng(8)
    prompts = [r.integers(1, cfg["vocab_size"] - 1, size=n).tolist() for n in (50, 131)]
    bt = simple_tables([len(p) + 4 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    ref = oracle.forward(ids, pos, slots, bt, ctx, cu)
    # (the E4M3 grid is 16x coarser than bf16: a 1-ulp flip of a rotated K or of V now moves the CACHED value by up to a whole E4M3
    # step before both sides read the same bytes again — measured max 4.25 storage ulps of the logits against 3.5 with the 16-bit cache)
    check_logits(eng.forward_raw(ids, pos, slots, bt, ctx, cu), ref, f"fp8 kv {quant} prefill", max_ulps=8.0)
    seqs = [list(p) for p in prompts]
    for step in range(3):
        nxt = orc.argmax_f32(ref)
        for s, t in zip(seqs, nxt):
            s.append(int(t))
        a = (np.array([s[-1] for s in seqs], np.uint32), np.array([len(s) - 1 for s in seqs], np.int64),
             np.array([int(bt[b, (len(s) - 1) // 64]) * 64 + (len(s) - 1) % 64 for b, s in enumerate(seqs)], np.int64), bt,
             np.array([len(s) for s in seqs], np.uint32))
        ref = oracle.forward(*a)
        check_logits(eng.forward_raw(*a), ref, f"fp8 kv {quant} decode {step}", max_ulps=8.0)
    eng.close()
    # per_block_bytes halves (kvcache_allocator.rs:447-468 with dtype_size 1)
    L = _lib.load()
    mc = model_config(dict(om_cfg=None, **cfg)) if False else model_config(cfg)
    import ctypes as C
    e16 = _lib.EngineConfig(block_size=64, tp_world_size=1, fp8_kvcache=0)
    e8 = _lib.EngineConfig(block_size=64, tp_world_size=1, fp8_kvcache=1)
    assert L.vra_kv_per_block_bytes(C.byref(mc), C.byref(e16)) == 2 * L.vra_kv_per_block_bytes(C.byref(mc), C.byref(e8))
