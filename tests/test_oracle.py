"""Pins the CPU oracle (oracle/vra_oracle.c + oracle/model.py) BEFORE it is trusted as the checker:

* known-answer tests of the public on-disk int4 formats (AutoGPTQ v1 / AutoAWQ GEMM): hand-built
  u32 words -> expected nibble indices, bit-exact, both formats, plus the CDNA4 tile layout round trip
* 16-bit rounding KATs (round-to-nearest-even, NaN)
* every float primitive against an independent numpy float64 restatement (tolerance: 1 storage ulp)
* the whole model against HuggingFace transformers fixtures (tests/golden/hf_*_tiny.npz, made by
  tests/golden/make_hf_golden.py in the build container): logits and greedy tokens
The reference's own tests hold no vectors for this path — "parity unpinned" w.r.t. them.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from oracle.model import OracleModel, make_random_checkpoint  # noqa: E402

BF16, F16, F32 = 0, 1, 2
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------ int4 formats
def test_gptq_word_kat():
    """AutoGPTQ v1: qweight[K/8, N] u32, nibble i of a word (bits 4i..4i+3) is row 8*r + i of that column."""
    qw = np.zeros((2, 3), np.uint32)
    qw[0, 0] = 0x76543210
    qw[1, 0] = 0xFEDCBA98
    qw[0, 1] = 0x0000000F
    qw[1, 2] = 0xA0000000
    idx = orc.gptq_unpack(qw, 16, 3)
    assert idx[:, 0].tolist() == list(range(16))
    assert idx[:, 1].tolist() == [15] + [0] * 15
    assert idx[:, 2].tolist() == [0] * 15 + [10]
    assert (orc.gptq_pack(idx) == qw).all()


def test_gptq_zeros_kat():
    """qzeros[G, N/8] packed along N, stored value = z - 1 (0x77777777 <=> zero point 8 everywhere)."""
    qz = np.array([[0x77777777, 0x76543210]], np.uint32)
    z = orc.gptq_unpack_zeros(qz, 1, 16)
    assert z[0, :8].tolist() == [8] * 8
    assert z[0, 8:].tolist() == [1, 2, 3, 4, 5, 6, 7, 8]


def test_awq_word_kat():
    """AutoAWQ GEMM: qweight[K, N/8] u32, nibble i holds column order[i] with order = [0,2,4,6,1,3,5,7]; zeros not offset."""
    qw = np.array([[0x76543210]], np.uint32)
    idx = orc.awq_unpack(qw, 1, 8)
    assert idx[0].tolist() == [0, 4, 1, 5, 2, 6, 3, 7]        # column c reads nibble rev[c], rev = [0,4,1,5,2,6,3,7]
    cols = np.arange(8, dtype=np.uint8)[None, :]
    assert orc.awq_pack(cols)[0, 0] == 0x75316420            # nibble i = order[i]
    assert (orc.awq_unpack(orc.awq_pack(cols), 1, 8) == cols).all()
    z = orc.awq_unpack_zeros(np.array([[0x75316420]], np.uint32), 1, 8)
    assert z[0].tolist() == list(range(8))                   # raw zero points, no -1 offset


def test_pack_unpack_round_trip_random():
    r = np.random.default_rng(0)
    idx = r.integers(0, 16, size=(256, 64), dtype=np.uint8)
    assert (orc.gptq_unpack(orc.gptq_pack(idx), 256, 64) == idx).all()
    assert (orc.awq_unpack(orc.awq_pack(idx), 256, 64) == idx).all()


def test_tile_layout_definition_and_round_trip():
    """CDNA4 tile layout: word[((nb*KT + kt)*64 + lane)*4 + j], lane = oct*16 + nn, column nb*16 + nn,
    rows kt*128 + j*32 + oct*8 + e, nibble position p holds e = 2p (p < 4) or 2(p-4)+1."""
    r = np.random.default_rng(1)
    K, N = 256, 32
    idx = r.integers(0, 16, size=(K, N), dtype=np.uint8)
    tiled = orc.tile_from_indices(idx).reshape(-1)
    KT = K // 128
    for (nb, kt, lane, j) in [(0, 0, 0, 0), (1, 1, 37, 3), (0, 1, 63, 2), (1, 0, 16, 1)]:
        w = int(tiled[((nb * KT + kt) * 64 + lane) * 4 + j])
        octv, nn = lane >> 4, lane & 15
        for p in range(8):
            e = 2 * p if p < 4 else 2 * (p - 4) + 1
            assert (w >> (4 * p)) & 15 == idx[kt * 128 + j * 32 + octv * 8 + e, nb * 16 + nn]
    assert (orc.tile_to_indices(tiled, K, N) == idx).all()
    # both checkpoint formats repack to the same tiles
    assert (orc.gptq_repack(orc.gptq_pack(idx)).reshape(-1) == tiled).all()
    assert (orc.awq_repack(orc.awq_pack(idx)).reshape(-1) == tiled).all()


# ------------------------------------------------------------------------------------------------ 16-bit rounding
def test_bf16_rounding_kats():
    f = np.array([1.0, 1.00390625, 1.01171875, 1.005859375, -2.5, 3.0e38, 1e-40, np.inf], np.float32)
    b = orc.to_bf16(f)
    back = orc.from_bf16(b)
    assert back[0] == 1.0
    assert back[1] == 1.0              # 1 + 2^-8: tie -> even mantissa
    assert back[2] == 1.015625         # 1 + 3*2^-8: tie -> even (rounds up)
    assert back[3] == 1.0078125        # above the tie
    assert back[4] == -2.5 and np.isinf(back[7])
    assert np.isnan(orc.from_bf16(orc.to_bf16(np.array([np.nan], np.float32))))[0]
    # against the definition: keep the top 16 bits after adding 0x7fff + lsb
    r = np.random.default_rng(3).standard_normal(4096).astype(np.float32)
    u = r.view(np.uint32).astype(np.uint64)
    want = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert (orc.to_bf16(r) == want).all()


def test_f16_rounding_matches_numpy():
    r = (np.random.default_rng(4).standard_normal(4096) * 100).astype(np.float32)
    assert (orc.to_f16(r) == r.astype(np.float16).view(np.uint16)).all()
    assert (orc.from_f16(orc.to_f16(r)) == r.astype(np.float16).astype(np.float32)).all()


# ------------------------------------------------------------------------------------------------ float primitives vs f64
def ulp(ref, dt):
    bits = 8 if dt == BF16 else 11
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - (bits - 1))


def assert_within_ulps(got_bits, ref64, dt, n=1.0, floor=0.0, what=""):
    got = orc.from_dt(got_bits, dt).astype(np.float64)
    tol = np.maximum(n * ulp(ref64, dt), floor)
    bad = np.abs(got - ref64) > tol * 1.0001
    assert not bad.any(), f"{what}: {int(bad.sum())} beyond {n} ulp, worst {np.abs(got - ref64).max()}"


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("awq", [False, True])
@pytest.mark.parametrize("gs", [128, 32, -1])
def test_wna16_gemm_vs_float64(dt, awq, gs):
    r = np.random.default_rng(7)
    M, K, N = 3, 256, 48
    g = gs if gs > 0 else K
    idx = r.integers(0, 16, size=(K, N), dtype=np.uint8)
    zeros = r.integers(0, 16, size=(K // g, N), dtype=np.uint8) if awq else None
    scales = orc.to_dt((0.002 + 0.018 * r.random((K // g, N))).astype(np.float32), dt)
    x = orc.to_dt(r.standard_normal((M, K)).astype(np.float32), dt)
    got = orc.wna16_gemm(x, idx, zeros, scales, gs, dt)
    z = zeros.astype(np.float64) if awq else np.full((K // g, N), 8.0)
    w = (idx.astype(np.float64) - np.repeat(z, g, 0)) * np.repeat(orc.from_dt(scales, dt).astype(np.float64), g, 0)
    ref = orc.from_dt(x, dt).astype(np.float64) @ w
    # contract: the exact W4A16 product with ONE rounding at the output
    assert_within_ulps(got, ref, dt, n=0.51, floor=1e-6, what="wna16_gemm")


def test_dequant_is_marlin_style_single_rounding():
    r = np.random.default_rng(8)
    idx = r.integers(0, 16, size=(128, 16), dtype=np.uint8)
    zeros = np.full((1, 16), 8, np.uint8)
    scales = orc.to_bf16((0.01 * (1 + r.random((1, 16)))).astype(np.float32))
    w = orc.dequant(idx, zeros, scales, 128, BF16)
    ref = (idx.astype(np.float64) - 8.0) * orc.from_bf16(scales).astype(np.float64)
    assert (w == orc.to_bf16(ref.astype(np.float32))).all()


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("awq", [False, True])
def test_marlin_rounded_gemm_is_dequant_then_dense_gemm(dt, awq):
    """orc.wna16_gemm(marlin_rounded=True) — the arithmetic of the reference's Marlin kernels (gptq.rs:116-178: every weight
    rounded to 16 bits before the MMA) — is bit for bit orc.dequant + orc.gemm_wdense, with bias and residual; and it is NOT the
    exact product (the two contracts are distinguishable at these sizes)."""
    r = np.random.default_rng(21 + dt + 2 * awq)
    M, K, N, g = 3, 512, 48, 128
    idx = r.integers(0, 16, size=(K, N), dtype=np.uint8)
    zeros = r.integers(0, 16, size=(K // g, N), dtype=np.uint8) if awq else None
    scales = orc.to_dt((0.002 + 0.018 * r.random((K // g, N))).astype(np.float32), dt)
    x = orc.to_dt(r.standard_normal((M, K)).astype(np.float32), dt)
    bias = orc.to_dt(r.standard_normal(N).astype(np.float32) * 0.1, dt)
    res = orc.to_dt(r.standard_normal((M, N)).astype(np.float32), dt)
    got = orc.wna16_gemm(x, idx, zeros, scales, g, dt, bias, res, marlin_rounded=True)
    w = orc.dequant(idx, zeros if awq else np.full((K // g, N), 8, np.uint8), scales, g, dt)
    assert (got == orc.gemm_wdense(x, w, bias, res, dt)).all()
    exact = orc.wna16_gemm(x, idx, zeros, scales, g, dt, bias, res)
    assert (got != exact).any(), "exact and Marlin-rounded products coincide: the test no longer separates the contracts"


def test_model_weight_rounding_switch():
    """oracle/model.py WEIGHT_ROUNDING: 'marlin' routes every int4 linear through the Marlin-rounded product"""
    from oracle import model as om
    cfg = dict(arch="llama", hidden_size=128, intermediate_size=256, num_layers=1, num_heads=2, num_kv_heads=1, head_dim=64, vocab_size=64,
               max_position_embeddings=64, rms_norm_eps=1e-5, rope_theta=10000.0, quant_method="gptq", group_size=128, dtype=BF16)
    w = om.make_random_checkpoint(cfg, 3)
    m = om.OracleModel(cfg, w, num_blocks=2)
    ids, pos, bt = np.arange(5, dtype=np.uint32), np.arange(5, dtype=np.int64), np.array([[0]], np.uint32)
    a = m.forward(ids, pos, pos.copy(), bt, [5], [0, 5])
    try:
        om.WEIGHT_ROUNDING = "marlin"
        m.reset_cache()
        b = m.forward(ids, pos, pos.copy(), bt, [5], [0, 5])
    finally:
        om.WEIGHT_ROUNDING = "exact"
    m.reset_cache()
    assert (m.forward(ids, pos, pos.copy(), bt, [5], [0, 5]) == a).all()
    assert (a != b).any() and np.abs(a - b).max() < 0.25 * np.abs(a).max()


@pytest.mark.parametrize("dt", [BF16, F16])
def test_rms_norm_silu_add_vs_float64(dt):
    r = np.random.default_rng(9)
    x = orc.to_dt(r.standard_normal((5, 96)).astype(np.float32) * 3, dt)
    w = orc.to_dt((1 + 0.1 * r.standard_normal(96)).astype(np.float32), dt)
    xf, wf = orc.from_dt(x, dt).astype(np.float64), orc.from_dt(w, dt).astype(np.float64)
    ref = xf / np.sqrt((xf * xf).mean(-1, keepdims=True) + 1e-5) * wf
    assert_within_ulps(orc.rms_norm(x, w, 1e-5, dt), ref, dt, n=1.0, what="rms_norm")
    g = orc.to_dt(r.standard_normal((5, 96)).astype(np.float32) * 2, dt)
    gf = orc.from_dt(g, dt).astype(np.float64)
    silu = orc.from_dt(orc.to_dt((gf / (1 + np.exp(-gf))).astype(np.float32), dt), dt).astype(np.float64)  # silu rounded, then * up
    assert_within_ulps(orc.silu_mul(g, x, dt), silu * xf, dt, n=1.0, floor=1e-6, what="silu_mul")
    assert_within_ulps(orc.add(x, g, dt), xf + gf, dt, n=0.51, floor=1e-6, what="add")


def ref_attention(q, k, v, scale):
    """q [Tq,Hq,D], k/v [Tk,Hkv,D] float64, causal with the query block at the END of the keys."""
    Tq, Hq, D = q.shape
    Tk, Hkv, _ = k.shape
    out = np.zeros_like(q)
    for h in range(Hq):
        kh = h // (Hq // Hkv)
        s = q[:, h] @ k[:, kh].T * scale
        for i in range(Tq):
            s[i, Tk - Tq + i + 1:] = -np.inf
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        out[:, h] = p @ v[:, kh]
    return out


@pytest.mark.parametrize("dt", [BF16, F16])
def test_paged_attention_shuffled_blocks_with_cached_prefix(dt):
    """decode and chunked prefill over a paged cache whose block table is a random permutation; the first
    `cached` tokens are already in the cache (prefix hit), the query block is the tail (attention.rs:808-820)."""
    r = np.random.default_rng(10)
    Hq, Hkv, D, BS, NB = 4, 2, 32, 8, 16
    ctx, cached = 29, 16
    k = orc.to_dt(r.standard_normal((ctx, Hkv, D)).astype(np.float32), dt)
    v = orc.to_dt(r.standard_normal((ctx, Hkv, D)).astype(np.float32), dt)
    q = orc.to_dt(r.standard_normal((ctx - cached, Hq, D)).astype(np.float32), dt)
    table = r.permutation(NB)[:4].astype(np.uint32)
    slots = np.array([int(table[p // BS]) * BS + p % BS for p in range(ctx)], np.int64)
    kc = np.zeros((NB, Hkv, BS, D), np.uint16)
    vc = np.zeros((NB, Hkv, D, BS), np.uint16)
    orc.reshape_and_cache(k, v, kc, vc, slots, BS, dt)
    # cache geometry: K [NB,Hkv,BS,D], V [NB,Hkv,D,BS]
    p = 13
    assert (kc[table[p // BS], 1, p % BS, :] == k[p, 1]).all() and (vc[table[p // BS], 0, :, p % BS] == v[p, 0]).all()
    kf, vf, qf = (orc.from_dt(a, dt).astype(np.float64) for a in (k, v, q))
    out = orc.paged_attention(q, kc, vc, table[None, :], np.array([ctx], np.uint32), np.array([0, ctx - cached], np.uint32), Hkv, BS, D ** -0.5, dt)
    assert_within_ulps(out, ref_attention(qf, kf, vf, D ** -0.5), dt, n=1.5, floor=2e-3 if dt == BF16 else 3e-4, what="prefill over prefix")
    # decode: one query = the last token
    out1 = orc.paged_attention(q[-1:], kc, vc, table[None, :], np.array([ctx], np.uint32), np.array([0, 1], np.uint32), Hkv, BS, D ** -0.5, dt)
    assert_within_ulps(out1, ref_attention(qf[-1:], kf, vf, D ** -0.5), dt, n=1.5, floor=2e-3 if dt == BF16 else 3e-4, what="decode")
    # negative slots are skipped (padded graph lanes, Appendix A6 fix)
    kc2 = kc.copy()
    orc.reshape_and_cache(k[:1], v[:1], kc2, vc.copy(), np.array([-1], np.int64), BS, dt)
    assert (kc2 == kc).all()


def test_rope_rotate_half_convention():
    """non-interleaved (GPT-NeoX / HF rotate_half): out[i] = x[i]cos - x[i+D/2]sin ; out[i+D/2] = x[i+D/2]cos + x[i]sin."""
    D, T = 16, 5
    r = np.random.default_rng(11)
    cos, sin = orc.rope_tables(D, 10000.0, 64)
    x = r.standard_normal((T, 2, D)).astype(np.float32)
    pos = np.array([0, 1, 7, 33, 63], np.int64)
    got = orc.rope(x, cos, sin, pos, False, F32, F32)
    c, s = cos[pos][:, None, :].astype(np.float64), sin[pos][:, None, :].astype(np.float64)
    x1, x2 = x[..., : D // 2].astype(np.float64), x[..., D // 2:].astype(np.float64)
    ref = np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], -1)
    assert np.abs(got - ref).max() < 1e-6
    assert (got[0] == x[0]).all()   # position 0 is the identity


def test_argmax_first_max_and_causal_mask():
    lg = np.array([[1.0, 5.0, 5.0, -1.0], [np.float32(-np.inf), -3.0, -3.0, -3.0]], np.float32)
    assert orc.argmax_f32(lg).tolist() == [1, 1]      # first maximal index (candle semantics)
    m = orc.from_bf16(orc.causal_mask(4, 0, BF16))
    assert (np.triu(np.ones((4, 4)), 1) * (m == -np.inf) == np.triu(np.ones((4, 4)), 1)).all() and (np.tril(m) == 0).all()


# ------------------------------------------------------------------------------------------------ whole model vs HuggingFace
def load_golden(name):
    z = np.load(os.path.join(GOLD, name))
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    w = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    return cfg, w, z["prompt"], z["logits"], z["tokens"]


def oracle_generate(cfg, w, prompt, forced, dt):
    """prompt, then one decode step per token of `forced` (teacher forcing with the fixture's tokens, so that every
    step is comparable even if a near-tie flips a greedy choice) through the oracle's paged cache; the block table
    is deliberately not the identity.  Returns logits [1 + len(forced), V] and the oracle's own greedy tokens."""
    BS = 8
    cfg = dict(cfg, dtype=dt)
    if dt == F16:  # the fixture's weights are exactly representable in both 16-bit types
        w = {k: orc.to_f16(orc.from_bf16(v)) for k, v in w.items()}
    m = OracleModel(cfg, w, num_blocks=16, block_size=BS)
    table = np.array([5, 2, 9, 0, 7, 1], np.uint32)
    slot = lambda p: int(table[p // BS]) * BS + p % BS
    n = len(prompt)
    logits = [m.forward(prompt, np.arange(n, dtype=np.int64), np.array([slot(p) for p in range(n)], np.int64), table[None, :],
                        np.array([n], np.uint32), np.array([0, n], np.uint32))[0]]
    for s, tok in enumerate(forced):
        pos = n + s
        logits.append(m.forward(np.array([tok], np.uint32), np.array([pos], np.int64), np.array([slot(pos)], np.int64), table[None, :],
                                np.array([pos + 1], np.uint32), None)[0])
    logits = np.stack(logits)
    return logits, orc.argmax_f32(logits).tolist()


def check_against_hf(logits, toks, hf_logits, hf_tokens, rel, what):
    scale = np.abs(hf_logits).max()
    err = np.abs(logits - hf_logits).max() / scale
    assert err < rel, f"{what}: max |dlogit| / max|logit| = {err:.2e}"
    for s, (a, b) in enumerate(zip(toks, hf_tokens)):
        if a != b:  # only a near-tie may flip the greedy choice
            assert hf_logits[s][b] - hf_logits[s][a] < 2 * rel * scale, f"{what} step {s}: {a} vs HF {b} is not a near-tie"
    return err


@pytest.mark.parametrize("name", ["hf_llama_tiny.npz", "hf_qwen2_tiny.npz", "hf_qwen3_tiny.npz"])
@pytest.mark.parametrize("dt,rel", [(F16, 4e-3), (BF16, 3e-2)])
def test_oracle_model_matches_huggingface(name, dt, rel):
    """HF evaluates in float32; the oracle rounds every op's output to the storage type like the reference does, so
    the tolerance is a few storage ulps of the logit scale: 4e-3 for f16 (11 bits), 3e-2 for bf16 (8 bits)."""
    cfg, w, prompt, hf_logits, hf_tokens = load_golden(name)
    logits, toks = oracle_generate(cfg, w, prompt, hf_tokens[:-1].tolist(), dt)
    check_against_hf(logits, toks, hf_logits, hf_tokens.tolist(), rel, f"{name} dt={dt}")


def test_oracle_quantized_model_consistent_with_its_dequantized_twin():
    """a GPTQ and an AWQ toy model against the SAME model with explicitly dequantised dense weights: pins the quant
    path of the whole-model oracle to its dense path (which HF pins above)."""
    for qm in ("gptq", "awq"):
        cfg = dict(arch="llama", hidden_size=128, intermediate_size=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=32,
                   vocab_size=64, max_position_embeddings=64, rms_norm_eps=1e-5, rope_theta=10000.0, quant_method=qm, group_size=128,
                   dtype=F16)
        w = make_random_checkpoint(cfg, seed=5)
        dense = {}
        for k, v in w.items():
            if k.endswith(".qweight"):
                p = k[:-8]
                sc = w[p + ".scales"]
                G, N = sc.shape
                if qm == "awq":
                    K = v.shape[0]
                    idx, z = orc.awq_unpack(v, K, N), orc.awq_unpack_zeros(w[p + ".qzeros"], G, N).astype(np.float64)
                else:
                    K = v.shape[0] * 8
                    idx, z = orc.gptq_unpack(v, K, N), np.full((G, N), 8.0)
                wd = (idx.astype(np.float64) - np.repeat(z, K // G, 0)) * np.repeat(orc.from_f16(sc).astype(np.float64), K // G, 0)
                dense[p + ".weight"] = orc.to_f16(wd.T.astype(np.float32))   # [N, K]; rounds each weight once
            elif not k.endswith((".qzeros", ".scales", ".g_idx")):
                dense[k] = v
        ids = np.arange(1, 11, dtype=np.uint32)
        args = (ids, np.arange(10, dtype=np.int64), np.arange(10, dtype=np.int64), np.array([[0, 1]], np.uint32), np.array([10], np.uint32),
                np.array([0, 10], np.uint32))
        a = OracleModel(cfg, w, 4, 8).forward(*args)
        b = OracleModel(dict(cfg, quant_method=None), dense, 4, 8).forward(*args)
        assert np.abs(a - b).max() / np.abs(b).max() < 5e-3, qm


def test_rope_tables_dynamic_and_yarn_bit_exact_host_vs_oracle_and_numpy_kat():
    """rotary_emb.rs:281-415,435-541: the product's table builder (vra_rope_tables_f32, host code: no GPU) against the oracle's
    restatement, bit for bit, and both against a third evaluation in numpy float32 of the published formulas"""
    import ctypes as C

    from vllm_rs_amd import _lib
    from vllm_rs_amd.engine import model_config
    L = _lib.load()
    base = dict(hidden_size=256, intermediate_size=512, num_layers=1, num_heads=4, num_kv_heads=2, head_dim=64, vocab_size=512,
                max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0)
    f32 = np.float32
    for rs in (dict(rope_type="dynamic", factor=4.0, original_max_position_embeddings=1024), dict(rope_type="dynamic", alpha=2.5),
               dict(rope_type="dynamic", factor=2.0),  # no original_max_position_embeddings: max_position_embeddings / factor (rotary_emb.rs:150-164)
               dict(rope_type="yarn", factor=4.0, original_max_position_embeddings=1024),
               dict(rope_type="yarn", factor=8.0, original_max_position_embeddings=512, beta_fast=16.0, beta_slow=2.0, attn_factor=0.9)):
        mc = model_config(dict(base, rope_scaling=rs))
        n, d = 4096, 64
        c, s_ = np.empty((n, d // 2), f32), np.empty((n, d // 2), f32)
        L.vra_rope_tables_f32(C.byref(mc), n, c.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p))
        st = {"dynamic": 3, "yarn": 4}[rs["rope_type"]]
        omax = mc.rope_original_max_position
        oc, os_ = orc.rope_tables_ext(d, 10000.0, n, st, rs.get("alpha", rs.get("factor", 1.0)), omax, "alpha" in rs, rs.get("beta_fast", 32.0),
                                      rs.get("beta_slow", 1.0), rs.get("attn_factor", 1.0), rs.get("extrapolation_factor", 1.0))
        assert np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and np.array_equal(s_.view(np.uint32), os_.view(np.uint32)), rs
        # ---- numpy restatement (same formulas, numpy's float32 elementary functions): agreement to a few float32 ulps of the angle
        i = np.arange(0, d, 2)
        if st == 3:
            fct = rs.get("alpha", rs.get("factor"))
            sc = fct if "alpha" in rs else fct * int(omax * fct) / omax - (fct - 1.0)
            inv = (f32(1.0) / (np.float64(10000.0 * sc) ** (d / (d - 2))) ** (i / d)).astype(f32) if False else (1.0 / ((10000.0 * sc) ** (d / (d - 2))) ** (i / d)).astype(f32)
            ms = f32(1.0)
        else:
            fac, bf, bs = f32(rs["factor"]), f32(rs.get("beta_fast", 32.0)), f32(rs.get("beta_slow", 1.0))
            pw = np.power(f32(10000.0), (i / d).astype(f32)).astype(f32)
            cd = lambda r: f32(d) * np.log(f32(omax) / (r * f32(2.0 * np.pi)), dtype=f32) / (f32(2.0) * np.log(f32(10000.0), dtype=f32))
            low, high = max(np.floor(cd(bf)), f32(0.0)), min(np.ceil(cd(bs)), f32(d - 1))
            ramp = np.clip((np.arange(d // 2, dtype=f32) - low) * f32(1.0 / (float(high) - float(low))), 0, 1).astype(f32)
            mask = ((f32(1.0) - ramp) * f32(rs.get("extrapolation_factor", 1.0))).astype(f32)
            inv = ((f32(1.0) / (fac * pw)) * (f32(1.0) - mask) + (f32(1.0) / pw) * mask).astype(f32)
            ms = f32((0.1 * np.log(fac, dtype=f32) + 1.0 if fac > 1 else 1.0)) * f32(rs.get("attn_factor", 1.0))
        ang = (np.arange(n, dtype=f32)[:, None] * inv[None, :]).astype(f32)
        assert np.abs(np.cos(ang) * ms - c).max() < 2e-3 and np.abs(np.sin(ang) * ms - s_).max() < 2e-3, rs  # |d angle| ~ 4096 * 1e-7 rad


def test_oracle_paged_attention_sliding_window_matches_the_additive_mask():
    """attention.rs:607-616 + mask.rs:18-54: the windowed paged attention == softmax((q k^T) * scale + causal_mask(L, W)) v, row by row"""
    r = np.random.default_rng(3)
    Hq, Hkv, D, BS, L, W = 4, 2, 64, 64, 90, 17
    k = orc.to_bf16(r.standard_normal((L, Hkv, D)).astype(np.float32))
    v = orc.to_bf16(r.standard_normal((L, Hkv, D)).astype(np.float32))
    q = orc.to_bf16(r.standard_normal((L, Hq, D)).astype(np.float32))
    kc, vc = np.zeros((2, Hkv, BS, D), np.uint16), np.zeros((2, Hkv, D, BS), np.uint16)
    orc.reshape_and_cache(k, v, kc, vc, np.arange(L, dtype=np.int64), BS, 0)
    bt = np.array([[0, 1]], np.uint32)
    got = orc.from_bf16(orc.paged_attention(q, kc, vc, bt, np.array([L], np.uint32), np.array([0, L], np.uint32), Hkv, BS, D ** -0.5, 0, sliding_window=W))
    mask = orc.from_bf16(orc.causal_mask(L, W, 0)).astype(np.float64)
    kf, vf, qf = orc.from_bf16(k).astype(np.float64), orc.from_bf16(v).astype(np.float64), orc.from_bf16(q).astype(np.float64)
    for h in range(Hq):
        s_ = qf[:, h] @ kf[:, h // 2].T * D ** -0.5 + mask
        p_ = np.exp(s_ - s_.max(-1, keepdims=True))
        ref = (p_ / p_.sum(-1, keepdims=True)) @ vf[:, h // 2]
        assert np.abs(got[:, h] - ref).max() < 2e-2
    full = orc.from_bf16(orc.paged_attention(q, kc, vc, bt, np.array([L], np.uint32), np.array([0, L], np.uint32), Hkv, BS, D ** -0.5, 0))
    wide = orc.from_bf16(orc.paged_attention(q, kc, vc, bt, np.array([L], np.uint32), np.array([0, L], np.uint32), Hkv, BS, D ** -0.5, 0, sliding_window=L))
    assert np.array_equal(full, wide)
