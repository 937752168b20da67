"""GPU parity tests of the device sampler and the repetition penalties (LogitsProcessor,
src/utils/logits_processor.rs:72-345; ModelRunner::sample, src/core/runner.rs:1390-1570).

Index work is bit-exact: the ordered top-k candidate ids (ties by token id) and the set top-p keeps must equal the oracle's
restatement of the reference's loops.  Probabilities: f32 softmax, relative 1e-5.  The draw uses the same counter-hash uniform
on both sides, so tokens are compared one to one (a uniform landing within 1e-6 of a CDF boundary may pick the neighbour)."""
import ctypes as C

import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import small_cfg
from vllm_rs_amd import ops
from vllm_rs_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _sample(logits, k, p, t, seed):
    L = ops.lib()
    B, V = logits.shape
    d_l, d_o = ops.dev(logits.astype(np.float32)), ops.DevBuf(B * 4)
    d_i, d_p = ops.DevBuf(B * 256 * 4), ops.DevBuf(B * 256 * 4)
    L.vra_sample(d_l.ptr, d_o.ptr, B, V, k, p, t, seed, d_i.ptr, d_p.ptr, 0)
    ops.check_error()
    return d_o.numpy(np.uint32, (B,)), d_i.numpy(np.uint32, (B, 256)), d_p.numpy(np.float32, (B, 256))


@pytest.mark.parametrize("V", [512, 32000, 128256])
@pytest.mark.parametrize("k,p,t", [(32, 0.95, 0.7), (256, 0.5, 1.0), (1, -1.0, 1.3), (50, -1.0, 0.8), (0, 0.9, 1.0), (8, 0.999, 0.2), (64, 0.01, 2.0)])
def test_candidates_and_draw_match_the_reference_loops(V, k, p, t):
    r = np.random.default_rng(V + k)
    B = 5
    logits = (r.standard_normal((B, V)) * 3).astype(np.float32)
    logits[1, :2000 if V > 2000 else 200] = np.round(logits[1, :2000 if V > 2000 else 200])  # many exact ties across the k-th place
    logits[2] = orc.from_bf16(orc.to_bf16(logits[2]))                                          # bf16-valued logits, as the model emits them
    logits[3, 7] = logits[3].max() + 30.0                                                     # one dominant token
    seed = 1234 + k
    got, gi, gp = _sample(logits, k, p, t, seed)
    u = orc.hash_unit(seed, B)
    for b in range(B):
        ids, probs = orc.sample_candidates(logits[b], k, p, t)
        n = len(ids)
        assert np.array_equal(gi[b, :n], ids), f"row {b}: candidate ids / order differ"
        assert (gi[b, n:] == 0xFFFFFFFF).all()
        assert np.array_equal(gp[b, :n] > 0, probs > 0), f"row {b}: top-p keeps a different set"
        np.testing.assert_allclose(gp[b, :n], probs, rtol=2e-5, atol=1e-9)
        tok, uu, run = orc.sample_draw(ids, probs, u[b])
        if int(got[b]) != tok:
            near = np.abs(run - uu).min()
            assert near < 2e-6 * max(run[-1], 1e-30) + 1e-9, f"row {b}: token {got[b]} vs {tok}, u not at a boundary ({near})"


def test_full_distribution_sampling_is_the_inverse_cdf():
    """Sampling::All (temperature only): token = first index whose running softmax mass exceeds u"""
    r = np.random.default_rng(3)
    B, V = 64, 4096
    logits = (r.standard_normal((B, V)) * 2).astype(np.float32)
    got, _, _ = _sample(logits, 0, -1.0, 1.0, 77)
    u = orc.hash_unit(77, B)
    bad = 0
    for b in range(B):
        ids, probs = orc.sample_candidates(logits[b], 0, -1.0, 1.0)
        tok, uu, run = orc.sample_draw(ids, probs, u[b])
        if int(got[b]) != tok:
            assert abs(int(got[b]) - tok) <= 1, (b, got[b], tok)   # f32 running sums in another order: a neighbour at most
            bad += 1
    assert bad <= 3


def test_empirical_frequencies_follow_the_clamped_distribution():
    V, k, p, t = 1000, 5, 0.9, 1.0
    r = np.random.default_rng(9)
    row = (r.standard_normal(V) * 2).astype(np.float32)
    ids, probs = orc.sample_candidates(row, k, p, t)
    want = probs / probs.sum()
    B = 4096
    counts = np.zeros(len(ids))
    for s in range(4):
        got, _, _ = _sample(np.tile(row, (B, 1)), k, p, t, 1000 + s)
        assert np.isin(got, ids[probs > 0]).all(), "a token outside the kept candidates was drawn"
        for i, tok in enumerate(ids):
            counts[i] += (got == tok).sum()
    freq = counts / counts.sum()
    assert np.abs(freq - want).max() < 0.02, (freq, want)


@pytest.mark.parametrize("fp,pp", [(0.5, 0.0), (0.0, 1.5), (1.2, 0.3), (1.0, 0.0), (0.0, 0.0)])
def test_penalties_bit_exact(fp, pp):
    r = np.random.default_rng(11)
    B, V, W = 6, 32000, 128
    logits = (r.standard_normal((B, V)) * 3).astype(np.float32)
    ctx = r.integers(0, 300, size=(B, W)).astype(np.uint32)     # few distinct ids: repeated tokens
    ctx[0, 5] = V + 10                                          # out-of-vocabulary ids are ignored
    lens = np.array([128, 128, 1, 0, 77, 128], np.int32)
    L = ops.lib()
    d_l, d_c, d_n = ops.dev(logits), ops.dev(ctx), ops.dev(lens)
    d_f, d_p = ops.dev(np.full(B, fp, np.float32)), ops.dev(np.full(B, pp, np.float32))   # (kept alive until the kernel has run)
    L.vra_apply_penalties(d_l.ptr, d_c.ptr, d_n.ptr, B, W, V, d_f.ptr, d_p.ptr, 0)
    ops.check_error()
    got = d_l.numpy(np.float32, (B, V))
    for b in range(B):
        assert np.array_equal(got[b], orc.apply_penalties(logits[b], ctx[b, :lens[b]].tolist(), fp, pp)), f"row {b}"


def test_engine_stochastic_generation():
    """the engine loop with a stochastic strategy: same seed => same tokens, another seed => other tokens; every sampled token
    lies in the top-k of the oracle's logits for the actual prefix; the reference's default strategy (nothing set) is
    top-k 32 / top-p 0.95 / temperature 0.7 (A4); temperature 0 stays greedy; hipGraph replay + sampler kernels"""
    cfg = small_cfg(quant_method="gptq")
    w = om.make_random_checkpoint(cfg, 2)
    prompt = list(range(5, 60))

    def run(seed, sampling, n=24):
        eng = Engine(cfg, num_gpu_blocks=16, max_num_seqs=4, max_model_len=512, use_graph=True, seed=seed).load_weights(w)
        out = eng.generate([prompt], max_tokens=n, ignore_eos=True, sampling=sampling)[0].tolist()
        eng.close()
        return out
    sp = dict(temperature=1.0, top_k=8, top_p=0.9)
    a, b, c = run(7, sp), run(7, sp), run(8, sp)
    assert a == b and a != c
    greedy = run(7, None)
    assert run(7, dict(temperature=0.0)) == greedy and a != greedy
    dflt = run(7, {})
    # every token of `a` is one of the 8 most probable continuations of its prefix according to the oracle
    oracle = om.OracleModel(cfg, w, num_blocks=16)
    seq = list(prompt)
    bt = np.arange(8, dtype=np.uint32)[None]
    pos = np.arange(len(seq), dtype=np.int64)
    logits = oracle.forward(np.array(seq, np.uint32), pos, pos.copy(), bt, [len(seq)], [0, len(seq)])
    for tok in a:
        top = np.argsort(-logits[0], kind="stable")[:10]   # 2 spare places: near-ties between GPU and oracle logits
        assert tok in top, (tok, top)
        seq.append(tok)
        n = len(seq)
        logits = oracle.forward(np.array([tok], np.uint32), np.array([n - 1], np.int64), np.array([n - 1], np.int64), bt, [n])
    assert len(dflt) == 24


def test_engine_penalties_act_after_128_sampled_tokens():
    """runner.rs:1519-1541: penalties use the last 128 sampled tokens once MORE than 128 were sampled — a greedy run with a
    large presence penalty follows the plain greedy run for 129 tokens and then leaves it"""
    cfg = small_cfg(quant_method="gptq", max_position_embeddings=1024)
    w = om.make_random_checkpoint(cfg, 4)
    prompt = list(range(9, 40))

    def run(sampling):
        eng = Engine(cfg, num_gpu_blocks=32, max_num_seqs=4, max_model_len=1024, use_graph=False).load_weights(w)
        out = eng.generate([prompt], max_tokens=200, ignore_eos=True, sampling=sampling)[0].tolist()
        eng.close()
        return out
    plain = run(dict(temperature=0.0))
    pen = run(dict(temperature=0.0, presence_penalty=50.0))
    assert plain[:129] == pen[:129]
    assert plain != pen, "the penalty never acted"
    first = next(i for i in range(len(plain)) if plain[i] != pen[i])
    # from the first divergence on, a penalised token never repeats one of the 128 before it
    for i in range(first, len(pen)):
        assert pen[i] not in pen[i - 128:i], (i, pen[i])
