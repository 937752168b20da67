"""CPU: the acceptance logic of the parity helpers that separate a sequence amplifying rounding noise from a kernel fault
(tests/test_gpu_engine.py check_logits_conditioned) — no GPU, no library call."""
import numpy as np
import pytest

from tests.test_gpu_engine import BF16, check_logits_conditioned


def _case(rows=8, vocab=64, seed=0):
    r = np.random.default_rng(seed)
    ref = (r.standard_normal((rows, vocab)) * 2).astype(np.float32)
    ulp = 2.0 ** (np.floor(np.log2(np.abs(ref).max(axis=-1, keepdims=True))) - 7)
    return ref, ulp


def test_all_rows_within_the_limit_never_ask_for_the_other_order():
    ref, ulp = _case()
    got = ref + 0.5 * ulp

    def never():
        raise AssertionError("the other order must not be evaluated when every row is within the limit")
    assert check_logits_conditioned(got, ref, "within", BF16, 8.0, never) <= 8.0


def test_a_row_the_oracle_itself_moves_on_is_accepted_and_the_others_keep_the_limit():
    ref, ulp = _case()
    got = ref + 0.5 * ulp
    got[3] = ref[3] + 100 * ulp[3]
    alt = ref.copy()
    alt[3] = ref[3] - 120 * ulp[3]
    assert check_logits_conditioned(got, ref, "amplified row", BF16, 8.0, lambda: alt) <= 8.0


def test_a_row_the_oracle_does_not_move_on_fails():
    ref, ulp = _case()
    got = ref.copy()
    got[3] = ref[3] + 100 * ulp[3]
    alt = ref.copy()
    alt[3] = ref[3] + 10 * ulp[3]
    with pytest.raises(AssertionError, match="row 3"):
        check_logits_conditioned(got, ref, "kernel fault", BF16, 8.0, lambda: alt)


def test_more_than_a_quarter_of_the_rows_out_fails_whatever_the_oracle_does():
    ref, ulp = _case()
    got = ref.copy()
    got[:3] = ref[:3] + 50 * ulp[:3]
    alt = ref + 500 * ulp
    with pytest.raises(AssertionError, match="3 of 8 rows"):
        check_logits_conditioned(got, ref, "shape-wide", BF16, 8.0, lambda: alt)


def test_non_finite_logits_fail():
    ref, ulp = _case()
    got = ref.copy()
    got[2, 5] = np.inf
    with pytest.raises(AssertionError):
        check_logits_conditioned(got, ref, "inf", BF16, 8.0, lambda: ref)


# ---- the oracle's mirror of the engine's dense prefill rule (oracle/model.py dense_prefill_rows): CPU, with a stub for the library
class _StubRule:
    def __init__(self, rows):
        self.rows = rows

    def vra_debug_dense_prefill_min_rows(self):
        return self.rows

    def vra_debug_norm_deferred_mask(self, *a):
        return 0


def _tiny_cfg(**kw):
    cfg = dict(arch="llama", hidden_size=128, intermediate_size=256, num_layers=1, num_heads=2, num_kv_heads=1, head_dim=64, vocab_size=64,
               max_position_embeddings=128, rms_norm_eps=1e-5, rope_theta=10000.0, quant_method="gptq", group_size=128, dtype=BF16)
    cfg.update(kw)
    return cfg


def test_the_dense_prefill_rule_is_a_row_rule_of_quantised_models_only(monkeypatch):
    from oracle import model as om
    cfg = _tiny_cfg()
    monkeypatch.setattr(om, "ENGINE_RULE", None)
    assert not om.dense_prefill_rows(cfg, 10 ** 6), "no engine installed: the reference's arithmetic everywhere"
    monkeypatch.setattr(om, "ENGINE_RULE", _StubRule(768))
    assert om.dense_prefill_rows(cfg, 768) and om.dense_prefill_rows(cfg, 5000) and not om.dense_prefill_rows(cfg, 767)
    assert not om.dense_prefill_rows(_tiny_cfg(quant_method=None), 5000), "dense checkpoints have no int4 GEMM to re-round"
    monkeypatch.setattr(om, "ENGINE_RULE", _StubRule(0))
    assert not om.dense_prefill_rows(cfg, 5000), "0 = the path is switched off"


def test_a_step_above_the_row_rule_is_the_marlin_rounded_forward(monkeypatch):
    """OracleModel.forward switches the int4 GEMMs of such a step to w = rnd((q - z) * s) — the same bits as a forward with
    WEIGHT_ROUNDING = "marlin" — and leaves shorter steps on the exact product; the switch does not leak into the next forward"""
    from oracle import model as om
    cfg = _tiny_cfg()
    w = om.make_random_checkpoint(cfg, 3)
    ids = np.arange(12, dtype=np.uint32) % cfg["vocab_size"]
    pos, slots = np.arange(12, dtype=np.int64), np.arange(12, dtype=np.int64)
    bt, ctx, cu = np.zeros((1, 1), np.uint32), np.array([12], np.uint32), np.array([0, 12], np.uint32)

    def run(rule, rounding):
        monkeypatch.setattr(om, "ENGINE_RULE", rule)
        monkeypatch.setattr(om, "WEIGHT_ROUNDING", rounding)
        return om.OracleModel(cfg, w, num_blocks=2).forward(ids, pos, slots, bt, ctx, cu)

    exact, marlin = run(None, "exact"), run(None, "marlin")
    assert not np.array_equal(exact, marlin), "the two roundings must be distinguishable for this test to mean anything"
    assert np.array_equal(run(_StubRule(8), "exact"), marlin)      # 12 rows >= 8: the dense path's arithmetic
    assert np.array_equal(run(_StubRule(13), "exact"), exact)      # 12 rows < 13: the int4 kernels' exact product
    assert om.Linear.marlin_step is False
