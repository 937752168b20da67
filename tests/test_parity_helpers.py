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
