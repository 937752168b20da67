"""CPU check of the per-stage TP oracle used to localise a deviating tensor-parallel forward (tests/tp_stages.py): its last stage
reproduces OracleModel.forward's logits, its partials add up to the all-reduced hidden state, and `first_deviation` names the
rank and stage of a planted error."""
import numpy as np

from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import prefill_inputs, simple_tables, small_cfg
from tests.tp_stages import first_deviation, oracle_stages

BF16 = 0


def test_stage_oracle_matches_the_whole_forward_and_localises_a_planted_error():
    cfg = small_cfg(num_layers=1, quant_method="gptq")
    W = 2
    w = om.make_random_checkpoint(cfg, 3)
    a, b = om.OracleModel(cfg, w, num_blocks=8, tp_world=W), om.OracleModel(cfg, w, num_blocks=8, tp_world=W)
    r = np.random.default_rng(3)
    prompts = [r.integers(1, cfg["vocab_size"] - 1, size=n).tolist() for n in (9, 4)]
    bt = simple_tables([len(p) + 2 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    logits = a.forward(ids, pos, slots, bt, ctx, cu)
    st = oracle_stages(b, ids, pos, slots, bt, ctx, cu)
    assert len(st) == W and set(st[0]) == {"q", "k", "v", "attn", "o_partial", "h_after_o", "act", "down_partial", "h_after_down"}
    H = cfg["hidden_size"]
    h2 = st[0]["h_after_down"].reshape(-1, H)
    rows = np.asarray(cu[1:], np.int64) - 1
    x = orc.rms_norm(np.ascontiguousarray(h2[rows]), b.final_norm, cfg["rms_norm_eps"], BF16)
    assert np.array_equal(orc.dense_gemm(x, b.lm_head, None, BF16, 2), logits)
    assert (a.kc[0] == b.kc[0]).all() and (a.vc[0] == b.vc[0]).all()
    assert first_deviation(st, st, BF16) == ""
    bad = [{n: v.copy() for n, v in s.items()} for s in st]
    bad[1]["down_partial"][5] ^= 0x0400  # one value of rank 1's down_proj partial, far beyond 2 ulp
    msg = first_deviation(bad, st, BF16)
    assert "rank 1" in msg and "down_partial" in msg and "rank 0" not in msg
