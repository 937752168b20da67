"""Evidence for the logit tolerance (north_star: "logits within 1e-3 for bf16").

bf16 logits of magnitude ~4 are spaced 3e-2 apart, so 1e-3 cannot be met by ANY implementation that rounds logits to bf16
as the reference does (llama.rs:317-319) — including the oracle itself.  These tests measure instead of asserting that:
  * a float64 forward pass with no intermediate rounding (tests/truth64.py) is the yardstick;
  * err(oracle, truth) is what the reference's op-by-op rounding costs by itself;
  * the GPU path must not be further from the truth than 1.5x the oracle's own error (both are printed);
  * the same on the int4 GEMM alone, against BOTH definitions of the W4A16 product: the exact product with one rounding
    (what these kernels and the oracle compute) and Marlin's 'round every dequantised weight to 16 bits first' (what the
    reference's kernel computes, src/utils/gptq.rs:133-178) — the two differ by less than they each differ from float64."""
import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import prefill_inputs, simple_tables, small_cfg
from tests.truth64 import TruthModel
from tests.util import BF16, F16, make_quant, rand_dt, rng, ulp_of
from vllm_rs_amd import ops
from vllm_rs_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quant,dt,arch", [("gptq", BF16, "llama"), ("awq", BF16, "qwen2"), (None, BF16, "llama"), ("gptq", F16, "llama")])
def test_gpu_is_not_further_from_float64_truth_than_the_oracle(quant, dt, arch):
    cfg = small_cfg(quant_method=quant, dtype=dt, arch=arch, attention_bias=(arch == "qwen2"), num_layers=2)
    w = om.make_random_checkpoint(cfg, 5)
    eng = Engine(cfg, num_gpu_blocks=16, max_num_seqs=4, max_model_len=512, use_graph=False).load_weights(w)
    oracle, truth = om.OracleModel(cfg, w, num_blocks=16), TruthModel(cfg, w)
    r = np.random.default_rng(5)
    prompt = r.integers(1, cfg["vocab_size"] - 1, size=90).tolist()
    bt = simple_tables([len(prompt) + 8])
    ids, pos, slots, ctx, cu = prefill_inputs([prompt], bt)
    rows = []
    g = eng.forward_raw(ids, pos, slots, bt, ctx, cu)
    o = oracle.forward(ids, pos, slots, bt, ctx, cu)
    t = truth.forward(ids, pos)
    rows.append((g, o, t))
    seq = list(prompt)
    for _ in range(6):  # greedy decode on the TRUTH's tokens, all three models in step
        seq.append(int(np.argmax(t[0])))
        n = len(seq)
        a = (np.array([seq[-1]], np.uint32), np.array([n - 1], np.int64), np.array([int(bt[0, (n - 1) // 64]) * 64 + (n - 1) % 64], np.int64), bt,
             np.array([n], np.uint32))
        g, o, t = eng.forward_raw(*a), oracle.forward(*a), truth.forward(a[0], a[1])
        rows.append((g, o, t))
    eng.close()
    e_gpu = max(float(np.abs(g - t).max()) for g, o, t in rows)
    e_orc = max(float(np.abs(o - t).max()) for g, o, t in rows)
    r_gpu = float(np.sqrt(np.mean([np.mean((g - t) ** 2) for g, o, t in rows])))
    r_orc = float(np.sqrt(np.mean([np.mean((o - t) ** 2) for g, o, t in rows])))
    d_go = max(float(np.abs(g - o).max()) for g, o, t in rows)
    scale = max(float(np.abs(t).max()) for g, o, t in rows)
    spacing = float(ulp_of(np.float32(scale), dt))
    print(f"[tolerance] {arch}/{quant}/{'bf16' if dt == BF16 else 'f16'}: logit scale {scale:.2f} (storage spacing {spacing:.4f}); "
          f"max |gpu-truth| {e_gpu:.4f}, max |oracle-truth| {e_orc:.4f}, rms {r_gpu:.5f} vs {r_orc:.5f}; max |gpu-oracle| {d_go:.4f}")
    assert e_orc > 1e-3 or dt == F16, "the oracle itself meets 1e-3 here: tighten the engine tolerance"
    assert e_gpu <= 1.5 * e_orc + 0.5 * spacing, (e_gpu, e_orc)
    assert r_gpu <= 1.5 * r_orc + 1e-6, (r_gpu, r_orc)


@pytest.mark.parametrize("M", [1, 4, 16, 32, 300])
@pytest.mark.parametrize("awq", [False, True])
def test_gemm_against_exact_and_marlin_rounding_definitions(M, awq):
    """kernels E (M <= 4), C (5..32) and D (>= 256 rows) against (1) the exact W4A16 product and (2) Marlin's variant with
    every dequantised weight rounded to bf16 before the multiply; float64 of the exact product is the yardstick"""
    K, N, gs, dt = 4096, 4096, 128, BF16
    r = rng(M + 11 * awq)
    q = make_quant(r, K, N, gs, dt, awq)
    x = rand_dt(r, (M, K), dt)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    got = orc.from_dt(ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), ops.dev(q["qzeros"]) if awq else None, M, K, N, gs, awq).numpy(np.uint16, (M, N)), dt)
    exact = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt), dt)
    wd = orc.dequant(q["idx"], q["zeros"], q["scales"], gs, dt)            # round_dt((q - z) * s), [K, N]
    marlin = orc.from_dt(orc.gemm_wdense(x, wd, None, None, dt), dt)          # x . W_rounded, f64 accumulation, one rounding
    z = np.full((K // gs, N), 8.0) if q["zeros"] is None else q["zeros"].astype(np.float64)
    w64 = (q["idx"].astype(np.float64) - np.repeat(z, gs, axis=0)) * np.repeat(orc.from_dt(q["scales"], dt).astype(np.float64), gs, axis=0)
    t64 = orc.from_dt(x, dt).astype(np.float64) @ w64
    # unit: one storage ulp at the ROW's output scale (rms of the exact outputs).  Rounding the weights first perturbs every
    # output by an absolute amount that does not shrink with the output's own magnitude (outputs are cancellations of
    # O(sqrt(K)) larger terms), so a per-element ulp would make the comparison meaningless for the small outputs
    ulp = ulp_of(np.sqrt((exact.astype(np.float64) ** 2).mean(axis=1, keepdims=True)).astype(np.float32), dt)
    own = ulp_of(np.maximum(np.abs(exact), 2.0 ** -6), dt)
    d_ge_own = np.abs(got - exact) / own
    d_ge, d_gm, d_em = np.abs(got - exact) / ulp, np.abs(got - marlin) / ulp, np.abs(exact - marlin) / ulp
    e_g, e_m, e_e = np.abs(got - t64) / ulp, np.abs(marlin - t64) / ulp, np.abs(exact - t64) / ulp
    print(f"[w4a16] M={M} awq={awq} (unit: bf16 ulp at the row rms): gpu vs exact-product oracle max {d_ge.max():.2f} "
          f"({100 * (d_ge > 0).mean():.2f}% differ; max {d_ge_own.max():.2f} in ulps of the element itself); gpu vs Marlin-rounded max {d_gm.max():.2f}; "
          f"exact vs Marlin-rounded max {d_em.max():.2f}; distance to float64: gpu {e_g.max():.3f} (rms {np.sqrt((e_g ** 2).mean()):.3f}), exact oracle "
          f"{e_e.max():.3f} (rms {np.sqrt((e_e ** 2).mean()):.3f}), Marlin-rounded {e_m.max():.3f} (rms {np.sqrt((e_m ** 2).mean()):.3f})")
    assert d_ge_own.max() <= 1.0001                    # same definition as the oracle: one rounding flip at most
    # the reference's definition (weights rounded first) sits a few row-scale ulps away from BOTH exact-product
    # implementations — measured max 4 (8 at 1.2 M outputs), rms ~0.4; what matters is which one is closer to float64:
    assert d_gm.max() <= 16.0 and abs(d_gm.max() - d_em.max()) <= 1.0
    assert np.sqrt((e_g ** 2).mean()) <= np.sqrt((e_m ** 2).mean()) + 0.02   # and not further from float64 than the reference's definition is
