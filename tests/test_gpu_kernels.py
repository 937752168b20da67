"""GPU parity tests: every HIP entry point of include/vllm_rs_amd.h §A/§B against the CPU oracle,
called through the C ABI (ctypes).  Bit-exact for integer/byte work; for 16-bit float outputs the
GPU accumulates in f32 (MFMA order) while the oracle accumulates in double, so a small fraction of
elements may land on the neighbouring storage value: tolerance = 1 storage ulp on <= 2 % of the
elements (stated per test)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import BF16, F16, F32, assert_close_dt, make_quant, rand_dt, rng, ulp_of
from vllm_rs_amd import ops

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- int4 layout: bit exact
@pytest.mark.parametrize("K,N", [(128, 16), (256, 64), (4096, 1024), (1024, 14336 // 8)])
@pytest.mark.parametrize("awq", [False, True])
def test_repack_bit_exact(K, N, awq):
    r = rng(K + N + awq)
    q = make_quant(r, K, N, 128, BF16, awq)
    d = ops.dev(q["qweight"])
    tiled = ops.marlin_weight_repack(d, q["qweight"].shape, 4, awq)
    got = tiled.numpy(np.uint32, (K // 16, N * 2))
    ref = orc.awq_repack(q["qweight"]) if awq else orc.gptq_repack(q["qweight"])
    assert np.array_equal(got, ref)
    # unpack indices straight from the tiled tensor: bit-exact int4 codes
    idx = ops.unpack_indices(tiled, K, N)
    assert np.array_equal(idx, q["idx"])


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("awq,gs,layout", [(False, 128, 0), (False, 128, 1), (True, 128, 1), (False, 64, 1), (False, -1, 1), (True, 32, 0)])
def test_dequant_bit_exact(dt, awq, gs, layout):
    K, N = 512, 256
    r = rng(7 + gs + layout)
    q = make_quant(r, K, N, gs, dt, awq)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    sc = q["scales"]
    if layout == 1:
        sc = orc.marlin_permute_scales(sc, grouped=(gs > 0 and gs < K))
    got = ops.dequant(tiled, ops.dev(sc), ops.dev(q["qzeros"]) if awq else None, K, N, gs, awq, layout, dt)
    ref = orc.dequant(q["idx"], q["zeros"], q["scales"], gs, dt)
    assert np.array_equal(got, ref)


# ---------------------------------------------------------------- dequant-fused GEMM
GEMM_SHAPES = [(512, 256), (4096, 1024), (1024, 4096), (3584, 512), (8192, 1024), (1024, 8192)]  # the last two: a Llama-3-70B TP=8 rank


@pytest.mark.parametrize("M", [1, 2, 5, 8, 9, 16, 32, 33, 64, 100])
@pytest.mark.parametrize("K,N", GEMM_SHAPES)
def test_wna16_gemm_gptq(M, K, N):
    r = rng(M * 131 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape)
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), None, M, K, N, 128)
    got = out.numpy(np.uint16, (M, N))
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    assert_close_dt(got, ref, BF16, name=f"gemm M={M} K={K} N={N}", abs_floor=2e-3)


@pytest.mark.parametrize("M", [1, 4, 32])
@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("awq", [False, True])
def test_marlin_ffi(M, dt, awq):
    """the reference boundary: marlin_*(in, qweight, scales(permuted), qzeros, g_idx, out, m, k, n, workspace, gs, stream)"""
    K, N, gs = 1024, 512, 128
    r = rng(M + dt + awq * 7)
    q = make_quant(r, K, N, gs, dt, awq)
    x = rand_dt(r, (M, K), dt)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True)
    ws = ops.DevBuf(N * 4).zero()
    out = ops.gptq_matmul(ops.dev(x), tiled, ops.dev(sc), ops.dev(q["qzeros"]), None, ws, 4, gs, awq, M, K, N, dt)
    got = out.numpy(np.uint16, (M, N))
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt)
    assert_close_dt(got, ref, dt, name="marlin ffi", abs_floor=2e-3)
    assert not ws.numpy(np.uint32, (N,)).any(), "workspace must stay zero (wna16.rs:238-242)"


@pytest.mark.parametrize("M", [1, 3, 16, 40])
def test_wna16_bias_residual(M):
    K, N = 512, 256
    r = rng(M)
    q = make_quant(r, K, N, 128, BF16, True)
    x, bias, res = rand_dt(r, (M, K), BF16), rand_dt(r, (N,), BF16), rand_dt(r, (M, N), BF16)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, True)
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), ops.dev(q["qzeros"]), M, K, N, 128, True, 0, ops.dev(bias), ops.dev(res))
    got = out.numpy(np.uint16, (M, N))
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, BF16, bias, res)
    # three roundings (GEMM, +bias, +residual): a 1-ulp flip of an INTERMEDIATE moves the result by
    # one ulp of that intermediate's magnitude, however small the final sum is
    g0 = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, BF16), BF16)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, BF16)[None, :]))
    assert_close_dt(got, ref, BF16, max_ulp=1.0, name="bias+residual", mag=mag)


@pytest.mark.parametrize("M", [1, 2, 8, 12, 32, 48])
def test_gate_up_silu(M):
    K, N = 512, 1024
    r = rng(M + 99)
    qg, qu = make_quant(r, K, N, 128, BF16), make_quant(r, K, N, 128, BF16)
    x = rand_dt(r, (M, K), BF16)
    tg = ops.marlin_weight_repack(ops.dev(qg["qweight"]), qg["qweight"].shape)
    tu = ops.marlin_weight_repack(ops.dev(qu["qweight"]), qu["qweight"].shape)
    out = ops.wna16_gate_up_silu(ops.dev(x), tg, ops.dev(qg["scales"]), None, tu, ops.dev(qu["scales"]), None, M, K, N, 128)
    got = out.numpy(np.uint16, (M, N))
    g = orc.wna16_gemm(x, qg["idx"], None, qg["scales"], 128, BF16)
    u = orc.wna16_gemm(x, qu["idx"], None, qu["scales"], 128, BF16)
    ref = orc.silu_mul(g, u, BF16)
    # a 1-ulp flip of gate or up moves the product by up to ~2 ulp
    assert_close_dt(got, ref, BF16, max_ulp=3.0, max_mismatch_frac=0.04, name="gate_up_silu", abs_floor=4e-3)


# ---------------------------------------------------------------- kernel D (many rows: prefill)
# shapes that the default dispatch routes to gemm_q4_big_kernel: M >= 256 and >= 192 workgroup tiles (256 columns x 128 rows)
@pytest.mark.parametrize("N", [16384, 12288])   # 64 column tiles x 3 row tiles of 128 (4 m-tiles per wave) / 48 x 5 row tiles of 64 (2)
@pytest.mark.parametrize("dt,awq,gs", [(BF16, False, 128), (F16, False, 128), (BF16, True, 128), (BF16, False, -1), (F16, True, 256)])
def test_wna16_gemm_many_rows(dt, awq, gs, N):
    M, K = 300, 512                      # the last row tile is ragged (44 rows)
    r = rng(11 + dt + awq + gs)
    q = make_quant(r, K, N, gs, dt, awq)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    qz = ops.dev(q["qzeros"]) if awq else None
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), qz, M, K, N, gs, awq, 0, None, None, dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"] if awq else None, q["scales"], gs, dt)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, name="kernel D", abs_floor=2e-3)
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), qz, M, K, N, gs, awq, 0, ops.dev(bias), ops.dev(res), dt)
    ref2 = orc.wna16_gemm(x, q["idx"], q["zeros"] if awq else None, q["scales"], gs, dt, bias, res)
    g0 = orc.from_dt(ref, dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    # 4.9 M outputs and three roundings (GEMM, +bias, +residual), each of which may land on the neighbouring value:
    # ~5e-5 of the outputs see two flips, ~2e-7 all three (measured: 263 and 1 elements) => bound 3 ulp of the
    # intermediate magnitude (the 10 k-element test above never sees a double flip)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref2, dt, max_ulp=3.0, name="kernel D bias+residual", mag=mag)


@pytest.mark.parametrize("awq", [False, True])
def test_gate_up_silu_many_rows(awq):
    M, K, N = 260, 512, 8192             # dual: 64 column tiles of 128 x 3 row tiles
    r = rng(23 + awq)
    qg, qu = make_quant(r, K, N, 128, BF16, awq), make_quant(r, K, N, 128, BF16, awq)
    x = rand_dt(r, (M, K), BF16)
    tg = ops.marlin_weight_repack(ops.dev(qg["qweight"]), qg["qweight"].shape, 4, awq)
    tu = ops.marlin_weight_repack(ops.dev(qu["qweight"]), qu["qweight"].shape, 4, awq)
    zg, zu = (ops.dev(qg["qzeros"]), ops.dev(qu["qzeros"])) if awq else (None, None)
    out = ops.wna16_gate_up_silu(ops.dev(x), tg, ops.dev(qg["scales"]), zg, tu, ops.dev(qu["scales"]), zu, M, K, N, 128, awq)
    g = orc.wna16_gemm(x, qg["idx"], qg["zeros"] if awq else None, qg["scales"], 128, BF16)
    u = orc.wna16_gemm(x, qu["idx"], qu["zeros"] if awq else None, qu["scales"], 128, BF16)
    ref = orc.silu_mul(g, u, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, max_ulp=3.0, max_mismatch_frac=0.04, name="kernel D gate_up_silu", abs_floor=4e-3)


@pytest.mark.parametrize("M", [143, 160, 200, 221])
def test_gate_up_silu_143_to_221_rows_at_the_llama3_8b_widths(M):
    """the row range where the launcher's cost model changed its choice in round 5 (kernel B's pair form priced at its measured
    420 TFLOP/s: kernel D with four m-tiles per wave takes these launches now, wna16_gemm.hip `vra_gemm_q4_big_fits`), at the
    real gate/up shape of BASELINE config 2: K = 4096, N = 14336, ragged last row tile at every M"""
    K, N = 4096, 14336
    r = rng(1000 + M)
    qg, qu = make_quant(r, K, N, 128, BF16), make_quant(r, K, N, 128, BF16)
    x = rand_dt(r, (M, K), BF16)
    tg = ops.marlin_weight_repack(ops.dev(qg["qweight"]), qg["qweight"].shape)
    tu = ops.marlin_weight_repack(ops.dev(qu["qweight"]), qu["qweight"].shape)
    out = ops.wna16_gate_up_silu(ops.dev(x), tg, ops.dev(qg["scales"]), None, tu, ops.dev(qu["scales"]), None, M, K, N, 128)
    g = orc.wna16_gemm(x, qg["idx"], None, qg["scales"], 128, BF16)
    u = orc.wna16_gemm(x, qu["idx"], None, qu["scales"], 128, BF16)
    ref = orc.silu_mul(g, u, BF16)
    # 2..3 million outputs per case: the tail of "a 1-ulp accumulation-order flip of gate, amplified by silu" shows up (3..7 elements
    # beyond 3 ulp of the PRODUCT at gate < -4, where d ln silu / d ln g = 1 + g (1 - sigmoid(g)) reaches -4..-7), so the bound is the
    # propagated one: one ulp of gate through silu' (+ half an ulp of the re-rounded silu), one ulp of up, 1.5 ulp of the result
    gf, uf = orc.from_dt(g, BF16).astype(np.float64), orc.from_dt(u, BF16).astype(np.float64)
    sg = 1.0 / (1.0 + np.exp(-gf))
    tol = (ulp_of(gf, BF16) * np.abs(sg * (1.0 + gf * (1.0 - sg))) + 0.5 * ulp_of(gf * sg, BF16)) * np.abs(uf) + ulp_of(uf, BF16) * np.abs(gf * sg) + 1.5 * ulp_of(orc.from_dt(ref, BF16), BF16)
    got = out.numpy(np.uint16, (M, N))
    diff = np.abs(orc.from_dt(got, BF16).astype(np.float64) - orc.from_dt(ref, BF16))
    bad = diff > np.maximum(tol * 1.0001, 4e-3)
    assert not bad.any(), f"gate_up_silu {M} rows: {int(bad.sum())} elements beyond the propagated one-ulp bound; worst excess {float((diff / np.maximum(tol, 4e-3)).max()):.2f}x"
    assert float(((diff > 3.0001 * ulp_of(orc.from_dt(ref, BF16), BF16)) & (diff > 4e-3)).mean()) < 2e-5  # (the plain 3-ulp bound: all but a handful)
    assert float((got != ref).mean()) <= 0.04


@pytest.mark.parametrize("gs,with_g_idx", [(128, True), (64, True), (64, False), (32, False), (128, False)])
def test_gemm_half_q_half_alt(gs, with_g_idx):
    """the non-Marlin GPTQ path (gptq.rs:181-198): asymmetric zeros (stored z-1), groups from g_idx — or, without it, from the
    extent of the scales allocation (the reference's signature carries no group size)"""
    M, K, N = 3, 256, 128
    r = rng(5 + gs)
    q = make_quant(r, K, N, gs, F16, False)
    zeros = r.integers(0, 15, size=(q["G"], N), dtype=np.uint8)  # stored value (z-1)
    qz = np.zeros((q["G"], N // 8), np.uint32)
    for n in range(N):
        qz[:, n // 8] |= zeros[:, n].astype(np.uint32) << (4 * (n % 8))
    x = rand_dt(r, (M, K), F16)
    g_idx = ops.dev((np.arange(K) // gs).astype(np.int32)) if with_g_idx else None
    out = ops.gptq_matmul(ops.dev(x), ops.dev(q["qweight"]), ops.dev(q["scales"]), ops.dev(qz), g_idx, None, 4, gs, False, M, K, N, F16)
    got = out.numpy(np.uint16, (M, N))
    ref = orc.wna16_gemm(x, q["idx"], orc.gptq_unpack_zeros(qz, q["G"], N), q["scales"], gs, F16)
    assert_close_dt(got, ref, F16, name="gptq alt", abs_floor=2e-3)


def test_gemm_half_q_half_alt_refuses_a_scales_view_without_g_idx():
    """without g_idx the group size comes from the extent of the scales allocation; a pointer INTO a larger allocation
    (a candle tensor with a start offset, gptq.rs:67) used to be mis-sized silently — now it is an error, and g_idx still works"""
    M, K, N, gs = 2, 256, 128, 64
    r = rng(77)
    q = make_quant(r, K, N, gs, F16, False)
    qz = np.full((q["G"], N // 8), 0x77777777, np.uint32)
    x = rand_dt(r, (M, K), F16)
    big = ops.DevBuf(q["scales"].nbytes + 4096)
    ops.lib().vra_memcpy_h2d(big.ptr + 1024, q["scales"].ctypes.data, q["scales"].nbytes, 0)
    view = big.ptr + 1024
    with pytest.raises(RuntimeError, match="base of"):
        ops.gptq_matmul(ops.dev(x), ops.dev(q["qweight"]), view, ops.dev(qz), None, None, 4, gs, False, M, K, N, F16)
    g_idx = ops.dev((np.arange(K) // gs).astype(np.int32))
    out = ops.gptq_matmul(ops.dev(x), ops.dev(q["qweight"]), view, ops.dev(qz), g_idx, None, 4, gs, False, M, K, N, F16)
    ref = orc.wna16_gemm(x, q["idx"], orc.gptq_unpack_zeros(qz, q["G"], N), q["scales"], gs, F16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, F16, name="gptq alt (view + g_idx)", abs_floor=2e-3)


@pytest.mark.parametrize("gs,with_g_idx", [(128, True), (64, False)])
def test_gemm_half_q_half_alt_8bit(gs, with_g_idx):
    """8-bit plain GPTQ through the same symbol (utils/mod.rs:1313, wna16.rs:154-176): four values per word, zero = stored + 1"""
    M, K, N = 5, 256, 128
    r = rng(8 + gs)
    G = K // gs
    idx = r.integers(0, 256, size=(K, N), dtype=np.uint8)
    zeros = r.integers(100, 156, size=(G, N), dtype=np.uint8)  # stored value (z-1), near the middle of the range as real checkpoints have it
    scales = orc.to_dt((0.0005 + 0.002 * r.random((G, N))).astype(np.float32), F16)
    qw = np.zeros((K // 4, N), np.uint32)
    for k in range(K):
        qw[k // 4] |= idx[k].astype(np.uint32) << (8 * (k % 4))
    qz = np.zeros((G, N // 4), np.uint32)
    for n in range(N):
        qz[:, n // 4] |= zeros[:, n].astype(np.uint32) << (8 * (n % 4))
    x = rand_dt(r, (M, K), F16)
    g_idx = ops.dev((np.arange(K) // gs).astype(np.int32)) if with_g_idx else None
    out = ops.gptq_matmul(ops.dev(x), ops.dev(qw), ops.dev(scales), ops.dev(qz), g_idx, None, 8, gs, False, M, K, N, F16)
    ref = orc.wna16_gemm(x, idx, (zeros.astype(np.int32) + 1).astype(np.uint8), scales, gs, F16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, F16, name="gptq alt 8-bit", abs_floor=2e-3)


@pytest.mark.parametrize("M", [1, 4, 8, 20, 32])
@pytest.mark.parametrize("out_f32", [False, True])
def test_dense_gemm(M, out_f32):
    K, N = 512, 2048
    r = rng(M)
    x, w = rand_dt(r, (M, K), BF16), rand_dt(r, (N, K), BF16, 0.05)
    out = ops.dense_gemm(ops.dev(x), ops.dev(w), None, M, K, N, BF16, F32 if out_f32 else BF16)
    ref = orc.dense_gemm(x, w, None, BF16, F32 if out_f32 else BF16)
    if out_f32:
        got = out.numpy(np.float32, (M, N))
        # values are bf16-rounded then widened (llama.rs:317-319): compare as bf16 bits
        assert np.array_equal(orc.to_bf16(got).astype(np.uint32) << 16, got.view(np.uint32)), "f32 logits must be bf16-representable"
        assert_close_dt(orc.to_bf16(got), orc.to_bf16(ref), BF16, name="dense f32", abs_floor=2e-3)
    else:
        assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name="dense", abs_floor=2e-3)


@pytest.mark.parametrize("M", [8, 9, 16, 17, 32])
@pytest.mark.parametrize("K,N,dt,with_bias", [(4096, 4112, BF16, False), (2048, 8192, BF16, True), (3584, 4096, F16, False), (1024, 2064, F16, True)])
def test_dense_gemm_decode_batches(M, K, N, dt, with_bias):
    """9..32 rows, K <= 4096: kernel W's dense variant (gemv_dw.cuh) — ragged unit distribution (N/16 not a multiple of the
    grid), waves without a k-tile (K < 4096), bias, both dtypes, f32 output == the model-dtype value widened"""
    r = rng(M + K + N)
    x, w = rand_dt(r, (M, K), dt), rand_dt(r, (N, K), dt, 0.05)
    bias = rand_dt(r, (N,), dt) if with_bias else None
    out = ops.dense_gemm(ops.dev(x), ops.dev(w), ops.dev(bias) if with_bias else None, M, K, N, dt, F32)
    ref = orc.dense_gemm(x, w, bias, dt, F32)
    got = out.numpy(np.float32, (M, N))
    assert np.array_equal(orc.from_dt(orc.to_dt(got, dt), dt), got), "f32 output must be representable in the model dtype"
    g0 = np.abs(orc.dense_gemm(x, w, None, dt, F32)) if with_bias else None  # the sum is rounded before the bias is added
    assert_close_dt(orc.to_dt(got, dt), orc.to_dt(ref, dt), dt, name="dense decode batch", abs_floor=2e-3, mag=g0)
    out16 = ops.dense_gemm(ops.dev(x), ops.dev(w), ops.dev(bias) if with_bias else None, M, K, N, dt, dt)
    assert np.array_equal(out16.numpy(np.uint16, (M, N)), orc.to_dt(got, dt)), "16-bit output == the f32 output narrowed"


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("M,K,N", [(1, 4096, 128256), (2, 2048, 32000), (3, 512, 2048), (8, 1024, 4112), (12, 1024, 4096),
                                   (9, 1024, 32000), (16, 512, 4096), (17, 2048, 16384), (31, 1024, 4112), (32, 4096, 32000)])
def test_dense_gemm_argmax(M, K, N, dt):
    """the greedy token out of the lm_head launch (kernel A's last-arriver reduction at <= 8 rows, the dense W kernel's at 4..32
    rows where it fits — round 4 —, two launches otherwise) ==
    vra_argmax_f32 of the logits the same launch wrote == the oracle's first maximal index; exact ties (duplicated weight
    rows, the maximum planted at chosen columns) resolve to the smaller index; the workspace re-arms itself between launches"""
    r = rng(M * 7 + K + N)
    x, w = rand_dt(r, (M, K), dt), rand_dt(r, (N, K), dt, 0.05)
    f = ops.DenseGemmArgmax()
    dx = ops.dev(x)
    for trial in range(3):
        if trial:  # plant the row-0 winner's weight row at a few other columns: exact ties on both sides of it
            ref = orc.dense_gemm(x, w, None, dt, F32)
            win = int(np.argmax(ref[0]))
            for c in r.integers(0, N, size=3):
                w[int(c)] = w[win]
        logits, toks = f(dx, ops.dev(w), None, M, K, N, dt)
        got = logits.numpy(np.float32, (M, N))
        assert np.array_equal(got, ops.dense_gemm(dx, ops.dev(w), None, M, K, N, dt, F32).numpy(np.float32, (M, N)))
        assert np.array_equal(toks, ops.argmax(logits, M, N)), (trial, toks)
        assert np.array_equal(toks, orc.argmax_f32(got)), (trial, toks)
        assert np.array_equal(toks, np.argmax(got, axis=-1))


def test_dense_gemm_argmax_hand_off_stress_over_all_xcds():
    """the last-arriver reduction of the fused lm_head argmax rests on an ISA-level ordering argument (gemv.cuh: agent-scope
    candidate stores acknowledged at vmcnt(0), a returning counter RMW) instead of a release / acquire pair: 200
    back-to-back launches over every CU of all 8 XCDs, x changing every launch (the candidates of the previous launch are stale
    in the workspace), every token checked against the argmax of the logits of the same launch"""
    M, K, N, dt = 4, 1024, 128256, BF16
    r = rng(99)
    w = ops.dev(rand_dt(r, (N, K), dt, 0.05))
    f = ops.DenseGemmArgmax()
    for it in range(200):
        x = rand_dt(r, (M, K), dt)
        logits, toks = f(ops.dev(x), w, None, M, K, N, dt)
        assert np.array_equal(toks, np.argmax(logits.numpy(np.float32, (M, N)), axis=-1)), it


def test_dense_gemm_argmax_dense_w_hand_off_stress():
    """the same hand-off out of the dense W kernel (gemv_dw.cuh, 9..32 rows: one workgroup per CU, 32 rows of candidates): 100
    back-to-back launches at the Llama-3 vocabulary, x changing every launch, every token against the logits of its launch"""
    M, K, N, dt = 32, 1024, 128256, BF16
    r = rng(98)
    w = ops.dev(rand_dt(r, (N, K), dt, 0.05))
    f = ops.DenseGemmArgmax()
    for it in range(100):
        x = rand_dt(r, (M, K), dt)
        logits, toks = f(ops.dev(x), w, None, M, K, N, dt)
        assert np.array_equal(toks, np.argmax(logits.numpy(np.float32, (M, N)), axis=-1)), it


# ---------------------------------------------------------------- norms / elementwise
@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("T,H", [(1, 4096), (7, 2048), (33, 3584)])
def test_rms_norm(dt, T, H):
    r = rng(T + H)
    x, w = rand_dt(r, (T, H), dt), orc.to_dt((1 + 0.02 * r.standard_normal(H)).astype(np.float32), dt)
    got = ops.rms_norm(ops.dev(x), ops.dev(w), T, H, 1e-5, dt).numpy(np.uint16, (T, H))
    assert_close_dt(got, orc.rms_norm(x, w, 1e-5, dt), dt, name="rms_norm")
    res = rand_dt(r, (T, H), dt)
    h, o = ops.add_rms_norm(ops.dev(x), ops.dev(res), ops.dev(w), T, H, 1e-5, dt)
    href = orc.add(x, res, dt)
    assert np.array_equal(h.numpy(np.uint16, (T, H)), href)
    assert_close_dt(o.numpy(np.uint16, (T, H)), orc.rms_norm(href, w, 1e-5, dt), dt, name="add_rms_norm")


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("T,Hq,Hkv,D,full", [(1, 32, 8, 128, False), (7, 4, 2, 64, False), (33, 8, 8, 128, False), (5, 16, 2, 96, False),
                                             (1, 32, 8, 128, True), (9, 4, 1, 64, True)])
def test_qk_rms_norm(dt, T, Hq, Hkv, D, full):
    """q_norm / k_norm of Attention::forward_ext (attention.rs:713-735), in place: per head over head_dim (weights [head_dim]) and
    over the whole row (weights [heads * head_dim]); the oracle = its RMSNorm on the reshaped rows (pinned against HuggingFace's
    Qwen3 by tests/golden/hf_qwen3_tiny.npz)"""
    r = rng(T + Hq + D + full)
    q, k = rand_dt(r, (T, Hq, D), dt, 2.0), rand_dt(r, (T, Hkv, D), dt, 2.0)
    nq, nk = (Hq * D, Hkv * D) if full else (D, D)
    wq = orc.to_dt((1 + 0.2 * r.standard_normal(nq)).astype(np.float32), dt)
    wk = orc.to_dt((1 + 0.2 * r.standard_normal(nk)).astype(np.float32), dt)
    dq, dk = ops.dev(q), ops.dev(k)
    ops.qk_rms_norm(dq, dk, ops.dev(wq), ops.dev(wk), T, Hq, Hkv, D, full, 1e-6, dt)
    if full:
        rq, rk = orc.rms_norm(q.reshape(T, Hq * D), wq, 1e-6, dt), orc.rms_norm(k.reshape(T, Hkv * D), wk, 1e-6, dt)
    else:
        rq, rk = orc.rms_norm(q.reshape(T * Hq, D), wq, 1e-6, dt), orc.rms_norm(k.reshape(T * Hkv, D), wk, 1e-6, dt)
    assert_close_dt(dq.numpy(np.uint16, (T, Hq, D)).reshape(rq.shape), rq, dt, name="q_norm")
    assert_close_dt(dk.numpy(np.uint16, (T, Hkv, D)).reshape(rk.shape), rk, dt, name="k_norm")


@pytest.mark.parametrize("dt", [BF16, F16])
def test_elementwise(dt):
    r = rng(11)
    n = 8 * 1000 + 5
    a, b = rand_dt(r, (n,), dt, 3.0), rand_dt(r, (n,), dt, 3.0)
    assert np.array_equal(ops.add(ops.dev(a), ops.dev(b), n, dt).numpy(np.uint16, (n,)), orc.add(a, b, dt))
    got = ops.silu_mul(ops.dev(a), ops.dev(b), n, dt).numpy(np.uint16, (n,))
    assert_close_dt(got, orc.silu_mul(a, b, dt), dt, name="silu_mul", max_mismatch_frac=0.001)


def test_embedding_and_argmax():
    r = rng(3)
    V, H, T = 1000, 256, 9
    table = rand_dt(r, (V, H), BF16)
    ids = r.integers(0, V, size=T).astype(np.uint32)
    got = ops.embedding(ops.dev(ids), ops.dev(table), T, H, V).numpy(np.uint16, (T, H))
    assert np.array_equal(got, table[ids])
    logits = r.standard_normal((5, 128256)).astype(np.float32)
    logits[1, 77] = logits[1, 90000] = 50.0  # tie -> first index
    logits[2, :] = -3.0  # all equal -> 0
    assert np.array_equal(ops.argmax(ops.dev(logits), 5, 128256), orc.argmax_f32(logits))


def test_fills_bit_exact():
    n = 10007
    b = ops.DevBuf(n * 4)
    ops.lib().vra_fill_hash_u32(b.ptr, n, 1234, 0)
    assert np.array_equal(b.numpy(np.uint32, (n,)), orc.fill_hash_u32(n, 1234))
    ops.lib().vra_fill_uniform(b.ptr, n, 5, 0.002, 0.02, BF16, 0)
    assert np.array_equal(b.numpy(np.uint16, (n,)), orc.fill_uniform((n,), 5, 0.002, 0.02, BF16))
    ops.lib().vra_fill_normal(b.ptr, n, 6, 1.0, 0.02, BF16, 0)
    assert np.array_equal(b.numpy(np.uint16, (n,)), orc.fill_normal((n,), 6, 1.0, 0.02, BF16))


# ---------------------------------------------------------------- rotary
@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("table_f32", [False, True])
@pytest.mark.parametrize("D,rot", [(128, 128), (64, 64), (128, 64)])
def test_fused_rope(interleaved, table_f32, D, rot):
    r = rng(D + rot)
    T, Hq, Hkv = 13, 8, 2
    cos, sin = orc.rope_tables(rot, 500000.0, 512, 2, 8.0, 1.0, 4.0, 8192)
    tdt = F32 if table_f32 else BF16
    cos_t, sin_t = (cos, sin) if table_f32 else (orc.to_bf16(cos), orc.to_bf16(sin))
    q, k = rand_dt(r, (T, Hq, D), BF16), rand_dt(r, (T, Hkv, D), BF16)
    pos = r.integers(0, 512, size=T).astype(np.int64)
    qd, kd = ops.dev(q), ops.dev(k)
    ops.FusedRope.apply_inplace(qd, kd, ops.dev(cos_t), ops.dev(sin_t), ops.dev(pos), interleaved, T, Hq, Hkv, D, BF16, tdt, rot)
    qr = orc.rope(q, cos_t, sin_t, pos, interleaved, BF16, tdt, rot)
    kr = orc.rope(k, cos_t, sin_t, pos, interleaved, BF16, tdt, rot)
    assert np.array_equal(qd.numpy(np.uint16, q.shape), qr)
    assert np.array_equal(kd.numpy(np.uint16, k.shape), kr)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("Hq,Hkv,D,T", [(32, 8, 128, 37), (14, 2, 64, 37), (32, 8, 128, 1100), (8, 2, 64, 1029)])
def test_rope_cache_prefill_equals_the_two_launches(dt, fp8, Hq, Hkv, D, T):
    """vra_rope_cache_prefill == vra_fused_rope + vra_reshape_and_cache bit for bit: rotated q, K-cache rows, V-cache columns
    (16-bit and FP8 caches), a padded token (slot -1) writes nothing.  T >= 1024: workgroups take 8 tokens and write V with one
    vector store per channel where the 8 slots are consecutive and aligned inside a block — runs of consecutive slots through
    shuffled blocks, a run that starts mid-block at an odd offset, a hole, and a ragged last group"""
    L = ops.lib()
    BS = 64
    NB = max(8, (T + BS - 1) // BS + 3)
    r = rng(D + Hq + dt + fp8 + T)
    cos, sin = orc.rope_tables(D, 500000.0, 512, 2, 8.0, 1.0, 4.0, 8192)
    cos_t, sin_t = orc.to_dt(cos, dt), orc.to_dt(sin, dt)
    q, k, v = rand_dt(r, (T, Hq, D), dt), rand_dt(r, (T, Hkv, D), dt), rand_dt(r, (T, Hkv, D), dt)
    pos = r.integers(0, 512, size=T).astype(np.int64)
    if T < 1024:
        slots = r.permutation(NB * BS)[:T].astype(np.int64)
    else:  # two "sequences": one from the start of its blocks, one that continues 13 tokens into a block
        blocks = r.permutation(NB)
        n1 = T // 2 + 3
        s1 = np.array([int(blocks[j // BS]) * BS + j % BS for j in range(n1)], np.int64)
        nb1 = (n1 + BS - 1) // BS
        s2 = np.array([int(blocks[nb1 + (13 + j) // BS]) * BS + (13 + j) % BS for j in range(T - n1)], np.int64)
        slots = np.concatenate([s1, s2])
        slots[100] = -1
    slots[5] = -1
    kvdt = 3 if fp8 else dt
    nbytes = NB * Hkv * BS * D * (1 if fp8 else 2)
    dcos, dsin, dpos, dslots = ops.dev(cos_t), ops.dev(sin_t), ops.dev(pos), ops.dev(slots)
    # two launches
    q1, k1, v1 = ops.dev(q), ops.dev(k), ops.dev(v)
    kc1, vc1 = ops.DevBuf(nbytes).fill_bytes(0x11), ops.DevBuf(nbytes).fill_bytes(0x11)
    L.vra_fused_rope(q1.ptr, k1.ptr, dcos.ptr, dsin.ptr, dpos.ptr, T, Hq, Hkv, D, D, 0, dt, dt, 0)
    L.vra_reshape_and_cache(k1.ptr, v1.ptr, kc1.ptr, vc1.ptr, dslots.ptr, T, Hkv, D, BS, dt, kvdt, 0)
    # one launch
    q2, k2, v2 = ops.dev(q), ops.dev(k), ops.dev(v)
    kc2, vc2 = ops.DevBuf(nbytes).fill_bytes(0x11), ops.DevBuf(nbytes).fill_bytes(0x11)
    L.vra_rope_cache_prefill(q2.ptr, k2.ptr, v2.ptr, kc2.ptr, vc2.ptr, dcos.ptr, dsin.ptr, dpos.ptr, dslots.ptr, T, Hq, Hkv, D, BS, dt, kvdt, 0)
    ops.check_error()
    assert np.array_equal(q1.numpy(np.uint16, q.shape), q2.numpy(np.uint16, q.shape))
    assert np.array_equal(kc1.numpy(np.uint8, (nbytes,)), kc2.numpy(np.uint8, (nbytes,)))
    assert np.array_equal(vc1.numpy(np.uint8, (nbytes,)), vc2.numpy(np.uint8, (nbytes,)))
    assert np.array_equal(k2.numpy(np.uint16, k.shape), k), "k itself is left un-rotated"


# ---------------------------------------------------------------- paged KV + attention
def _paged_setup(r, ctxs, Hkv, D, BS, dt, NB=64):
    """random K/V history scattered into a shuffled block pool through reshape_and_cache."""
    B = len(ctxs)
    max_blocks = max((c + BS - 1) // BS for c in ctxs)
    perm = r.permutation(NB)
    bt = np.zeros((B, max_blocks), np.uint32)
    nxt = 0
    ks, vs, slots = [], [], []
    for b, c in enumerate(ctxs):
        nb = (c + BS - 1) // BS
        bt[b, :nb] = perm[nxt:nxt + nb]
        nxt += nb
        ks.append(rand_dt(r, (c, Hkv, D), dt))
        vs.append(rand_dt(r, (c, Hkv, D), dt))
        slots.append(np.array([bt[b, j // BS] * BS + j % BS for j in range(c)], np.int64))
    k, v, sl = np.concatenate(ks), np.concatenate(vs), np.concatenate(slots)
    kc_ref = np.zeros((NB, Hkv, BS, D), np.uint16)
    vc_ref = np.zeros((NB, Hkv, D, BS), np.uint16)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, sl, BS, dt)
    # poison unused slots on the device with NaN bit patterns: kernels must never let them through
    kc, vc = ops.DevBuf(kc_ref.nbytes).fill_bytes(0xFF), ops.DevBuf(vc_ref.nbytes).fill_bytes(0xFF)
    return dict(B=B, bt=bt, max_blocks=max_blocks, k=k, v=v, slots=sl, kc_ref=kc_ref, vc_ref=vc_ref, kc=kc, vc=vc, ks=ks, vs=vs)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (28, 4, 128), (8, 1, 128), (32, 4, 64)])
@pytest.mark.parametrize("ctxs", [[1], [31, 32, 33], [64, 65, 200, 7], [1000]])
def test_paged_attention_decode(Hq, Hkv, D, ctxs):
    BS, dt = 64, BF16
    r = rng(Hq + D + sum(ctxs))
    s = _paged_setup(r, ctxs, Hkv, D, BS, dt)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt)
    pa.reshape_and_cache(ops.dev(s["k"]), ops.dev(s["v"]), s["kc"], s["vc"], ops.dev(s["slots"]), len(s["slots"]))
    # the scatter itself is byte work: bit-exact on every written slot
    kc_got = s["kc"].numpy(np.uint16, s["kc_ref"].shape)
    vc_got = s["vc"].numpy(np.uint16, s["vc_ref"].shape)
    written = np.zeros(s["kc_ref"].shape[0] * BS, bool)
    written[s["slots"]] = True
    wk = written.reshape(-1, 1, BS, 1)
    assert np.array_equal(np.where(wk, kc_got, 0), s["kc_ref"])
    assert np.array_equal(np.where(written.reshape(-1, 1, 1, BS), vc_got, 0), s["vc_ref"])
    B = s["B"]
    q = rand_dt(r, (B, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    out = pa.forward_decode(ops.dev(q), s["kc"], s["vc"], ops.dev(s["bt"]), ops.dev(cl), B, s["max_blocks"], max(ctxs))
    got = out.numpy(np.uint16, (B, Hq, D))
    ref = orc.paged_attention(q, s["kc_ref"], s["vc_ref"], s["bt"], cl, None, Hkv, BS, D ** -0.5, dt)
    # P is rounded to bf16 before P·V (MFMA operand): tolerance 2 ulp + small absolute floor
    assert_close_dt(got, ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name="decode attention", abs_floor=3e-3)


def test_paged_attention_decode_split_kv():
    Hq, Hkv, D, BS, dt = 32, 8, 128, 64, BF16
    ctxs = [5000, 4097]
    r = rng(77)
    s = _paged_setup(r, ctxs, Hkv, D, BS, dt, NB=160)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt)
    pa.reshape_and_cache(ops.dev(s["k"]), ops.dev(s["v"]), s["kc"], s["vc"], ops.dev(s["slots"]), len(s["slots"]))
    q = rand_dt(r, (2, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    ws = ops.DevBuf(ops.lib().vra_paged_attention_decode_workspace_bytes(2, Hq, D, 5000))
    out = pa.forward_decode(ops.dev(q), s["kc"], s["vc"], ops.dev(s["bt"]), ops.dev(cl), 2, s["max_blocks"], 5000, ws)
    ref = orc.paged_attention(q, s["kc_ref"], s["vc_ref"], s["bt"], cl, None, Hkv, BS, D ** -0.5, dt)
    assert_close_dt(out.numpy(np.uint16, (2, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name="split-kv", abs_floor=3e-3)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (4, 2, 64), (4, 4, 64)])
@pytest.mark.parametrize("ctxs,ws", [([1], False), ([31, 32, 33, 0], False), ([64, 65, 200, 7], False), ([3000, 2049], True),
                                     # round 5: the split boundaries of the small-batch rule (a split keeps >= 8 tiles: 2 splits from 512
                                     # tokens, 4 from 1024) and contexts beyond block 63 (the latency form re-bases its block-id vector)
                                     ([255, 256, 257, 300], True), ([511, 512, 513, 1030], True), ([5000, 2100], True)])
def test_fused_rope_cache_attention_decode(Hq, Hkv, D, ctxs, ws, dt):
    """one launch == FusedRope + reshape_and_cache + PagedAttention decode (attention.rs:745-820): the history is in the
    cache, the NEW token's q,k,v come in un-rotated; ctx 0 = padded graph lane (slot -1: writes nothing, output 0)."""
    BS, NB = 64, 128
    r = rng(Hq * 7 + D + sum(ctxs) + dt)
    B = len(ctxs)
    mb = max((c + BS - 1) // BS for c in ctxs)
    perm = r.permutation(NB)
    bt = np.zeros((B, mb), np.uint32)
    nxt = 0
    hk, hv, hs = [], [], []
    for b, c in enumerate(ctxs):
        nb = (c + BS - 1) // BS
        bt[b, :nb] = perm[nxt:nxt + nb]
        nxt += nb
        if c > 1:
            hk.append(rand_dt(r, (c - 1, Hkv, D), dt))
            hv.append(rand_dt(r, (c - 1, Hkv, D), dt))
            hs.append(np.array([int(bt[b, j // BS]) * BS + j % BS for j in range(c - 1)], np.int64))
    kc_ref = np.zeros((NB, Hkv, BS, D), np.uint16)
    vc_ref = np.zeros((NB, Hkv, D, BS), np.uint16)
    kc, vc = ops.DevBuf(kc_ref.nbytes).fill_bytes(0xFF), ops.DevBuf(vc_ref.nbytes).fill_bytes(0xFF)   # NaN poison
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt)
    if hk:
        hk, hv, hs = np.concatenate(hk), np.concatenate(hv), np.concatenate(hs)
        orc.reshape_and_cache(hk, hv, kc_ref, vc_ref, hs, BS, dt)
        pa.reshape_and_cache(ops.dev(hk), ops.dev(hv), kc, vc, ops.dev(hs), len(hs))
    q, k, v = rand_dt(r, (B, Hq, D), dt), rand_dt(r, (B, Hkv, D), dt), rand_dt(r, (B, Hkv, D), dt)
    pos = np.array([max(c - 1, 0) for c in ctxs], np.int64)
    slots = np.array([int(bt[b, (c - 1) // BS]) * BS + (c - 1) % BS if c > 0 else -1 for b, c in enumerate(ctxs)], np.int64)
    cos, sin = orc.rope_tables(D, 10000.0, 8192)
    cos, sin = orc.to_dt(cos, dt), orc.to_dt(sin, dt)
    cl = np.array(ctxs, np.uint32)
    # ---- oracle: the three-step composition on host copies of the caches
    qr = orc.rope(q, cos, sin, pos, False, dt, dt)
    kr = orc.rope(k, cos, sin, pos, False, dt, dt)
    orc.reshape_and_cache(kr, v, kc_ref, vc_ref, slots, BS, dt)
    ref = orc.paged_attention(qr, kc_ref, vc_ref, bt, cl, None, Hkv, BS, D ** -0.5, dt)
    wsb = ops.DevBuf(ops.lib().vra_paged_attention_decode_workspace_bytes(B, Hq, D, max(ctxs))) if ws else None
    dq, dk = ops.dev(q), ops.dev(k)
    out = pa.rope_cache_decode(dq, dk, ops.dev(v), kc, vc, ops.dev(cos), ops.dev(sin), ops.dev(pos), ops.dev(slots),
                               ops.dev(bt), ops.dev(cl), B, mb, max(ctxs), wsb)
    got = out.numpy(np.uint16, (B, Hq, D))
    for b, c in enumerate(ctxs):
        if c == 0:
            assert not got[b].any(), "padded lane must produce zeros"
    live = [b for b, c in enumerate(ctxs) if c > 0]
    assert_close_dt(got[live], ref[live], dt, max_ulp=2.0, max_mismatch_frac=0.5, name="fused decode", abs_floor=3e-3 if dt == BF16 else 5e-4)
    # the cache rows of the new token are the rotated k / the v, bit-exact; q and k inputs are untouched
    kc_got, vc_got = kc.numpy(np.uint16, kc_ref.shape), vc.numpy(np.uint16, vc_ref.shape)
    for b in live:
        blk, off = int(slots[b]) // BS, int(slots[b]) % BS
        assert np.array_equal(kc_got[blk, :, off, :], kr[b]) and np.array_equal(vc_got[blk, :, :, off], v[b])
    assert np.array_equal(dq.numpy(np.uint16, q.shape), q) and np.array_equal(dk.numpy(np.uint16, k.shape), k)


@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (4, 4, 64)])
@pytest.mark.parametrize("paged", [True, False])
def test_attention_prefill(Hq, Hkv, D, paged):
    """causal varlen prefill; paged mode includes a cached prefix (chunked prefill / prefix-cache hit)."""
    BS, dt = 64, BF16
    r = rng(Hq * 3 + D + paged)
    lens_q = [5, 70, 1, 33]
    prefix = [0, 64, 130, 0] if paged else [0, 0, 0, 0]
    ctxs = [a + b for a, b in zip(lens_q, prefix)]
    s = _paged_setup(r, ctxs, Hkv, D, BS, dt)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt)
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.uint32)
    cu_k = np.concatenate([[0], np.cumsum(ctxs)]).astype(np.uint32)
    Tq = int(cu_q[-1])
    q = rand_dt(r, (Tq, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    if paged:
        pa.reshape_and_cache(ops.dev(s["k"]), ops.dev(s["v"]), s["kc"], s["vc"], ops.dev(s["slots"]), len(s["slots"]))
        out = pa.forward_prefill(ops.dev(q), Tq, max(lens_q), ops.dev(cu_q), len(lens_q), k_cache=s["kc"], v_cache=s["vc"],
                                 block_tables=ops.dev(s["bt"]), context_lens=ops.dev(cl), max_blocks=s["max_blocks"])
        ref = orc.paged_attention(q, s["kc_ref"], s["vc_ref"], s["bt"], cl, cu_q, Hkv, BS, D ** -0.5, dt)
    else:
        out = pa.forward_prefill(ops.dev(q), Tq, max(lens_q), ops.dev(cu_q), len(lens_q), k=ops.dev(s["k"]), v=ops.dev(s["v"]), cu_k=ops.dev(cu_k))
        ref = orc.varlen_attention(q, s["k"], s["v"], cu_q, cu_k, D ** -0.5, dt)
    assert_close_dt(out.numpy(np.uint16, (Tq, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name="prefill attention", abs_floor=3e-3)


@pytest.mark.parametrize("W", [1, 32, 100, 4096])
@pytest.mark.parametrize("ctxs,ws", [([1, 31, 33, 200], False), ([1000, 77], False), ([3000, 2049], True)])
def test_paged_attention_decode_sliding_window(W, ctxs, ws):
    """PagedAttention::new(.., sliding_window, ..) (attention.rs:607-616): the query at position ctx-1 attends the last W keys; tiles
    in front of the window are skipped (and may hold anything: they are poisoned here), a window wider than the context is a no-op"""
    Hq, Hkv, D, BS, dt = 32, 8, 128, 64, BF16
    r = rng(W + sum(ctxs))
    s = _paged_setup(r, ctxs, Hkv, D, BS, dt, NB=128)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt, sliding_window=W)
    pa.reshape_and_cache(ops.dev(s["k"]), ops.dev(s["v"]), s["kc"], s["vc"], ops.dev(s["slots"]), len(s["slots"]))
    B = s["B"]
    q = rand_dt(r, (B, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    wsb = ops.DevBuf(ops.lib().vra_paged_attention_decode_workspace_bytes(B, Hq, D, max(ctxs))) if ws else None
    out = pa.forward_decode(ops.dev(q), s["kc"], s["vc"], ops.dev(s["bt"]), ops.dev(cl), B, s["max_blocks"], max(ctxs), wsb)
    ref = orc.paged_attention(q, s["kc_ref"], s["vc_ref"], s["bt"], cl, None, Hkv, BS, D ** -0.5, dt, sliding_window=W)
    assert_close_dt(out.numpy(np.uint16, (B, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name=f"decode attention, window {W}", abs_floor=3e-3)
    if W >= max(ctxs):  # no-op window: the very same bits as the plain entry point
        plain = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt).forward_decode(ops.dev(q), s["kc"], s["vc"], ops.dev(s["bt"]), ops.dev(cl), B,
                                                                                 s["max_blocks"], max(ctxs), wsb)
        assert np.array_equal(out.numpy(np.uint16, (B, Hq, D)), plain.numpy(np.uint16, (B, Hq, D)))


@pytest.mark.parametrize("W", [1, 16, 50, 300])
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (4, 4, 64)])
def test_attention_prefill_sliding_window(W, Hq, Hkv, D):
    """causal varlen prefill over the paged cache (with cached prefixes) under a sliding window: query at position p sees p-W+1 .. p"""
    BS, dt = 64, BF16
    r = rng(Hq * 3 + D + W)
    lens_q = [5, 70, 1, 133]
    prefix = [0, 64, 130, 40]
    ctxs = [a + b for a, b in zip(lens_q, prefix)]
    s = _paged_setup(r, ctxs, Hkv, D, BS, dt)
    pa = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, dt, sliding_window=W)
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.uint32)
    Tq = int(cu_q[-1])
    q = rand_dt(r, (Tq, Hq, D), dt)
    cl = np.array(ctxs, np.uint32)
    pa.reshape_and_cache(ops.dev(s["k"]), ops.dev(s["v"]), s["kc"], s["vc"], ops.dev(s["slots"]), len(s["slots"]))
    out = pa.forward_prefill(ops.dev(q), Tq, max(lens_q), ops.dev(cu_q), len(lens_q), k_cache=s["kc"], v_cache=s["vc"],
                             block_tables=ops.dev(s["bt"]), context_lens=ops.dev(cl), max_blocks=s["max_blocks"])
    ref = orc.paged_attention(q, s["kc_ref"], s["vc_ref"], s["bt"], cl, cu_q, Hkv, BS, D ** -0.5, dt, sliding_window=W)
    assert_close_dt(out.numpy(np.uint16, (Tq, Hq, D)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.5, name=f"prefill attention, window {W}", abs_floor=3e-3)


def test_causal_mask_and_cast():
    L = 37
    m = ops.DevBuf(L * L * 2)
    ops.lib().vra_causal_mask(m.ptr, L, 0, BF16, 0)
    assert np.array_equal(m.numpy(np.uint16, (L, L)), orc.causal_mask(L, 0, BF16))
    ops.lib().vra_causal_mask(m.ptr, L, 8, BF16, 0)
    assert np.array_equal(m.numpy(np.uint16, (L, L)), orc.causal_mask(L, 8, BF16))
    r = rng(1)
    x = r.standard_normal(1001).astype(np.float32)
    o = ops.DevBuf(1001 * 2)
    ops.lib().vra_cast(ops.dev(x).ptr, o.ptr, 1001, F32, BF16, 0)
    assert np.array_equal(o.numpy(np.uint16, (1001,)), orc.to_bf16(x))
    ops.lib().vra_cast(ops.dev(x).ptr, o.ptr, 1001, F32, F16, 0)
    assert np.array_equal(o.numpy(np.uint16, (1001,)), x.astype(np.float16).view(np.uint16))


def test_swap_blocks_roundtrip():
    r = rng(2)
    blocks = r.integers(0, 2 ** 32, size=(8, 1024), dtype=np.uint64).astype(np.uint32)
    src, dst = ops.dev(blocks), ops.DevBuf(blocks.nbytes).zero()
    pairs = np.array([0, 3, 5, 1, 7, 7], np.int64)
    ops.lib().vra_swap_blocks(src.ptr, dst.ptr, pairs.ctypes.data_as(C.c_void_p), 3, 4096, 0, 0)
    got = dst.numpy(np.uint32, (8, 1024))
    assert np.array_equal(got[3], blocks[0]) and np.array_equal(got[1], blocks[5]) and np.array_equal(got[7], blocks[7])
    assert not got[0].any()


def test_argument_errors_are_reported():
    L = ops.lib()
    L.vra_clear_error()
    L.gptq_repack(None, None, 16, 16, 0)
    assert b"null" in L.vra_last_error()
    L.vra_clear_error()
    d = ops.DevBuf(1024)
    L.gptq_repack(d.ptr, d.ptr, 3, 16, 0)  # K = 24, not a multiple of 128
    assert L.vra_last_error() != b""
    L.vra_clear_error()


def test_rccl_all_reduce_single_rank():
    """ncclGetUniqueId -> ncclCommInitRank -> ncclAllReduce at world size 1: a REAL one-rank RCCL communicator (no dummy
    object), all-reduce(sum) in the storage dtype, in place and out of place (identity at one rank).  The two-rank product
    path — launcher, shard loading, Model::forward's world > 1 branch, the one-shot all-reduce (and RCCL when the box has two
    GPUs) — is tests/test_gpu_tp.py."""
    L = ops.lib()
    uid = (C.c_uint8 * 128)()
    assert L.vra_comm_unique_id(uid) == 0
    comm = L.vra_comm_create(uid, 0, 1, 0)
    assert comm, "vra_comm_create failed: " + (L.vra_last_error() or b"").decode()
    try:
        assert L.vra_comm_rank(comm) == 0 and L.vra_comm_world_size(comm) == 1
        r = rng(77)
        for dt in (BF16, F16):
            x = rand_dt(r, (3, 4096), dt)
            src, dst = ops.dev(x), ops.DevBuf(x.nbytes)
            L.vra_all_reduce(comm, src.ptr, dst.ptr, x.size, dt, 0)
            ops.check_error()
            assert np.array_equal(dst.numpy(np.uint16, x.shape), x)
            L.vra_all_reduce(comm, src.ptr, src.ptr, x.size, dt, 0)
            ops.check_error()
            assert np.array_equal(src.numpy(np.uint16, x.shape), x)
    finally:
        L.vra_comm_destroy(comm)


def test_no_device_side_timeouts():
    """runs last in this file: none of the split-K exchanges above gave up waiting for a slice"""
    assert ops.lib().vra_take_device_error() == 0
