"""GPU parity tests of kernel E (csrc/gemv_q4s.cuh): the int4 GEMV of decode batches of 1..4 rows — the kernel behind the
bs = 1 headline and the one `roofline` is quoted on — through the C ABI, against the CPU oracle (exact W4A16 product, one
rounding: <= 1 storage ulp) at the REAL widths of the BASELINE configs:
  Llama-3-8B   o_proj 4096x4096, q/k/v 4096x6144, gate/up 4096x14336 (pair), down 14336x4096 (7 k-tiles per wave)
  Qwen2-7B     K = 3584 (28 k-tiles: waves 12..15 have one tile fewer), AWQ zero points
  Llama-3-70B  TP=8 rank: K = 1024 (8 k-tiles: half of the waves idle) x 8192, K = 8192 (4 tiles per wave), q/k/v 8192 x 1280 (80 units)
and across scale layouts (row-major checkpoint tensors, the Marlin-permuted scales of the reference's FFI), group sizes
(128, 256, channel-wise), dtypes, fused bias / residual / RMSNorm / SiLU*mul."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import BF16, F16, assert_close_dt, make_quant, rand_dt, rng
from vllm_rs_amd import ops

pytestmark = pytest.mark.gpu


def _tiled(q, awq=False):
    return ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 6144), (14336, 4096), (3584, 4608), (1024, 8192), (8192, 2304), (4096, 2064), (8192, 1280)])
def test_gemv_s_gptq_real_widths(M, K, N):
    r = rng(M * 7 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    out = ops.wna16_gemm(ops.dev(x), _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128)
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name=f"gemv_s M={M} K={K} N={N}", abs_floor=2e-3)


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_gemv_s_qwen2_down_projection_ten_tiles_per_wave(M, dt):
    """K = 18944 (Qwen2-7B down: 148 k-tiles, 10 per wave): four row regions per tile no longer fit the LDS, kernel E keeps M of
    them (1..3 rows; 4 rows go to kernel A as before) — AWQ zero points, residual"""
    K, N = 18944, 3584
    r = rng(M + K)
    q = make_quant(r, K, N, 128, dt, True)
    x, res = rand_dt(r, (M, K), dt), rand_dt(r, (M, N), dt)
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, True), ops.dev(q["scales"]), ops.dev(q["qzeros"]), M, K, N, 128, True, 0, None, ops.dev(res), dtype=dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, dt, None, res)
    g0 = np.abs(orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, dt), dt))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, name=f"gemv_s K=18944 M={M}", mag=g0)


@pytest.mark.parametrize("M", [1, 4])
@pytest.mark.parametrize("dt,awq,gs,layout", [(BF16, True, 128, 0), (F16, True, 128, 0), (F16, False, 128, 0), (BF16, False, 128, 1), (F16, True, 128, 1),
                                              (BF16, False, 256, 0), (BF16, True, 512, 1), (BF16, False, -1, 0), (F16, True, -1, 0)])
def test_gemv_s_formats(M, dt, awq, gs, layout):
    K, N = 3584, 4608
    r = rng(M + gs + layout * 3 + awq)
    q = make_quant(r, K, N, gs, dt, awq)
    x = rand_dt(r, (M, K), dt)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True) if layout == 1 else q["scales"]
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, awq), ops.dev(sc), ops.dev(q["qzeros"]) if awq else None, M, K, N, gs, awq, layout, dtype=dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, name=f"gemv_s formats dt={dt} awq={awq} gs={gs} layout={layout}", abs_floor=2e-3)


@pytest.mark.parametrize("M", [1, 3])
@pytest.mark.parametrize("awq", [False, True])
def test_gemv_s_marlin_ffi_symbols(M, awq):
    """through the reference's own symbols (gptq.rs:118-178): Marlin-permuted scales are read in place — no conversion
    launch, no scratch copy — and the workspace stays zero"""
    K, N, gs, dt = 4096, 4096, 128, BF16
    r = rng(M + 17 * awq)
    q = make_quant(r, K, N, gs, dt, awq)
    x = rand_dt(r, (M, K), dt)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True)
    ws = ops.DevBuf(N * 4).zero()
    out = ops.gptq_matmul(ops.dev(x), _tiled(q, awq), ops.dev(sc), ops.dev(q["qzeros"]), None, ws, 4, gs, awq, M, K, N, dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, name="marlin ffi (kernel E)", abs_floor=2e-3)
    assert not ws.numpy(np.uint32, (N,)).any()


@pytest.mark.parametrize("M", [1, 2, 4])
def test_gemv_s_bias_residual(M):
    K, N = 4096, 4096
    r = rng(M + 5)
    q = make_quant(r, K, N, 128, BF16, True)
    x, bias, res = rand_dt(r, (M, K), BF16), rand_dt(r, (N,), BF16), rand_dt(r, (M, N), BF16)
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, True), ops.dev(q["scales"]), ops.dev(q["qzeros"]), M, K, N, 128, True, 0, ops.dev(bias), ops.dev(res))
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, BF16, bias, res)
    g0 = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], 128, BF16), BF16)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, BF16)[None, :]))
    # three roundings (GEMM, + bias, + residual): each may flip by one ulp of ITS magnitude; two flips on one element are
    # rare but happen (1 of 8192 here)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, max_ulp=2.0, name="gemv_s bias+residual", mag=mag)


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("K,N,awq", [(4096, 14336, False), (3584, 18944, True), (8192, 3584, False)])
def test_gemv_s_gate_up_pair(M, K, N, awq):
    r = rng(M + K)
    qg, qu = make_quant(r, K, N, 128, BF16, awq), make_quant(r, K, N, 128, BF16, awq)
    x = rand_dt(r, (M, K), BF16)
    z = (lambda q: ops.dev(q["qzeros"])) if awq else (lambda q: None)
    out = ops.wna16_gate_up_silu(ops.dev(x), _tiled(qg, awq), ops.dev(qg["scales"]), z(qg), _tiled(qu, awq), ops.dev(qu["scales"]), z(qu), M, K, N, 128, awq)
    g = orc.wna16_gemm(x, qg["idx"], qg["zeros"], qg["scales"], 128, BF16)
    u = orc.wna16_gemm(x, qu["idx"], qu["zeros"], qu["scales"], 128, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), orc.silu_mul(g, u, BF16), BF16, max_ulp=3.0, max_mismatch_frac=0.04, name="gemv_s gate/up", abs_floor=4e-3)


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("K,N", [(4096, 6144), (8192, 1280)])
def test_gemv_s_fused_rms_norm(M, dt, K, N):
    """RMSNorm fused into kernel E (round 5: the factor rstd applied in the epilogue, no cross-wave step before the stream) against
    the oracle's deferred order, and against the two separate device calls (reference order).  8192 x 1280: the
    q/k/v launch of a Llama-3-70B TP=8 rank — 80 units, fewer than half the CUs (kernel E since round 3)"""
    r = rng(M + dt + N)
    q = make_quant(r, K, N, 128, dt, False)
    x, nw = rand_dt(r, (M, K), dt, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), dt)
    bias = rand_dt(r, (N,), dt)
    t = _tiled(q)
    out = ops.rms_norm_wna16_gemm(ops.dev(x), ops.dev(nw), 1e-5, t, ops.dev(q["scales"]), None, M, K, N, 128, bias=ops.dev(bias), dtype=dt)
    got = out.numpy(np.uint16, (M, N))
    # round 5: kernel E applies the normalisation factor in its EPILOGUE (x staged as round(x * g), rstd on the f32 dot products:
    # gemv_q4s.cuh) — the oracle's stated second order, orc.rms_norm_deferred + row_scale
    xg, rs = orc.rms_norm_deferred(x, nw, 1e-5, dt)
    ref = orc.wna16_gemm(xg, q["idx"], None, q["scales"], 128, dt, bias, row_scale=rs)
    g0 = np.abs(orc.from_dt(orc.wna16_gemm(xg, q["idx"], None, q["scales"], 128, dt, row_scale=rs), dt))
    # GEMM and + bias are two roundings: a few double flips in 25k outputs
    # (abs_floor: a flipped activation moves an output by ~|w| * ulp(x) whatever the output's own magnitude)
    assert_close_dt(got, ref, dt, max_ulp=2.0, max_mismatch_frac=0.03, name="fused norm gemv", mag=g0, abs_floor=2e-3 if dt == BF16 else 3e-4)
    # against the reference order (norm, then GEMM, as separate launches: others.rs:11-29): another rounding pattern of the same size —
    # outputs differ by single roundings of the intermediate
    sep = ops.wna16_gemm(ops.rms_norm(ops.dev(x), ops.dev(nw), M, K, 1e-5, dt), t, ops.dev(q["scales"]), None, M, K, N, 128, bias=ops.dev(bias), dtype=dt)
    sepv = sep.numpy(np.uint16, (M, N))
    frac = float((got != sepv).mean())
    print(f"[fused norm] M={M} dt={dt}: {100 * frac:.3f}% of outputs differ from rms_norm + gemm as separate launches (reference order)")
    # (measured: 25 % of the outputs differ — K = 4096 independent roundings of x̂ —, none by more than 2 storage ulps of the row's scale)
    gv, sv = orc.from_dt(got, dt), orc.from_dt(sepv, dt)
    row_ulp = 2.0 ** (np.floor(np.log2(np.abs(sv).max(axis=-1, keepdims=True))) - (7 if dt == BF16 else 10))
    assert frac < 0.6 and float((np.abs(gv - sv) / row_ulp).max()) <= 4.0


@pytest.mark.parametrize("M", [1, 3])
@pytest.mark.parametrize("case", ["tiny", "outlier"])
def test_gemv_s_fused_rms_norm_f16_range(M, case):
    """the deferred order stages round(x * g) WITHOUT rstd.  In f16 that leaves the range the reference's
    round(round(x * rstd) * g) lives in: 'tiny' = layer-0 magnitudes (|x| ~ 1e-2 times norm weights ~ 1e-2: x * g ~ 1e-4 sits in f16's
    subnormals, 2^-14 = 6.1e-5), 'outlier' = a massive-activation channel with g > 1 (x * g = 9e4 > 65504: inf without a rescale).
    Kernel E stages each wave's slices times a power of two of its own (gemv_q4s.cuh) — checked here against the REFERENCE-order
    oracle (others.rs:11-29: norm, then GEMM), not the mirrored one: finite, and within 4 ulps of the row scale like the
    well-scaled case of test_gemv_s_fused_rms_norm."""
    K, N, dt = 4096, 6144, F16
    r = rng(M + len(case))
    q = make_quant(r, K, N, 128, dt, False)
    if case == "tiny":
        x = orc.to_dt((0.01 * r.standard_normal((M, K))).astype(np.float32), dt)
        nw = orc.to_dt((0.01 * (1.0 + 0.2 * r.standard_normal(K))).astype(np.float32), dt)
    else:
        xf = r.standard_normal((M, K)).astype(np.float32)
        gf = (1.0 + 0.1 * r.standard_normal(K)).astype(np.float32)
        xf[:, 1234] = 300.0
        gf[1234] = 300.0
        x, nw = orc.to_dt(xf, dt), orc.to_dt(gf, dt)
    out = ops.rms_norm_wna16_gemm(ops.dev(x), ops.dev(nw), 1e-5, _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128, dtype=dt)
    got = orc.from_dt(out.numpy(np.uint16, (M, N)), dt)
    assert np.isfinite(got).all(), "the staged operand left f16's range"
    ref = orc.from_dt(orc.wna16_gemm(orc.rms_norm(x, nw, 1e-5, dt), q["idx"], None, q["scales"], 128, dt), dt)
    row_ulp = 2.0 ** (np.floor(np.log2(np.abs(ref).max(axis=-1, keepdims=True))) - 10)
    dev = float((np.abs(got - ref) / row_ulp).max())
    print(f"[f16 range] {case} M={M}: max {dev:.2f} ulps of the row scale against the reference order")
    assert dev <= 4.0, dev


@pytest.mark.parametrize("M", [1, 4])
def test_gemv_s_fused_rms_norm_gate_up(M):
    K, N = 4096, 14336
    r = rng(M + 31)
    qg, qu = make_quant(r, K, N, 128, BF16), make_quant(r, K, N, 128, BF16)
    x, nw = rand_dt(r, (M, K), BF16, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), BF16)
    out = ops.rms_norm_wna16_gate_up_silu(ops.dev(x), ops.dev(nw), 1e-5, _tiled(qg), ops.dev(qg["scales"]), None, _tiled(qu), ops.dev(qu["scales"]), None,
                                          M, K, N, 128)
    xg, rs = orc.rms_norm_deferred(x, nw, 1e-5, BF16)  # kernel E: rstd in the epilogue (see test_gemv_s_fused_rms_norm)
    g = orc.wna16_gemm(xg, qg["idx"], None, qg["scales"], 128, BF16, row_scale=rs)
    u = orc.wna16_gemm(xg, qu["idx"], None, qu["scales"], 128, BF16, row_scale=rs)
    assert_close_dt(out.numpy(np.uint16, (M, N)), orc.silu_mul(g, u, BF16), BF16, max_ulp=3.0, max_mismatch_frac=0.05, name="fused norm gate/up", abs_floor=4e-3)


def test_gemv_s_repeated_launches_are_bitwise_stable():
    """fixed summation order (no atomics): the same launch twice gives the same bits, also with the tail prefetch active"""
    K, N, M = 4096, 4096, 2
    r = rng(3)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    t, sc, xd = _tiled(q), ops.dev(q["scales"]), ops.dev(x)
    a = ops.wna16_gemm(xd, t, sc, None, M, K, N, 128).numpy(np.uint16, (M, N))
    for _ in range(5):
        assert np.array_equal(a, ops.wna16_gemm(xd, t, sc, None, M, K, N, 128).numpy(np.uint16, (M, N)))


# ---------------------------------------------------------------------------------------------
# kernel W (csrc/gemv_q4w.cuh): 5..32 rows, K <= 4096 — x fragments in registers, weights dequantised once for all rows
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [5, 8, 16, 17, 31, 32])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 6144), (3584, 4608), (1024, 8192), (2048, 2064)])
def test_gemv_w_gptq_real_widths(M, K, N):
    r = rng(M * 11 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    out = ops.wna16_gemm(ops.dev(x), _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128)
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name=f"gemv_w M={M} K={K} N={N}", abs_floor=2e-3)


@pytest.mark.parametrize("M", [7, 32])
@pytest.mark.parametrize("dt,awq,gs,layout", [(BF16, True, 128, 0), (F16, True, 128, 1), (F16, False, 128, 0), (BF16, False, 128, 1), (BF16, False, 256, 0),
                                              (BF16, True, -1, 0)])
def test_gemv_w_formats_bias_residual(M, dt, awq, gs, layout):
    K, N = 3584, 4608
    r = rng(M + gs + layout * 3 + awq + 100)
    q = make_quant(r, K, N, gs, dt, awq)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True) if layout == 1 else q["scales"]
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, awq), ops.dev(sc), ops.dev(q["qzeros"]) if awq else None, M, K, N, gs, awq, layout, ops.dev(bias), ops.dev(res),
                         dtype=dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt, bias, res)
    g0 = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt), dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, name=f"gemv_w formats dt={dt} awq={awq} gs={gs} layout={layout}", mag=mag)


@pytest.mark.parametrize("M", [6, 16, 32])
@pytest.mark.parametrize("K,N,dt", [(4096, 6144, BF16), (2048, 4096, BF16), (3584, 4608, F16), (1024, 2048, F16)])
def test_gemv_w_fused_rms_norm_qkv_shape(M, K, N, dt):
    """K < 4096: waves without a k-tile (their share of the row's sum of squares is masked after the X·Xᵀ MFMAs); K = 3584: the
    fourth tile exists for half of the waves only"""
    r = rng(M + 55 + K)
    q = make_quant(r, K, N, 128, dt, False)
    x, nw = rand_dt(r, (M, K), dt, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), dt)
    bias = rand_dt(r, (N,), dt)
    out = ops.rms_norm_wna16_gemm(ops.dev(x), ops.dev(nw), 1e-5, _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128, bias=ops.dev(bias), dtype=dt)
    xn = orc.rms_norm(x, nw, 1e-5, dt)
    ref = orc.wna16_gemm(xn, q["idx"], None, q["scales"], 128, dt, bias)
    g0 = np.abs(orc.from_dt(orc.wna16_gemm(xn, q["idx"], None, q["scales"], 128, dt), dt))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.03, name="gemv_w fused norm", mag=g0, abs_floor=2e-3)


@pytest.mark.parametrize("M", [5, 12, 16])
@pytest.mark.parametrize("awq", [False, True])
def test_gemv_w_gate_up_pair(M, awq):
    K, N = 4096, 14336
    r = rng(M + K + awq)
    qg, qu = make_quant(r, K, N, 128, BF16, awq), make_quant(r, K, N, 128, BF16, awq)
    x, nw = rand_dt(r, (M, K), BF16, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), BF16)
    z = (lambda q: ops.dev(q["qzeros"])) if awq else (lambda q: None)
    out = ops.rms_norm_wna16_gate_up_silu(ops.dev(x), ops.dev(nw), 1e-5, _tiled(qg, awq), ops.dev(qg["scales"]), z(qg), _tiled(qu, awq), ops.dev(qu["scales"]), z(qu),
                                          M, K, N, 128, awq)
    xn = orc.rms_norm(x, nw, 1e-5, BF16)
    g = orc.wna16_gemm(xn, qg["idx"], qg["zeros"], qg["scales"], 128, BF16)
    u = orc.wna16_gemm(xn, qu["idx"], qu["zeros"], qu["scales"], 128, BF16)
    # gate and up each carry a possible 1-ulp flip (f32 vs f64 accumulation, one flipped normalised activation): their product moves
    # by up to the sum of both relative errors, silu's slope adds a little: 4 ulps over 230k outputs
    assert_close_dt(out.numpy(np.uint16, (M, N)), orc.silu_mul(g, u, BF16), BF16, max_ulp=4.0, max_mismatch_frac=0.05, name="gemv_w gate/up", abs_floor=8e-3)


# ---------------------------------------------------------------------------------------------
# kernel W in row blocks (round 4): short prefills of 33..256 rows, K <= 4096 — blockIdx.y = block of 32 rows, every workgroup an
# independent 32-row launch of the decode kernel over a contiguous run of units (parity cases at M in {64, 128, 200, 256})
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [33, 64, 128, 200, 256])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 6144), (3584, 4608)])
def test_gemv_w_row_blocks_gptq_real_widths(M, K, N):
    r = rng(M * 13 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    out = ops.wna16_gemm(ops.dev(x), _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128)
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name=f"gemv_w row blocks M={M} K={K} N={N}", abs_floor=2e-3)


@pytest.mark.parametrize("M", [64, 200])
@pytest.mark.parametrize("dt,awq,gs,layout", [(BF16, True, 128, 0), (F16, False, 128, 1), (BF16, False, 256, 0)])
def test_gemv_w_row_blocks_formats_bias_residual_in_place(M, dt, awq, gs, layout):
    """o_proj as the engine calls it: the residual is the output buffer (h += o_proj(attn)), every workgroup reads its own rows x
    columns of it before the stream and writes them behind it"""
    K, N = 4096, 4096
    r = rng(M + gs + layout * 3 + awq + 200)
    q = make_quant(r, K, N, gs, dt, awq)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True) if layout == 1 else q["scales"]
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, awq), ops.dev(sc), ops.dev(q["qzeros"]) if awq else None, M, K, N, gs, awq, layout, ops.dev(bias), ops.dev(res),
                         dtype=dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt, bias, res)
    g0 = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt), dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, name=f"gemv_w row blocks formats dt={dt} awq={awq} gs={gs} layout={layout}", mag=mag)


@pytest.mark.parametrize("M", [64, 128, 200, 256])
@pytest.mark.parametrize("K,N,dt", [(4096, 6144, BF16), (3584, 4608, F16)])
def test_gemv_w_row_blocks_fused_rms_norm_qkv_shape(M, K, N, dt):
    r = rng(M + 77 + K)
    q = make_quant(r, K, N, 128, dt, False)
    x, nw = rand_dt(r, (M, K), dt, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), dt)
    bias = rand_dt(r, (N,), dt)
    out = ops.rms_norm_wna16_gemm(ops.dev(x), ops.dev(nw), 1e-5, _tiled(q), ops.dev(q["scales"]), None, M, K, N, 128, bias=ops.dev(bias), dtype=dt)
    xn = orc.rms_norm(x, nw, 1e-5, dt)
    ref = orc.wna16_gemm(xn, q["idx"], None, q["scales"], 128, dt, bias)
    g0 = np.abs(orc.from_dt(orc.wna16_gemm(xn, q["idx"], None, q["scales"], 128, dt), dt))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, max_mismatch_frac=0.03, name="gemv_w row blocks fused norm", mag=g0, abs_floor=2e-3)


def test_gemv_w_row_blocks_rows_are_independent_of_the_batch_they_ride_in():
    """row m of a 200-row launch == row m of a 32-row launch of the same rows (same kernel, same summation order): bitwise"""
    K, N = 4096, 4096
    r = rng(4242)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (200, K), BF16)
    big = ops.wna16_gemm(ops.dev(x), _tiled(q), ops.dev(q["scales"]), None, 200, K, N, 128).numpy(np.uint16, (200, N))
    for lo in (0, 96, 192):
        hi = min(lo + 32, 200)
        small = ops.wna16_gemm(ops.dev(x[lo:hi]), _tiled(q), ops.dev(q["scales"]), None, hi - lo, K, N, 128).numpy(np.uint16, (hi - lo, N))
        assert np.array_equal(big[lo:hi], small), f"rows {lo}..{hi}"


@pytest.mark.parametrize("M", [17, 32, 33, 64, 100, 128])
@pytest.mark.parametrize("awq", [False, True])
def test_gemv_w_row_blocks_gate_up_pair(M, awq):
    """norm + gate + up + SiLU*mul of 17..128 rows in the SEQUENTIAL pair form of kernel W (PSEQ, gemv_q4w.cuh): the units of a
    workgroup alternate gate / up blocks of the single-stream two-m-tile kernel, the up unit's epilogue applies SiLU(gate) * up.
    Decode batches of 17..32 rows: one launch instead of a norm launch + kernel C; a 128-token prefill: 4 row blocks x 64 column
    groups of 14 pairs instead of norm + kernel D."""
    K, N = 4096, 14336
    r = rng(M * 3 + K + awq)
    qg, qu = make_quant(r, K, N, 128, BF16, awq), make_quant(r, K, N, 128, BF16, awq)
    x, nw = rand_dt(r, (M, K), BF16, 2.0), orc.to_dt((1.0 + 0.1 * r.standard_normal(K)).astype(np.float32), BF16)
    z = (lambda q: ops.dev(q["qzeros"])) if awq else (lambda q: None)
    out = ops.rms_norm_wna16_gate_up_silu(ops.dev(x), ops.dev(nw), 1e-5, _tiled(qg, awq), ops.dev(qg["scales"]), z(qg), _tiled(qu, awq), ops.dev(qu["scales"]), z(qu),
                                          M, K, N, 128, awq)
    xn = orc.rms_norm(x, nw, 1e-5, BF16)
    g = orc.wna16_gemm(xn, qg["idx"], qg["zeros"], qg["scales"], 128, BF16)
    u = orc.wna16_gemm(xn, qu["idx"], qu["zeros"], qu["scales"], 128, BF16)
    # (as test_gemv_w_gate_up_pair, over up to 3.7 M outputs instead of 230 k: gate and up each carry a possible 1-ulp flip and SiLU's
    # slope stretches the gate's — one output in 1.4 M reached 4.x ulp at 100 rows: 6)
    assert_close_dt(out.numpy(np.uint16, (M, N)), orc.silu_mul(g, u, BF16), BF16, max_ulp=6.0, max_mismatch_frac=0.05, name="gemv_w row blocks gate/up", abs_floor=8e-3)


# ---------------------------------------------------------------------------------------------
# kernel W, K > 4096 (down_proj): K slices across workgroups — the slices' f32 partial tiles meet through memory and the last slice
# of a unit group sums them in slice order.  Slice shapes: 14336 = 4 x 28 tiles (waves 4..7 hold three), 18944 = 4 x 30 + 28
# (uneven last slice, 51 unit groups, 4 or 5 units each), 11008 = 2 x 29 + 28, 8192 = 2 x 32, 5632 = 2 x 22 (two waves hold two tiles)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [5, 16, 17, 32])
@pytest.mark.parametrize("K,N", [(14336, 4096), (18944, 3584), (11008, 4096), (8192, 8192), (5632, 2048)])
def test_gemv_w_k_slices_gptq_real_widths(M, K, N):
    r = rng(M * 13 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    tiled, sc = _tiled(q), ops.dev(q["scales"])
    out = ops.wna16_gemm(ops.dev(x), tiled, sc, None, M, K, N, 128)
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    got = out.numpy(np.uint16, (M, N))
    assert_close_dt(got, ref, BF16, name=f"gemv_w k-slices M={M} K={K} N={N}", abs_floor=4e-3)
    for _ in range(3):  # fixed slice order: bitwise stable from launch to launch, flags back at zero
        again = ops.wna16_gemm(ops.dev(x), tiled, sc, None, M, K, N, 128).numpy(np.uint16, (M, N))
        assert np.array_equal(got, again)
    assert ops.lib().vra_take_device_error() == 0


@pytest.mark.parametrize("M", [33, 64, 96, 150])
@pytest.mark.parametrize("K,N", [(14336, 4096), (18944, 3584), (8192, 8192)])
def test_gemv_w_k_slices_in_row_blocks(M, K, N):
    """down_proj of short prefills (33..96 and 129..160 rows, round 4): the K-sliced launch in row blocks of 32 rows (blockIdx.y),
    every row block with its own slabs and flags; a row of such a launch is bit-identical to the same row of a 32-row launch"""
    r = rng(M * 17 + K + N)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    tiled, sc = _tiled(q), ops.dev(q["scales"])
    got = ops.wna16_gemm(ops.dev(x), tiled, sc, None, M, K, N, 128).numpy(np.uint16, (M, N))
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16)
    assert_close_dt(got, ref, BF16, name=f"gemv_w k-slices in row blocks M={M} K={K} N={N}", abs_floor=4e-3)
    for _ in range(2):  # fixed slice order per row block: bitwise stable, flags back at zero
        assert np.array_equal(got, ops.wna16_gemm(ops.dev(x), tiled, sc, None, M, K, N, 128).numpy(np.uint16, (M, N)))
    r0 = ((M - 1) // 32) * 32  # the last (possibly ragged) row block as a launch of its own
    one = ops.wna16_gemm(ops.dev(np.ascontiguousarray(x[r0:])), tiled, sc, None, M - r0, K, N, 128).numpy(np.uint16, (M - r0, N))
    if M - r0 >= 17:  # (same m-tile count: 17..32 rows run the two-m-tile kernel as the row blocks do)
        assert np.array_equal(got[r0:], one)
    assert ops.lib().vra_take_device_error() == 0


@pytest.mark.parametrize("M", [7, 32, 64, 150])
@pytest.mark.parametrize("dt,awq,gs,layout", [(BF16, True, 128, 0), (F16, True, 128, 1), (F16, False, 128, 0), (BF16, False, 256, 0), (BF16, True, -1, 0)])
def test_gemv_w_k_slices_formats_bias_residual_in_place(M, dt, awq, gs, layout):
    K, N = 18944, 3584
    r = rng(M + gs + layout * 3 + awq + 300)
    q = make_quant(r, K, N, gs, dt, awq)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True) if layout == 1 else q["scales"]
    out = ops.wna16_gemm(ops.dev(x), _tiled(q, awq), ops.dev(sc), ops.dev(q["qzeros"]) if awq else None, M, K, N, gs, awq, layout, ops.dev(bias), ops.dev(res),
                         dtype=dt)
    ref = orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt, bias, res)
    g0 = orc.from_dt(orc.wna16_gemm(x, q["idx"], q["zeros"], q["scales"], gs, dt), dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, max_ulp=2.0, name=f"gemv_w k-slices formats dt={dt} awq={awq} gs={gs} layout={layout}", mag=mag)
    assert ops.lib().vra_take_device_error() == 0
