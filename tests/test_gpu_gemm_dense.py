"""Kernel X (vllm_rs_amd/csrc/gemm_dense.cuh): the prefill GEMM of long prompts as dequant pass + 256-row dense GEMM on Marlin-rounded
weights — against the oracle's Marlin variant (orc.dequant + orc.gemm_wdense == orc.wna16_gemm(marlin_rounded=True); the arithmetic of
the reference's Marlin kernels, src/utils/gptq.rs:116-178).  Tolerance: <= 1 storage ulp of the output (f32 accumulation order)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import BF16, F16, assert_close_dt, make_quant, rand_dt, rng
from vllm_rs_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture
def dense_from():
    """lower the row threshold of the dense prefill path for one test, restore it afterwards"""
    L = ops.lib()
    old = L.vra_debug_dense_prefill_min_rows()

    def set_rows(rows):
        L.vra_debug_set_dense_prefill_min_rows(rows)

    yield set_rows
    L.vra_debug_set_dense_prefill_min_rows(old)


def frag_to_dense(wd, K, NV):
    """fragment order (n-frag, k-chunk of 32, lane = q * 16 + r, 8 elements) -> [K, NV]"""
    a = wd.reshape(NV // 16, K // 32, 4, 16, 8)  # [nf, kc, q, r, e]: k = kc * 32 + q * 8 + e, n = nf * 16 + r
    return np.ascontiguousarray(a.transpose(1, 2, 4, 0, 3).reshape(K, NV))


def dequant_frag(q, K, N, gs, dt, awq, layout=0, wd=None, nv=None, vfrag0=0, vstride=1):
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    sc = q["scales"]
    if layout == 1:
        sc = orc.marlin_permute_scales(sc, grouped=(gs > 0 and gs < K))
    nv = nv or N
    wd = wd or ops.DevBuf(K * nv * 2).fill_bytes(0xEE)
    d_sc, d_qz = ops.dev(sc), ops.dev(q["qzeros"]) if awq else None  # (named: a temporary would be freed before the launch)
    ops.lib().vra_wna16_dequant_frag(tiled.ptr, d_sc.ptr, d_qz.ptr if awq else None, wd.ptr, K, N, gs, int(awq), layout, dt, vfrag0, vstride, 0)
    ops.check_error()
    ops.lib().vra_device_sync()
    return wd


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("awq,gs,layout", [(False, 128, 0), (False, 128, 1), (True, 128, 1), (False, 64, 1), (False, -1, 1), (True, 32, 0)])
def test_dequant_frag_bit_exact(dt, awq, gs, layout):
    K, N = 512, 272
    if layout == 1:
        N = 256  # (the reference's scale permutation works in runs of 64 columns)
    q = make_quant(rng(3 + gs + layout + awq), K, N, gs, dt, awq)
    wd = dequant_frag(q, K, N, gs, dt, awq, layout)
    got = frag_to_dense(wd.numpy(np.uint16, (K * N,)), K, N)
    assert np.array_equal(got, orc.dequant(q["idx"], q["zeros"], q["scales"], gs, dt))


def test_dequant_frag_interleaves_gate_and_up():
    K, N = 256, 96
    qg, qu = make_quant(rng(1), K, N, 128, BF16), make_quant(rng(2), K, N, 128, BF16)
    wd = dequant_frag(qg, K, N, 128, BF16, False, nv=2 * N, vfrag0=0, vstride=2)
    dequant_frag(qu, K, N, 128, BF16, False, wd=wd, nv=2 * N, vfrag0=1, vstride=2)
    got = frag_to_dense(wd.numpy(np.uint16, (K * 2 * N,)), K, 2 * N).reshape(K, N // 16, 2, 16)
    assert np.array_equal(got[:, :, 0].reshape(K, N), orc.dequant(qg["idx"], None, qg["scales"], 128, BF16))
    assert np.array_equal(got[:, :, 1].reshape(K, N), orc.dequant(qu["idx"], None, qu["scales"], 128, BF16))


# rows: one tile, ragged second tile, three tiles with one row in the last; columns: one wave column short of a tile, a multiple of neither tile
@pytest.mark.parametrize("tile", [128, 256, 256 | 2 << 16, 256 | 4 << 16])  # (| split-K slices << 16: the slices of a tile meet through memory)
@pytest.mark.parametrize("M,K,N", [(256, 128, 256), (300, 512, 1024), (513, 256, 528), (257, 1024, 80)])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_dense_frag_gemm(M, K, N, dt, tile):
    if (K // 64) % (2 * max(tile >> 16, 1)):
        pytest.skip("every slice walks an even number of 64-wide K-steps")
    r = rng(M + K + N + dt)
    q = make_quant(r, K, N, 128, dt, False)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    wd = dequant_frag(q, K, N, 128, dt, False)
    out = ops.DevBuf(M * N * 2).fill_bytes(0xEE)
    L = ops.lib()
    d_x, d_bias, d_res = ops.dev(x), ops.dev(bias), ops.dev(res)
    L.vra_dense_frag_gemm(d_x.ptr, wd.ptr, None, None, out.ptr, M, K, N, 0, dt, tile, 0)
    ops.check_error()
    w = orc.dequant(q["idx"], None, q["scales"], 128, dt)
    ref = orc.gemm_wdense(x, w, None, None, dt)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, dt, name="kernel X", abs_floor=2e-3)
    L.vra_dense_frag_gemm(d_x.ptr, wd.ptr, d_bias.ptr, d_res.ptr, out.ptr, M, K, N, 0, dt, tile, 0)
    ops.check_error()
    ref2 = orc.gemm_wdense(x, w, bias, res, dt)
    g0 = orc.from_dt(ref, dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref2, dt, max_ulp=3.0, name="kernel X bias+residual", mag=mag)


@pytest.mark.parametrize("slices", [2, 4])
def test_dense_frag_gemm_tail_split(slices):
    """more tiles than CUs with a part-filled last round (2 x 136 = 272 tiles on the MI355X's 256 CUs): the first 256 tiles over the full K,
    the 16 of the last round split `slices` ways and summed through memory — same result as the unsplit launch up to the f32 order"""
    M, K, N = 512, 512, 136 * 256
    r = rng(91 + slices)
    q = make_quant(r, K, N, 128, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    wd = dequant_frag(q, K, N, 128, BF16, False)
    out = ops.DevBuf(M * N * 2).fill_bytes(0xEE)
    d_x = ops.dev(x)
    ops.lib().vra_dense_frag_gemm(d_x.ptr, wd.ptr, None, None, out.ptr, M, K, N, 0, BF16, 256 | slices << 24, 0)
    msg = ops.lib().vra_last_error().decode()
    if "tail split does not fit" in msg:  # (a part with another CU count)
        ops.lib().vra_clear_error()
        pytest.skip(msg)
    ops.check_error()
    ref = orc.gemm_wdense(x, orc.dequant(q["idx"], None, q["scales"], 128, BF16), None, None, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name=f"kernel X, tail split {slices}", abs_floor=2e-3)
    assert ops.lib().vra_take_device_error() == 0


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("awq", [False, True])
def test_dense_frag_gemm_gate_up(awq, tile):
    M, K, N = 260, 512, 1040
    r = rng(29 + awq)
    qg, qu = make_quant(r, K, N, 128, BF16, awq), make_quant(r, K, N, 128, BF16, awq)
    x = rand_dt(r, (M, K), BF16)
    wd = dequant_frag(qg, K, N, 128, BF16, awq, nv=2 * N, vfrag0=0, vstride=2)
    dequant_frag(qu, K, N, 128, BF16, awq, wd=wd, nv=2 * N, vfrag0=1, vstride=2)
    out = ops.DevBuf(M * N * 2).fill_bytes(0xEE)
    d_x = ops.dev(x)
    ops.lib().vra_dense_frag_gemm(d_x.ptr, wd.ptr, None, None, out.ptr, M, K, 2 * N, 1, BF16, tile, 0)
    ops.check_error()
    g = orc.wna16_gemm(x, qg["idx"], qg["zeros"], qg["scales"], 128, BF16, marlin_rounded=True)
    u = orc.wna16_gemm(x, qu["idx"], qu["zeros"], qu["scales"], 128, BF16, marlin_rounded=True)
    ref = orc.silu_mul(g, u, BF16)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, max_ulp=3.0, max_mismatch_frac=0.04, name="kernel X gate_up_silu", abs_floor=4e-3)


@pytest.mark.parametrize("dt,awq,gs", [(BF16, False, 128), (F16, True, 128), (BF16, False, -1)])
def test_wna16_gemm_takes_the_dense_path_for_long_prefills(dt, awq, gs, dense_from):
    """vra_wna16_gemm / vra_wna16_gate_up_silu from the row threshold on: Marlin's arithmetic (the oracle's marlin variant within 1 ulp),
    below it the exact product as before"""
    M, K, N = 384, 512, 768
    r = rng(41 + dt + awq)
    q = make_quant(r, K, N, gs, dt, awq)
    x, bias, res = rand_dt(r, (M, K), dt), rand_dt(r, (N,), dt), rand_dt(r, (M, N), dt)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, awq)
    qz = ops.dev(q["qzeros"]) if awq else None
    zeros = q["zeros"] if awq else None
    ref_m = orc.wna16_gemm(x, q["idx"], zeros, q["scales"], gs, dt, marlin_rounded=True)
    ref_e = orc.wna16_gemm(x, q["idx"], zeros, q["scales"], gs, dt)
    dense_from(256)
    got = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), qz, M, K, N, gs, awq, 0, None, None, dt).numpy(np.uint16, (M, N))
    assert_close_dt(got, ref_m, dt, name="dense path vs marlin oracle", abs_floor=2e-3)
    dense_from(0)
    got_e = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), qz, M, K, N, gs, awq, 0, None, None, dt).numpy(np.uint16, (M, N))
    assert_close_dt(got_e, ref_e, dt, name="int4 path vs exact oracle", abs_floor=2e-3)
    # the two arithmetics are different functions: each output sits closer to its own oracle variant
    assert (got != got_e).mean() > 0.01, "the threshold did not switch the arithmetic"
    dense_from(256)
    got2 = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), qz, M, K, N, gs, awq, 0, ops.dev(bias), ops.dev(res), dt).numpy(np.uint16, (M, N))
    ref2 = orc.wna16_gemm(x, q["idx"], zeros, q["scales"], gs, dt, bias, res, marlin_rounded=True)
    g0 = orc.from_dt(ref_m, dt)
    mag = np.maximum(np.abs(g0), np.abs(g0 + orc.from_dt(bias, dt)[None, :]))
    assert_close_dt(got2, ref2, dt, max_ulp=3.0, name="dense path bias+residual", mag=mag)


def test_marlin_ffi_long_prefill(dense_from):
    """the reference boundary at prefill sizes: marlin_4bit_bf16 with the reference's permuted scales (wna16.rs:180-218) on the dense path"""
    M, K, N, gs = 320, 1024, 512, 128
    r = rng(77)
    q = make_quant(r, K, N, gs, BF16, False)
    x = rand_dt(r, (M, K), BF16)
    tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape, 4, False)
    sc = orc.marlin_permute_scales(q["scales"], grouped=True)
    ws = ops.DevBuf(N * 4).zero()
    dense_from(256)
    out = ops.gptq_matmul(ops.dev(x), tiled, ops.dev(sc), ops.dev(q["qzeros"]), None, ws, 4, gs, False, M, K, N, BF16)
    ref = orc.wna16_gemm(x, q["idx"], None, q["scales"], gs, BF16, marlin_rounded=True)
    assert_close_dt(out.numpy(np.uint16, (M, N)), ref, BF16, name="marlin ffi, long prefill", abs_floor=2e-3)
    assert not ws.numpy(np.uint32, (N,)).any()
