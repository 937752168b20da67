"""Shared helpers for the parity tests (oracle = checker; product = libvllm_rs_amd.so)."""
import numpy as np

from oracle import oracle as orc

BF16, F16, F32 = 0, 1, 2


def rng(seed):
    return np.random.default_rng(seed)


def rand_dt(r, shape, dt, scale=1.0):
    return orc.to_dt((r.standard_normal(shape) * scale).astype(np.float32), dt)


def ulp_of(ref_f32, dt):
    """size of one storage ulp at |ref| (bf16: 8 significant bits, f16: 11)."""
    bits = 8 if dt == BF16 else 11
    mag = np.maximum(np.abs(ref_f32), 1e-30)
    return 2.0 ** (np.floor(np.log2(mag)) - (bits - 1))


def assert_close_dt(got_bits, ref_bits, dt, max_ulp=1.0, max_mismatch_frac=0.02, name="", abs_floor=0.0, mag=None):
    """bit patterns equal except for a small fraction of <= max_ulp differences (f32-vs-f64
    accumulation order flipping a rounding).  abs_floor: absolute slack for values near zero where
    cancellation makes ulp-relative comparison meaningless.  mag: magnitude at which the ulp is taken
    when the result went through larger, separately rounded intermediates (default: |ref|)."""
    got, ref = orc.from_dt(got_bits, dt), orc.from_dt(ref_bits, dt)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite output"
    diff = np.abs(got - ref)
    tol = np.maximum(max_ulp * ulp_of(ref if mag is None else np.maximum(np.abs(ref), mag), dt) * 1.0001, abs_floor)
    bad = diff > tol
    frac = float((got_bits != ref_bits).mean())
    assert not bad.any(), f"{name}: {int(bad.sum())} elements beyond {max_ulp} ulp; worst diff {diff.max()} at ref {ref.flat[int(diff.argmax())]}"
    assert frac <= max_mismatch_frac, f"{name}: {frac*100:.2f}% of elements differ (allowed {max_mismatch_frac*100}%)"
    return frac


def make_quant(r, K, N, group_size, dt, awq=False):
    """random int4 problem in checkpoint format. returns dict with idx, zeros(or None), scales (dt bits), packed tensors."""
    g = group_size if group_size > 0 else K
    G = K // g
    idx = r.integers(0, 16, size=(K, N), dtype=np.uint8)
    scales = orc.to_dt((0.002 + 0.018 * r.random((G, N))).astype(np.float32), dt)
    out = {"idx": idx, "scales": scales, "G": G}
    if awq:
        zeros = r.integers(0, 16, size=(G, N), dtype=np.uint8)
        out["zeros"] = zeros
        out["qweight"] = orc.awq_pack(idx)
        out["qzeros"] = orc.awq_pack(zeros)
    else:
        out["zeros"] = None
        out["qweight"] = orc.gptq_pack(idx)
        out["qzeros"] = np.full((G, N // 8), 0x77777777, np.uint32)
    return out
