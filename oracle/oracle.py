"""ctypes/numpy front-end of the CPU ORACLE (oracle/vra_oracle.c).

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  bf16/f16 tensors are numpy uint16 arrays holding the raw bits.

Parity status: pinned against hand-built GPTQ/AWQ bit-layout KATs and HuggingFace transformers
logits (tests/golden/), NOT against the reference's own tests (it has none for this path,
SURVEY.md §8c) — "parity unpinned" w.r.t. the reference's test-suite.
"""
import ctypes as C
import os
import subprocess

import numpy as np

BF16, F16, F32 = 0, 1, 2
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvra_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "vra_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


# ---------------------------------------------------------------- conversions
def to_bf16(x):
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_f32_to_bf16(_p(x), _p(out), C.c_int64(x.size))
    return out


def from_bf16(x):
    x = _c(x, np.uint16)
    out = np.empty(x.shape, np.float32)
    lib().orc_bf16_to_f32(_p(x), _p(out), C.c_int64(x.size))
    return out


def to_f16(x):
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_f32_to_f16(_p(x), _p(out), C.c_int64(x.size))
    return out


def from_f16(x):
    x = _c(x, np.uint16)
    out = np.empty(x.shape, np.float32)
    lib().orc_f16_to_f32(_p(x), _p(out), C.c_int64(x.size))
    return out


def to_dt(x, dt):
    return to_bf16(x) if dt == BF16 else to_f16(x) if dt == F16 else _c(x, np.float32)


def from_dt(x, dt):
    return from_bf16(x) if dt == BF16 else from_f16(x) if dt == F16 else _c(x, np.float32)


def np_dt(dt):
    return np.float32 if dt == F32 else np.uint16


# ---------------------------------------------------------------- synthetic fills
def fill_hash_u32(n, seed):
    out = np.empty(n, np.uint32)
    lib().orc_fill_hash_u32(_p(out), C.c_int64(out.size), C.c_uint64(seed))
    return out


def fill_awq_zeros(n, seed):
    out = np.empty(n, np.uint32)
    lib().orc_fill_awq_zeros(_p(out), C.c_int64(out.size), C.c_uint64(seed))
    return out


def fill_uniform(shape, seed, lo, hi, dt):
    out = np.empty(shape, np_dt(dt))
    lib().orc_fill_uniform(_p(out), C.c_int64(out.size), C.c_uint64(seed), C.c_float(lo), C.c_float(hi), dt)
    return out


def fill_normal(shape, seed, mean, std, dt):
    out = np.empty(shape, np_dt(dt))
    lib().orc_fill_normal(_p(out), C.c_int64(out.size), C.c_uint64(seed), C.c_float(mean), C.c_float(std), dt)
    return out


# ---------------------------------------------------------------- int4 formats
def gptq_unpack(qw, K, N):
    qw = _c(qw, np.uint32)
    idx = np.empty((K, N), np.uint8)
    lib().orc_gptq_unpack(_p(qw), K, N, _p(idx))
    return idx


def gptq_pack(idx):
    idx = _c(idx, np.uint8)
    K, N = idx.shape
    qw = np.empty((K // 8, N), np.uint32)
    lib().orc_gptq_pack(_p(idx), K, N, _p(qw))
    return qw


def gptq_unpack_zeros(qz, G, N):
    qz = _c(qz, np.uint32)
    z = np.empty((G, N), np.uint8)
    lib().orc_gptq_unpack_zeros(_p(qz), G, N, _p(z))
    return z


def awq_unpack(qw, K, N):
    qw = _c(qw, np.uint32)
    idx = np.empty((K, N), np.uint8)
    lib().orc_awq_unpack(_p(qw), K, N, _p(idx))
    return idx


def awq_pack(idx):
    idx = _c(idx, np.uint8)
    K, N = idx.shape
    qw = np.empty((K, N // 8), np.uint32)
    lib().orc_awq_pack(_p(idx), K, N, _p(qw))
    return qw


def awq_unpack_zeros(qz, G, N):
    return awq_unpack(qz, G, N)


def tile_from_indices(idx):
    idx = _c(idx, np.uint8)
    K, N = idx.shape
    out = np.empty((K // 16, N * 2), np.uint32)
    lib().orc_tile_from_indices(_p(idx), K, N, _p(out))
    return out


def tile_to_indices(tiled, K, N):
    tiled = _c(tiled, np.uint32)
    idx = np.empty((K, N), np.uint8)
    lib().orc_tile_to_indices(_p(tiled), K, N, _p(idx))
    return idx


def gptq_repack(qw):
    qw = _c(qw, np.uint32)
    rows, cols = qw.shape
    out = np.empty((rows // 2, cols * 2), np.uint32)
    lib().orc_gptq_repack(_p(qw), rows, cols, _p(out))
    return out


def awq_repack(qw):
    qw = _c(qw, np.uint32)
    rows, cols = qw.shape
    out = np.empty((rows // 16, cols * 16), np.uint32)
    lib().orc_awq_repack(_p(qw), rows, cols, _p(out))
    return out


def marlin_permute_scales(s, grouped=True):
    s = _c(s, np.uint16)
    out = np.empty_like(s)
    lib().orc_marlin_permute_scales(_p(s), _p(out), s.shape[0], s.shape[1], 1 if grouped else 0)
    return out


def dequant(idx, zeros, scales, group_size, dt):
    idx = _c(idx, np.uint8)
    K, N = idx.shape
    zeros = _c(zeros, np.uint8)
    scales = _c(scales)
    w = np.empty((K, N), np_dt(dt))
    lib().orc_dequant(_p(idx), _p(zeros), _p(scales), K, N, group_size, dt, _p(w))
    return w


def gemm_wdense(x, w, bias, residual, dt):
    x, w, bias, residual = _c(x), _c(w), _c(bias), _c(residual)
    M, K = x.shape
    N = w.shape[1]
    out = np.empty((M, N), np_dt(dt))
    lib().orc_gemm_wdense(_p(x), _p(w), _p(bias), _p(residual), M, K, N, dt, _p(out))
    return out


def wna16_gemm(x, idx, zeros, scales, group_size, dt, bias=None, residual=None, row_scale=None, marlin_rounded=False):
    """row_scale [M] f32: the deferred RMSNorm factor, applied to the f32 sums before the output rounding (rms_norm_deferred).
    marlin_rounded: every dequantised weight rounded to dt before the product (what the reference's Marlin kernels compute,
    gptq.rs:116-178) — bit-identical to dequant() + gemm_wdense()."""
    x, idx, zeros, scales, bias, residual = _c(x), _c(idx, np.uint8), _c(zeros, np.uint8), _c(scales), _c(bias), _c(residual)
    M, K = x.shape
    N = idx.shape[1]
    out = np.empty((M, N), np_dt(dt))
    if marlin_rounded:
        assert row_scale is None, "the reference applies the norm before the GEMM"
        lib().orc_wna16_gemm_marlin(_p(x), _p(idx), _p(zeros), _p(scales), _p(bias), _p(residual), M, K, N, group_size, dt, _p(out))
    elif row_scale is None:
        lib().orc_wna16_gemm(_p(x), _p(idx), _p(zeros), _p(scales), _p(bias), _p(residual), M, K, N, group_size, dt, _p(out))
    else:
        rs = _c(row_scale, np.float32)
        assert rs.shape == (M,)
        lib().orc_wna16_gemm_row_scale(_p(x), _p(idx), _p(zeros), _p(scales), _p(bias), _p(residual), _p(rs), M, K, N, group_size, dt, _p(out))
    return out


def gptq_gemv_fast(x, qw, scales, group_size, dt):
    x, qw, scales = _c(x), _c(qw, np.uint32), _c(scales)
    M, K = x.shape
    N = qw.shape[1]
    out = np.empty((M, N), np_dt(dt))
    lib().orc_gptq_gemv_fast(_p(x), _p(qw), _p(scales), M, K, N, group_size, dt, _p(out))
    return out


def dense_gemm(x, w, bias, dt, out_dt):
    x, w, bias = _c(x), _c(w), _c(bias)
    M, K = x.shape
    N = w.shape[0]
    out = np.empty((M, N), np_dt(out_dt))
    lib().orc_dense_gemm(_p(x), _p(w), _p(bias), M, K, N, dt, out_dt, _p(out))
    return out


# ---------------------------------------------------------------- norms / elementwise
def rms_norm(x, w, eps, dt):
    x, w = _c(x), _c(w)
    T, H = x.shape
    out = np.empty((T, H), np_dt(dt))
    lib().orc_rms_norm(_p(x), _p(w), T, H, C.c_float(eps), dt, _p(out))
    return out


def rms_norm_deferred(x, w, eps, dt):
    """-> (round(x * w) [T, H], rstd [T] f32): the order of the engine's 1..4-row decode launches (vra_oracle.c orc_rms_norm_deferred)"""
    x, w = _c(x), _c(w)
    T, H = x.shape
    out = np.empty((T, H), np_dt(dt))
    rstd = np.empty(T, np.float32)
    lib().orc_rms_norm_deferred(_p(x), _p(w), T, H, C.c_float(eps), dt, _p(out), _p(rstd))
    return out, rstd


def add(a, b, dt):
    a, b = _c(a), _c(b)
    out = np.empty(a.shape, np_dt(dt))
    lib().orc_add(_p(a), _p(b), C.c_int64(a.size), dt, _p(out))
    return out


def silu_mul(g, u, dt):
    g, u = _c(g), _c(u)
    out = np.empty(g.shape, np_dt(dt))
    lib().orc_silu_mul(_p(g), _p(u), C.c_int64(g.size), dt, _p(out))
    return out


def embedding(ids, table, dt):
    ids, table = _c(ids, np.uint32), _c(table)
    T, H = ids.shape[0], table.shape[1]
    out = np.empty((T, H), np_dt(dt))
    lib().orc_embedding(_p(ids), _p(table), T, H, dt, _p(out))
    return out


# ---------------------------------------------------------------- rotary
def rope_tables(rot_dim, theta, n_pos, scaling_type=0, factor=1.0, low=1.0, high=4.0, orig_max=8192):
    cos = np.empty((n_pos, rot_dim // 2), np.float32)
    sin = np.empty((n_pos, rot_dim // 2), np.float32)
    lib().orc_rope_tables(rot_dim, C.c_double(theta), scaling_type, C.c_double(factor), C.c_double(low),
                          C.c_double(high), C.c_double(orig_max), n_pos, _p(cos), _p(sin))
    return cos, sin


def rope_tables_ext(rot_dim, theta, n_pos, scaling_type, factor, orig_max, dynamic_alpha=False, beta_fast=32.0, beta_slow=1.0, attn_factor=1.0,
                    extrapolation_factor=1.0):
    """scaling_type 3 = dynamic (NTK; `alpha` form when dynamic_alpha), 4 = yarn (rotary_emb.rs:281-415,435-541)"""
    cos = np.empty((n_pos, rot_dim // 2), np.float32)
    sin = np.empty((n_pos, rot_dim // 2), np.float32)
    lib().orc_rope_tables_ext(rot_dim, C.c_double(theta), scaling_type, C.c_double(factor), int(bool(dynamic_alpha)), C.c_double(orig_max),
                              C.c_double(beta_fast), C.c_double(beta_slow), C.c_double(attn_factor), C.c_double(extrapolation_factor), n_pos,
                              _p(cos), _p(sin))
    return cos, sin


def rope(x, cos, sin, positions, is_interleaved, dt, table_dt, rot_dim=None):
    """x [T,heads,D] (copied), returns rotated copy."""
    x = np.array(x, copy=True)
    T, heads, D = x.shape
    positions = _c(positions, np.int64)
    cos, sin = _c(cos), _c(sin)
    lib().orc_rope(_p(x), T, heads, D, rot_dim or D, _p(cos), _p(sin), _p(positions), 1 if is_interleaved else 0, dt, table_dt)
    return x


# ---------------------------------------------------------------- paged attention
FP8 = 3  # cache dtype id: one OCP E4M3 byte per element (uint8 arrays)


def to_e4m3(x):
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    lib().orc_f32_to_e4m3(_p(x), _p(out), x.size)
    return out


def from_e4m3(x):
    x = _c(x, np.uint8)
    out = np.empty(x.shape, np.float32)
    lib().orc_e4m3_to_f32(_p(x), _p(out), x.size)
    return out


def reshape_and_cache(k, v, kc, vc, slots, BS, dt, kv_dt=None):
    k, v = _c(k), _c(v)
    T, Hkv, D = k.shape
    slots = _c(slots, np.int64)
    lib().orc_reshape_and_cache_kv(_p(k), _p(v), _p(kc), _p(vc), _p(slots), T, Hkv, D, BS, dt, dt if kv_dt is None else kv_dt)


def paged_attention(q, kc, vc, block_tables, context_lens, cu_q, Hkv, BS, scale, dt, softcap=0.0, kv_dt=None, sliding_window=0):
    q = _c(q)
    Tq, Hq, D = q.shape
    block_tables = _c(block_tables, np.uint32)
    context_lens = _c(context_lens, np.uint32)
    cu_q = _c(cu_q, np.uint32)
    B, max_blocks = block_tables.shape
    out = np.empty(q.shape, np_dt(dt))
    lib().orc_paged_attention_kv_sw(_p(out), _p(q), _p(kc), _p(vc), _p(block_tables), _p(context_lens), _p(cu_q), B, Hq, Hkv,
                                    D, BS, max_blocks, C.c_float(scale), C.c_float(softcap), int(sliding_window), dt, dt if kv_dt is None else kv_dt)
    return out


def varlen_attention(q, k, v, cu_q, cu_k, scale, dt, softcap=0.0):
    q, k, v = _c(q), _c(k), _c(v)
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    cu_q, cu_k = _c(cu_q, np.uint32), _c(cu_k, np.uint32)
    out = np.empty(q.shape, np_dt(dt))
    lib().orc_varlen_attention(_p(out), _p(q), _p(k), _p(v), _p(cu_q), _p(cu_k), len(cu_q) - 1, Hq, Hkv, D,
                               C.c_float(scale), C.c_float(softcap), dt)
    return out


def causal_mask(L, sliding_window, dt):
    out = np.empty((L, L), np_dt(dt))
    lib().orc_causal_mask(_p(out), L, sliding_window, dt)
    return out


def argmax_f32(logits):
    logits = _c(logits, np.float32)
    out = np.empty(logits.shape[0], np.uint32)
    lib().orc_argmax_f32(_p(logits), logits.shape[0], logits.shape[1], _p(out))
    return out


# ---------------------------------------------------------------- sampling (LogitsProcessor, src/utils/logits_processor.rs:72-345)
def hash_unit(seed, rows):
    """the uniforms of csrc/common.cuh vra_hash_unit(seed, row): (hash32 >> 8) / 2^24, float32"""
    out = np.empty(rows, np.float32)
    M = (1 << 64) - 1
    for r in range(rows):
        z = (seed * 0x9E3779B97F4A7C15 + r * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z = z ^ (z >> 31)
        out[r] = np.float32((z >> 32) >> 8) * np.float32(1.0 / 16777216.0)
    return out


def sample_candidates(logits_row, top_k, top_p, temperature):
    """sample_topk_topp / sample_topk / sample_topp restated (logits_processor.rs:72-197): probabilities of the full softmax,
    the top_k most probable in descending order (stable: ties keep token order), top-p clamp.  Returns (ids, probs after the
    clamp) — top_k <= 0 with a top_p means k = 256 (the reference's device-sampler convention, :206-213)."""
    x = logits_row.astype(np.float32) * np.float32(1.0 / temperature)
    e = np.exp(x - x.max(), dtype=np.float32)
    prs = e / e.sum(dtype=np.float32)
    has_p = 0.0 < top_p < 1.0
    k = top_k if top_k > 0 else (256 if has_p else 0)
    if k == 0:
        return np.arange(len(prs), dtype=np.uint32), prs
    k = min(k, len(prs), 256)
    order = np.argsort(-x, kind="stable")[:k]            # descending value, ascending token id among ties
    p = prs[order].copy()
    sum_p = p.sum(dtype=np.float32)
    if has_p and top_p < sum_p:
        cumsum = np.float32(0.0)
        for i in range(k):
            if cumsum >= top_p:
                p[i] = 0.0
            else:
                cumsum += p[i]
    return order.astype(np.uint32), p


def sample_draw(ids, probs, u01):
    """WeightedIndex: u in [0, total), first index whose running sum exceeds u"""
    run = np.cumsum(probs.astype(np.float64))
    u = float(u01) * float(run[-1])
    i = int(np.searchsorted(run, u, side="right"))
    i = min(i, len(ids) - 1)
    while probs[i] == 0.0 and i > 0:
        i -= 1
    return int(ids[i]), u, run


def apply_penalties(logits, context, frequency_penalty, presence_penalty):
    """apply_penalties + the guard of apply_batch_repeat_penalty (logits_processor.rs:288-345); one row"""
    out = logits.astype(np.float32).copy()
    if len(context) <= 1 or not ((frequency_penalty not in (0.0, 1.0)) or (presence_penalty not in (0.0, 1.0))):
        return out
    counts = np.zeros(len(out), np.float32)
    for t in context:
        if t < len(out):
            counts[t] += 1.0
    return (out - counts * np.float32(frequency_penalty) - (counts > 0).astype(np.float32) * np.float32(presence_penalty)).astype(np.float32)
