"""Host-side mirror (Python, ctypes) of the reference's operator interface for this path.

Names and argument meaning follow the Rust call sites so the parity tests read like the reference:
`gptq_matmul` / `marlin_weight_repack` (src/utils/gptq.rs:243-263,357-360), `WNA16`
(src/models/layers/wna16.rs), `rms_norm`, `FusedRope.apply_inplace`, `PagedAttention.forward`.
Everything executes in libvllm_rs_amd.so on the GPU; there is no CPU fallback here.
"""
import ctypes as C

import numpy as np

from . import _lib

BF16, F16, F32 = 0, 1, 2
SCALES_ROWMAJOR, SCALES_MARLIN = 0, 1


def lib():
    return _lib.load()


def check_error():
    msg = lib().vra_last_error().decode()
    if msg:
        lib().vra_clear_error()
        raise RuntimeError(msg)


class DevBuf:
    """A caller-owned device buffer (the role candle's CudaStorage plays in the reference)."""

    def __init__(self, nbytes=None, array=None):
        L = lib()
        if array is not None:
            array = np.ascontiguousarray(array)
            nbytes = array.nbytes
        self.nbytes = int(nbytes)
        self.ptr = L.vra_malloc(max(self.nbytes, 16))
        if not self.ptr:
            raise MemoryError(L.vra_last_error().decode())
        if array is not None and self.nbytes:
            L.vra_memcpy_h2d(self.ptr, array.ctypes.data_as(C.c_void_p), self.nbytes, 0)
            L.vra_stream_sync(0)

    def zero(self):
        lib().vra_memset(self.ptr, 0, self.nbytes, 0)
        return self

    def fill_bytes(self, value):
        lib().vra_memset(self.ptr, value, self.nbytes, 0)
        return self

    def numpy(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        lib().vra_device_sync()
        lib().vra_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 0)
        lib().vra_stream_sync(0)
        return out

    def free(self):
        if self.ptr:
            lib().vra_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def dev(array):
    return DevBuf(array=array)


def _ptr(b):
    return None if b is None else (b.ptr if isinstance(b, DevBuf) else b)


# ---------------------------------------------------------------- gptq.rs mirror
def marlin_weight_repack(qweight_dev, shape, bits=4, is_awq=False):
    """MarlinRepack (src/utils/gptq.rs:266-360): returns a new device tensor, host shape [k/16, n*2]."""
    rows, cols = shape
    out = DevBuf(rows * cols * 4)
    if is_awq:
        lib().awq_repack(_ptr(qweight_dev), out.ptr, rows, cols, bits, 0)
    else:
        lib().gptq_repack(_ptr(qweight_dev), out.ptr, rows, cols, 0)
    check_error()
    return out


def gptq_matmul(x, qweight, scale, qzeros, g_idx, workspace, bits, group_size, is_awq, m, k, n, dtype=BF16):
    """gptq_matmul (src/utils/gptq.rs:243-263) → marlin_* when `workspace` is given, else
    gemm_half_q_half_alt. All tensors are DevBufs; returns the output DevBuf [m, n]."""
    assert bits == 4 or (bits == 8 and workspace is None)
    out = DevBuf(m * n * 2)
    L = lib()
    if workspace is not None:
        fn = {(BF16, False): L.marlin_4bit_bf16, (F16, False): L.marlin_4bit_f16,
              (BF16, True): L.marlin_awq_4bit_bf16, (F16, True): L.marlin_awq_4bit_f16}[(dtype, bool(is_awq))]
        fn(_ptr(x), _ptr(qweight), _ptr(scale), _ptr(qzeros), _ptr(g_idx), out.ptr, m, k, n, _ptr(workspace), group_size, 0)
    else:
        assert dtype == F16, "GPTQMatMul is only supported for f16 non-marlin matmul (gptq.rs:197)"
        L.gemm_half_q_half_alt(_ptr(x), _ptr(qweight), _ptr(qzeros), _ptr(scale), _ptr(g_idx), out.ptr, m, n, k, bits, 0)
    check_error()
    return out


def wna16_gemm(x, qweight_tiled, scales, qzeros, m, k, n, group_size, is_awq=False, scales_layout=SCALES_ROWMAJOR,
               bias=None, residual=None, dtype=BF16):
    out = DevBuf(m * n * 2)
    lib().vra_wna16_gemm(_ptr(x), _ptr(qweight_tiled), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(residual), out.ptr,
                         m, k, n, group_size, int(is_awq), scales_layout, dtype, 0)
    check_error()
    return out


def wna16_gate_up_silu(x, qw_g, sc_g, qz_g, qw_u, sc_u, qz_u, m, k, n, group_size, is_awq=False,
                       scales_layout=SCALES_ROWMAJOR, dtype=BF16):
    out = DevBuf(m * n * 2)
    lib().vra_wna16_gate_up_silu(_ptr(x), _ptr(qw_g), _ptr(sc_g), _ptr(qz_g), _ptr(qw_u), _ptr(sc_u), _ptr(qz_u), out.ptr,
                                 m, k, n, group_size, int(is_awq), scales_layout, dtype, 0)
    check_error()
    return out


def rms_norm_wna16_gemm(x, norm_w, eps, qweight_tiled, scales, qzeros, m, k, n, group_size, is_awq=False,
                        scales_layout=SCALES_ROWMAJOR, bias=None, dtype=BF16):
    out, ws = DevBuf(m * n * 2), DevBuf(m * k * 2)
    lib().vra_rms_norm_wna16_gemm(_ptr(x), _ptr(norm_w), eps, _ptr(qweight_tiled), _ptr(scales), _ptr(qzeros), _ptr(bias), out.ptr,
                                  ws.ptr, m, k, n, group_size, int(is_awq), scales_layout, dtype, 0)
    check_error()
    return out


def rms_norm_wna16_gate_up_silu(x, norm_w, eps, qw_g, sc_g, qz_g, qw_u, sc_u, qz_u, m, k, n, group_size, is_awq=False,
                                scales_layout=SCALES_ROWMAJOR, dtype=BF16):
    out, ws = DevBuf(m * n * 2), DevBuf(m * k * 2)
    lib().vra_rms_norm_wna16_gate_up_silu(_ptr(x), _ptr(norm_w), eps, _ptr(qw_g), _ptr(sc_g), _ptr(qz_g), _ptr(qw_u), _ptr(sc_u),
                                          _ptr(qz_u), out.ptr, ws.ptr, m, k, n, group_size, int(is_awq), scales_layout, dtype, 0)
    check_error()
    return out


def unpack_indices(qweight_tiled, k, n):
    out = DevBuf(k * n)
    lib().vra_wna16_unpack_indices(_ptr(qweight_tiled), out.ptr, k, n, 0)
    check_error()
    return out.numpy(np.uint8, (k, n))


def dequant(qweight_tiled, scales, qzeros, k, n, group_size, is_awq=False, scales_layout=SCALES_ROWMAJOR, dtype=BF16):
    out = DevBuf(k * n * 2)
    lib().vra_wna16_dequant(_ptr(qweight_tiled), _ptr(scales), _ptr(qzeros), out.ptr, k, n, group_size, int(is_awq),
                            scales_layout, dtype, 0)
    check_error()
    return out.numpy(np.uint16, (k, n))


def dense_gemm(x, w, bias, m, k, n, dtype=BF16, out_dtype=None):
    out_dtype = dtype if out_dtype is None else out_dtype
    out = DevBuf(m * n * (4 if out_dtype == F32 else 2))
    lib().vra_dense_gemm(_ptr(x), _ptr(w), _ptr(bias), out.ptr, m, k, n, dtype, out_dtype, 0)
    check_error()
    return out


# ---------------------------------------------------------------- others.rs / mlp.rs mirror
def rms_norm(x, weight, tokens, hidden, eps, dtype=BF16):
    out = DevBuf(tokens * hidden * 2)
    lib().vra_rms_norm(_ptr(x), _ptr(weight), out.ptr, tokens, hidden, eps, dtype, 0)
    check_error()
    return out


def qk_rms_norm(q, k, q_weight, k_weight, tokens, q_heads, kv_heads, head_dim, full_dim, eps, dtype=BF16):
    """in place on the device buffers q [tokens, q_heads, head_dim] and k [tokens, kv_heads, head_dim] (attention.rs:713-735)"""
    lib().vra_qk_rms_norm(_ptr(q), _ptr(k), _ptr(q_weight), _ptr(k_weight), tokens, q_heads, kv_heads, head_dim, int(full_dim), eps, dtype, 0)
    check_error()
    return q, k


def add_rms_norm(x, residual, weight, tokens, hidden, eps, dtype=BF16):
    h, out = DevBuf(tokens * hidden * 2), DevBuf(tokens * hidden * 2)
    lib().vra_add_rms_norm(_ptr(x), _ptr(residual), _ptr(weight), h.ptr, out.ptr, tokens, hidden, eps, dtype, 0)
    check_error()
    return h, out


def add(a, b, numel, dtype=BF16):
    out = DevBuf(numel * 2)
    lib().vra_add(_ptr(a), _ptr(b), out.ptr, numel, dtype, 0)
    check_error()
    return out


def silu_mul(gate, up, numel, dtype=BF16):
    out = DevBuf(numel * 2)
    lib().vra_silu_mul(_ptr(gate), _ptr(up), out.ptr, numel, dtype, 0)
    check_error()
    return out


def embedding(ids, table, tokens, hidden, vocab, dtype=BF16):
    out = DevBuf(tokens * hidden * 2)
    lib().vra_embedding(_ptr(ids), _ptr(table), out.ptr, tokens, hidden, vocab, dtype, 0)
    check_error()
    return out


class DenseGemmArgmax:
    """lm_head + greedy token in one launch (vra_dense_gemm_argmax); owns the zeroed workspace the launches share"""

    def __init__(self):
        self.ws = DevBuf(lib().vra_dense_gemm_argmax_workspace_bytes()).zero()

    def __call__(self, x, w, bias, m, k, n, dtype=BF16):
        logits, toks = DevBuf(m * n * 4), DevBuf(m * 4)
        lib().vra_dense_gemm_argmax(_ptr(x), _ptr(w), _ptr(bias), logits.ptr, toks.ptr, self.ws.ptr, m, k, n, dtype, 0)
        check_error()
        return logits, toks.numpy(np.uint32, (m,))


def argmax(logits, rows, cols):
    out = DevBuf(rows * 4)
    lib().vra_argmax_f32(_ptr(logits), out.ptr, rows, cols, 0)
    check_error()
    return out.numpy(np.uint32, (rows,))


# ---------------------------------------------------------------- rotary_emb.rs / attention.rs mirror
class FusedRope:
    """attention_rs::fused_rope::FusedRope (rotary_emb.rs:88-103)."""

    @staticmethod
    def apply_inplace(q, k, cos, sin, positions, is_rope_i, tokens, q_heads, kv_heads, head_dim, dtype=BF16,
                      table_dtype=None, rot_dim=None):
        lib().vra_fused_rope(_ptr(q), _ptr(k), _ptr(cos), _ptr(sin), _ptr(positions), tokens, q_heads, kv_heads, head_dim,
                             rot_dim or head_dim, int(is_rope_i), dtype, dtype if table_dtype is None else table_dtype, 0)
        check_error()


class PagedAttention:
    """attention_rs::PagedAttention (attention.rs:607-616,808-820): new(heads, D, scale, kv_heads, …)."""

    def __init__(self, num_heads, head_dim, scale, num_kv_heads, block_size=64, dtype=BF16, softcap=0.0, fp8_kvcache=False, sliding_window=0):
        self.Hq, self.D, self.scale, self.Hkv, self.BS, self.dtype, self.softcap = num_heads, head_dim, scale, num_kv_heads, block_size, dtype, softcap
        self.sliding_window = int(sliding_window or 0)  # PagedAttention::new(.., sliding_window, ..), attention.rs:607-616
        self.kv_dtype = 3 if fp8_kvcache else dtype   # VRA_FP8_E4M3 (PagedAttention::new(.., fp8_kvcache), attention.rs:607-616)

    def reshape_and_cache(self, k, v, k_cache, v_cache, slot_mapping, tokens):
        lib().vra_reshape_and_cache(_ptr(k), _ptr(v), _ptr(k_cache), _ptr(v_cache), _ptr(slot_mapping), tokens, self.Hkv,
                                    self.D, self.BS, self.dtype, self.kv_dtype, 0)
        check_error()

    def forward_decode(self, q, k_cache, v_cache, block_tables, context_lens, batch, max_blocks, max_context_len,
                       workspace=None):
        out = DevBuf(batch * self.Hq * self.D * 2)
        lib().vra_paged_attention_decode_sw(out.ptr, _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(block_tables),
                                            _ptr(context_lens), batch, self.Hq, self.Hkv, self.D, self.BS, max_blocks,
                                            max_context_len, self.scale, self.softcap, self.sliding_window, _ptr(workspace), self.dtype,
                                            self.kv_dtype, 0)
        check_error()
        return out

    def rope_cache_decode(self, q, k, v, k_cache, v_cache, cos, sin, positions, slot_mapping, block_tables, context_lens,
                          batch, max_blocks, max_context_len, workspace=None):
        """fused decode step: RoPE(q,k) + cache write + paged attention in one launch (vra_rope_cache_attention_decode)."""
        out = DevBuf(batch * self.Hq * self.D * 2)
        lib().vra_rope_cache_attention_decode(out.ptr, _ptr(q), _ptr(k), _ptr(v), _ptr(k_cache), _ptr(v_cache), _ptr(cos), _ptr(sin),
                                              _ptr(positions), _ptr(slot_mapping), _ptr(block_tables), _ptr(context_lens), batch,
                                              self.Hq, self.Hkv, self.D, self.BS, max_blocks, max_context_len, self.scale,
                                              _ptr(workspace), self.dtype, self.kv_dtype, 0)
        check_error()
        return out

    def forward_prefill(self, q, total_q, max_seqlen_q, cu_q, batch, k=None, v=None, cu_k=None, k_cache=None,
                        v_cache=None, block_tables=None, context_lens=None, max_blocks=0):
        out = DevBuf(total_q * self.Hq * self.D * 2)
        lib().vra_paged_attention_prefill_sw(out.ptr, _ptr(q), _ptr(k), _ptr(v), _ptr(k_cache), _ptr(v_cache),
                                             _ptr(block_tables), _ptr(context_lens), _ptr(cu_q), _ptr(cu_k), batch, total_q,
                                             max_seqlen_q, self.Hq, self.Hkv, self.D, self.BS, max_blocks, self.scale,
                                             self.softcap, self.sliding_window, self.dtype, self.kv_dtype, 0)
        check_error()
        return out
