"""ctypes loader for libvllm_rs_amd.so (the C-ABI shared library, include/vllm_rs_amd.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this module
raises — callers must build it first (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VRA_LIB", os.path.join(_PKG, "libvllm_rs_amd.so"))  # VRA_LIB: A/B kernel experiments
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "vllm_rs_amd.h")

_lib = None

c_i32, c_i64, c_u64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p, C.c_size_t


class ModelConfig(C.Structure):
    """vra_model_config — fields of Config/QuantConfig the hot path consumes (config.rs:218-255,735-757)."""
    _fields_ = [
        ("arch", c_i32), ("hidden_size", c_i32), ("intermediate_size", c_i32), ("num_layers", c_i32),
        ("num_heads", c_i32), ("num_kv_heads", c_i32), ("head_dim", c_i32), ("vocab_size", c_i32),
        ("max_position_embeddings", c_i32), ("rms_norm_eps", c_f32), ("rope_theta", C.c_double),
        ("rope_scaling_type", c_i32), ("rope_factor", C.c_double), ("rope_low_freq_factor", C.c_double),
        ("rope_high_freq_factor", C.c_double), ("rope_original_max_position", c_i32),
        ("attention_bias", c_i32), ("quant_method", c_i32), ("bits", c_i32), ("group_size", c_i32),
        ("dtype", c_i32), ("tie_word_embeddings", c_i32),
        ("rope_dynamic_alpha", c_i32), ("rope_yarn_beta_fast", C.c_double), ("rope_yarn_beta_slow", C.c_double),
        ("rope_yarn_attn_factor", C.c_double), ("rope_yarn_extrapolation_factor", C.c_double),
        ("rope_original_max_position_f", C.c_double), ("rope_yarn_explicit", c_i32), ("qk_norm", c_i32), ("sliding_window", c_i32),
    ]


class EngineConfig(C.Structure):
    """vra_engine_config — EngineConfig subset (config.rs:285-328)."""
    _fields_ = [
        ("block_size", c_i32), ("max_num_seqs", c_i32), ("max_model_len", c_i32), ("num_gpu_blocks", c_i32),
        ("kv_fraction", c_f32), ("prefill_chunk", c_i32), ("enable_prefix_cache", c_i32),
        ("prefix_cache_fraction", c_f32), ("use_graph", c_i32), ("tp_rank", c_i32), ("tp_world_size", c_i32),
        ("device", c_i32), ("seed", c_u64), ("fp8_kvcache", c_i32), ("cpu_mem_fold", c_f32),
        ("swap_cooling_ms", c_i32), ("min_tokens_left_for_swap", c_i32),
    ]


class SamplingParams(C.Structure):
    """vra_sampling_params (config.rs:476-520): unset = temperature < 0, top_k <= 0, top_p < 0, has_* = 0"""
    _fields_ = [("temperature", c_f32), ("top_k", c_i32), ("top_p", c_f32), ("has_frequency_penalty", c_i32),
                ("has_presence_penalty", c_i32), ("frequency_penalty", c_f32), ("presence_penalty", c_f32)]


MISSING = []


class StepMeta(C.Structure):
    """vra_step_meta (include/vllm_rs_amd.h): host view of one scheduled step of a host-only engine."""
    _fields_ = [("n_tokens", C.c_int32), ("n_seqs", C.c_int32), ("max_blocks", C.c_int32), ("max_seqlen_q", C.c_int32),
                ("max_context_len", C.c_int32), ("input_ids", C.POINTER(C.c_uint32)), ("positions", C.POINTER(C.c_int64)),
                ("slot_mapping", C.POINTER(C.c_int64)), ("block_tables", C.POINTER(C.c_uint32)),
                ("context_lens", C.POINTER(C.c_uint32)), ("cu_seqlens_q", C.POINTER(C.c_uint32)),
                ("request_ids", C.c_int64 * 64)]


def _sig(lib, name, restype, *argtypes):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        MISSING.append(name)  # reported by tests/test_abi.py; calling it raises AttributeError
        return None
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


def load():
    """Load the shared library; raises OSError/RuntimeError loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not built. Run `make -C vllm_rs_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = c_vp
    marlin = (P, P, P, P, P, P, c_i32, c_i32, c_i32, P, c_i32, c_i64)
    for n in ("marlin_4bit_bf16", "marlin_4bit_f16", "marlin_awq_4bit_bf16", "marlin_awq_4bit_f16"):
        _sig(lib, n, None, *marlin)
    _sig(lib, "gemm_half_q_half_alt", None, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "gptq_repack", None, P, P, c_i32, c_i32, c_i64)
    _sig(lib, "awq_repack", None, P, P, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_last_error", C.c_char_p)
    _sig(lib, "vra_clear_error", None)
    _sig(lib, "vra_take_device_error", C.c_int32)
    _sig(lib, "vra_version", C.c_char_p)
    _sig(lib, "vra_wna16_gemm", None, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_wna16_gate_up_silu", None, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_rms_norm_wna16_gemm", None, P, P, c_f32, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_rms_norm_wna16_gate_up_silu", None, P, P, c_f32, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_wna16_unpack_indices", None, P, P, c_i32, c_i32, c_i64)
    _sig(lib, "vra_wna16_dequant", None, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_wna16_dequant_frag", None, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_dense_frag_gemm", None, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_rms_norm", None, P, P, P, c_i32, c_i32, c_f32, c_i32, c_i64)
    _sig(lib, "vra_qk_rms_norm", None, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i64)
    _sig(lib, "vra_add_rms_norm", None, P, P, P, P, P, c_i32, c_i32, c_f32, c_i32, c_i64)
    _sig(lib, "vra_add", None, P, P, P, c_i64, c_i32, c_i64)
    _sig(lib, "vra_embedding", None, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_index_select_rows", None, P, P, P, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_silu_mul", None, P, P, P, c_i64, c_i32, c_i64)
    _sig(lib, "vra_fused_rope", None, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_reshape_and_cache", None, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_rope_cache_prefill", None, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_paged_attention_decode_workspace_bytes", c_sz, c_i32, c_i32, c_i32, c_i32)
    _sig(lib, "vra_paged_attention_decode", None, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
         c_f32, c_f32, P, c_i32, c_i32, c_i64)
    _sig(lib, "vra_paged_attention_prefill", None, P, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32,
         c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_paged_attention_decode_sw", None, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
         c_f32, c_f32, c_i32, P, c_i32, c_i32, c_i64)
    _sig(lib, "vra_paged_attention_prefill_sw", None, P, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32,
         c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_rope_cache_attention_decode", None, P, P, P, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32,
         c_i32, c_i32, c_i32, c_f32, P, c_i32, c_i32, c_i64)
    _sig(lib, "vra_causal_mask", None, P, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_swap_blocks", None, P, P, P, c_i32, c_i64, c_i32, c_i64)
    _sig(lib, "vra_dense_gemm", None, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_argmax_f32", None, P, P, c_i32, c_i32, c_i64)
    _sig(lib, "vra_dense_gemm_argmax_workspace_bytes", c_i64)
    _sig(lib, "vra_dense_gemm_argmax", None, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i64)
    _sig(lib, "vra_cast", None, P, P, c_i64, c_i32, c_i32, c_i64)
    _sig(lib, "vra_sample", None, P, P, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, P, P, c_i64)
    _sig(lib, "vra_apply_penalties", None, P, P, P, c_i32, c_i32, c_i32, P, P, c_i64)
    _sig(lib, "vra_fill_hash_u32", None, P, c_i64, c_u64, c_i64)
    _sig(lib, "vra_fill_awq_zeros", None, P, c_i64, c_u64, c_i64)
    _sig(lib, "vra_fill_uniform", None, P, c_i64, c_u64, c_f32, c_f32, c_i32, c_i64)
    _sig(lib, "vra_fill_normal", None, P, c_i64, c_u64, c_f32, c_f32, c_i32, c_i64)
    _sig(lib, "vra_fill_const_u32", None, P, c_i64, C.c_uint32, c_i64)
    _sig(lib, "vra_comm_unique_id", c_i32, P)
    _sig(lib, "vra_comm_create", P, P, c_i32, c_i32, c_i32)
    _sig(lib, "vra_comm_destroy", None, P)
    _sig(lib, "vra_comm_rank", c_i32, P)
    _sig(lib, "vra_comm_world_size", c_i32, P)
    _sig(lib, "vra_all_reduce", None, P, P, P, c_i64, c_i32, c_i64)
    _sig(lib, "vra_comm_ipc_begin", P, P, c_i32, c_i32, c_i32, P)
    _sig(lib, "vra_comm_ipc_connect", c_i32, P, P, c_i64)
    _sig(lib, "vra_all_reduce_fused", None, P, P, P, P, P, c_i64, c_i32, c_i32, c_i64)
    _sig(lib, "vra_comm_take_error", c_i32, P)
    _sig(lib, "vra_comm_error_detail", c_i32, P, P)
    _sig(lib, "vra_comm_error_word", P, P)
    _sig(lib, "vra_device_count", c_i32)
    _sig(lib, "vra_set_device", c_i32, c_i32)
    _sig(lib, "vra_malloc", P, c_sz)
    _sig(lib, "vra_free", None, P)
    _sig(lib, "vra_malloc_host", P, c_sz)
    _sig(lib, "vra_free_host", None, P)
    _sig(lib, "vra_memcpy_h2d", c_i32, P, P, c_sz, c_i64)
    _sig(lib, "vra_memcpy_d2h", c_i32, P, P, c_sz, c_i64)
    _sig(lib, "vra_memcpy_d2d", c_i32, P, P, c_sz, c_i64)
    _sig(lib, "vra_memset", c_i32, P, c_i32, c_sz, c_i64)
    _sig(lib, "vra_stream_sync", c_i32, c_i64)
    _sig(lib, "vra_device_sync", c_i32)
    _sig(lib, "vra_stream_create", c_i64)
    _sig(lib, "vra_stream_destroy", None, c_i64)
    _sig(lib, "vra_mem_info", c_i32, C.POINTER(c_sz), C.POINTER(c_sz))
    _sig(lib, "vra_event_create", P)
    _sig(lib, "vra_event_destroy", None, P)
    _sig(lib, "vra_event_record", c_i32, P, c_i64)
    _sig(lib, "vra_event_elapsed_ms", c_f32, P, P)
    # host runtime (Section C)
    MC, EC = C.POINTER(ModelConfig), C.POINTER(EngineConfig)
    _sig(lib, "vra_kv_per_block_bytes", c_i64, MC, EC)
    _sig(lib, "vra_kv_plan_num_blocks", c_i64, MC, EC, c_i64)
    _sig(lib, "vra_rope_tables_f32", None, MC, c_i32, P, P)
    _sig(lib, "vra_rope_table_rows", c_i32, MC)
    _sig(lib, "vra_marlin_permute_scales_u16", None, P, P, c_i32, c_i32, c_i32)
    _sig(lib, "vra_bm_create", P, c_i32, c_i32, c_i32, c_f32)
    _sig(lib, "vra_bm_destroy", None, P)
    _sig(lib, "vra_bm_num_free_blocks", c_i32, P)
    _sig(lib, "vra_bm_seq_create", c_i64, P, P, c_i32)
    _sig(lib, "vra_bm_seq_free", None, P, c_i64)
    _sig(lib, "vra_bm_can_allocate", c_i32, P, c_i64)
    _sig(lib, "vra_bm_allocate", c_i32, P, c_i64)
    _sig(lib, "vra_bm_can_append", c_i32, P, c_i64)
    _sig(lib, "vra_bm_may_append", c_i32, P, c_i64)
    _sig(lib, "vra_bm_append_token", None, P, c_i64, C.c_uint32)
    _sig(lib, "vra_bm_deallocate", None, P, c_i64)
    _sig(lib, "vra_bm_seq_len", c_i32, P, c_i64)
    _sig(lib, "vra_bm_seq_num_cached_tokens", c_i32, P, c_i64)
    _sig(lib, "vra_bm_seq_block_table", c_i32, P, c_i64, P, c_i32)
    _sig(lib, "vra_bm_prefix_cached_blocks", c_i32, P)
    _sig(lib, "vra_bm_evict_prefix", c_i32, P, c_i32)
    _sig(lib, "vra_pc_create", P, c_i32, c_i32)
    _sig(lib, "vra_pc_destroy", None, P)
    _sig(lib, "vra_pc_insert_prefix", c_i32, P, P, c_i32, P, c_i32, P, c_i32, P)
    _sig(lib, "vra_pc_match_prefix", c_i32, P, P, c_i32, P, c_i32)
    _sig(lib, "vra_pc_cached_blocks", c_i32, P)
    _sig(lib, "vra_pc_evict_blocks", c_i32, P, c_i32, P, c_i32)
    _sig(lib, "vra_engine_create", P, MC, EC)
    _sig(lib, "vra_engine_destroy", None, P)
    _sig(lib, "vra_engine_init_synthetic", c_i32, P)
    _sig(lib, "vra_engine_load_tensor", c_i32, P, C.c_char_p, P, P, c_i32, c_i32)
    _sig(lib, "vra_engine_finalize_weights", c_i32, P)
    _sig(lib, "vra_engine_copy_logits", c_i32, P, P, c_i32)
    _sig(lib, "vra_engine_debug_tp_snapshots", None, P, c_i32)
    _sig(lib, "vra_engine_norm_deferred", c_i32, P, c_i32, c_i32)
    _sig(lib, "vra_debug_dense_prefill_min_rows", c_i32)
    _sig(lib, "vra_debug_set_dense_prefill_min_rows", None, c_i32)
    _sig(lib, "vra_debug_norm_deferred_mask", c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32)
    _sig(lib, "vra_debug_gemv_s_fits", c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32)
    _sig(lib, "vra_engine_debug_read_tp_snapshot", c_i64, P, c_i32, P, c_i64)
    _sig(lib, "vra_engine_finalize_model", c_i32, P)
    _sig(lib, "vra_engine_update_config", c_i32, P, P)
    _sig(lib, "vra_engine_num_gpu_blocks", c_i32, P)
    _sig(lib, "vra_engine_swap_stats", None, P, P)
    _sig(lib, "vra_engine_swap_blocks", c_i32, P, P, c_i32, c_i32)
    _sig(lib, "vra_engine_plan_kv_blocks", c_i64, P)
    _sig(lib, "vra_engine_set_num_gpu_blocks", c_i32, P, c_i32)
    _sig(lib, "vra_engine_add_request", c_i64, P, P, c_i32, c_i32, c_i32, P, c_i32)
    _sig(lib, "vra_engine_add_request_ex", c_i64, P, P, c_i32, c_i32, c_i32, P, c_i32, P)
    _sig(lib, "vra_engine_step", c_i32, P, P)
    _sig(lib, "vra_engine_dry_schedule", c_i32, P, P, P)
    _sig(lib, "vra_engine_dry_commit", c_i32, P, P, c_i32)
    _sig(lib, "vra_engine_has_unfinished", c_i32, P)
    _sig(lib, "vra_engine_request_finished", c_i32, P, c_i64)
    _sig(lib, "vra_engine_request_output", c_i32, P, c_i64, P, c_i32)
    _sig(lib, "vra_engine_request_times", c_i32, P, c_i64, P)
    _sig(lib, "vra_engine_release_request", None, P, c_i64)
    _sig(lib, "vra_engine_forward_raw", c_i32, P, P, P, P, c_i32, c_i32, P, c_i32, P, P, c_i32, P)
    _sig(lib, "vra_engine_forward_tokens", c_i32, P, P, P, P, c_i32, c_i32, P, c_i32, P, P, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, P)
    _sig(lib, "vra_engine_timed_decode", C.c_double, P, c_i32)
    _sig(lib, "vra_engine_bench_replay", C.c_double, P, c_i32)
    _sig(lib, "vra_engine_set_comm", c_i32, P, P)
    _sig(lib, "vra_engine_bench_gemm", C.c_double, P, c_i32, c_i32, c_i32)
    _sig(lib, "vra_engine_gemm_bytes", c_i64, P, c_i32, c_i32)
    _sig(lib, "vra_engine_weight_bytes", c_i64, P)
    _sig(lib, "vra_engine_stream", c_i64, P)
    _sig(lib, "vra_engine_last_error", C.c_char_p, P)
    _lib = lib
    return lib


def declared_symbols():
    """Every function name declared in include/vllm_rs_amd.h (used by the CPU export test)."""
    import re
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", txt)
    return sorted(set(n for n in names if n.startswith(("vra_", "marlin_", "gemm_half", "gptq_repack", "awq_repack"))))
