// attn_launch.h — internal (native runtime) entry points of attention.hip beside the C ABI of include/vllm_rs_amd.h
#pragma once
#include <stdint.h>

// vra_rope_cache_attention_decode with the output ALSO in kernel W's fragment order (rows 0..31, K = q_heads * head_dim, a multiple
// of 128; gemv_q4s.cuh GemvSArgs::x_frag) for the o_proj launch that follows; out_frag = null: exactly the public entry point
void vra_rope_cache_attention_decode_frag(void* out, const void* q, const void* k, const void* v, void* k_cache, void* v_cache, const void* cos,
                                          const void* sin, const int64_t* positions, const int64_t* slot_mapping, const uint32_t* block_tables,
                                          const uint32_t* context_lens, int32_t batch, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                          int32_t block_size, int32_t max_blocks_per_seq, int32_t max_context_len, float scale, void* workspace,
                                          int32_t dtype, int32_t kv_dtype, void* out_frag, int64_t stream);
