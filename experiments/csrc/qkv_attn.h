// qkv_attn.h — the fused decode launch "RMSNorm + q/k/v GEMV + RoPE + KV-cache write + paged attention" (qkv_attn.hip):
// internal launcher used by the native runtime (host/model.cpp) for decode steps of 1..4 sequences.
#pragma once
#include <stdint.h>

#include "gemv_q4s.cuh"  // (vllm_rs_amd/csrc, on the include path of the EXPERIMENTS build)

struct QkvAttnTail {
  const uint32_t* epoch;  // device word, bumped once per forward BEFORE this launch (vra_embedding_bump): granule tags are
                          // (epoch << 8) + layer_tag, unique per (forward, layer) — a captured graph replays with fresh tags
  int layer_tag;          // 1 + layer index (< 256)
  int k_unit0, v_unit0;   // first 16-column unit of the k / v segment inside the fused q|k|v launch
  int uph;                // units per head (head_dim / 16)
  int attn_stride;        // the attention of (sequence s, kv head h) runs on the workgroup that owns unit k_unit0 + h*uph + s*attn_stride
  // attention (FusedDecodeArgs of attention.hip, 16-bit KV cache)
  void* out;                     // [B, Hq, D]
  void* kc;                      // K cache [NB, Hkv, BS, D]
  void* vc;                      // V cache [NB, Hkv, D, BS]
  const void* cosv;              // [n_pos, D/2] model dtype
  const void* sinv;
  const int64_t* positions;      // [B]
  const int64_t* slots;          // [B] (negative: padded lane, nothing is written)
  const uint32_t* block_tables;  // [B, max_blocks]
  const uint32_t* context_lens;  // [B] (includes the new token)
  int B, Hq, Hkv, BS, max_blocks, bs_shift;
  float scale_log2e;
  uint32_t* err;  // device error word (scratch): a granule wait that timed out
};

// shape / configuration test (no launch): rows 1..4 == sequences, 16-bit KV cache, head_dim 64 / 128, group <= 8 q heads per
// kv head, block size a multiple of 32, contexts up to `max_context_len` walked by ONE workgroup per (sequence, kv head)
bool vra_qkv_attn_fits(int M, int K, int group_size, int n_units, int Hq, int Hkv, int D, int BS, int kv_dtype, int dtype, int max_context_len);
// a.seg[] / a.nseg / bias pointers as for the plain q/k/v launch of kernel E (outputs are NOT written to a.seg[i].out: they
// only exist as granules in `gran`, [M][gran_ld] x 8 bytes); t.k_unit0 / v_unit0 / uph / attn_stride are filled here.
void vra_launch_qkv_attn(GemvSArgs a, QkvAttnTail t, void* gran, int group_size, bool awq, int dtype, int D, int64_t stream);
// bytes of the granule buffer for up to `max_rows` rows of q|k|v
size_t vra_qkv_attn_granule_bytes(int max_rows, int Hq, int Hkv, int D);
// (vra_embedding_bump, which bumps the forward's epoch word, is declared in vllm_rs_amd/csrc/gemm_launch.h)
// longest context the fused launch takes (tuning knob VRA_QKV_ATTN_MAX_CTX; 0 switches the fused launch off)
int vra_qkv_attn_max_ctx();
