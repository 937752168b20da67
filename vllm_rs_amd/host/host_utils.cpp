// host_utils.cpp — pure-host entry points of include/vllm_rs_amd.h §C: KV-cache sizing, rotary
// tables, marlin scale permutation.
#include "host_utils.h"

#include <math.h>

#include <vector>

#include "../../include/vllm_rs_amd.h"

// per_block_bytes (src/utils/kvcache_allocator.rs:447-468):
//   block_size * kv_heads_per_shard * head_dim * dtype_size * 2 (K and V) * num_layers
extern "C" int64_t vra_kv_per_block_bytes(const vra_model_config* mc, const vra_engine_config* ec) {
  const int world = ec->tp_world_size > 1 ? ec->tp_world_size : 1;
  const int hkv = mc->num_kv_heads >= world ? mc->num_kv_heads / world : 1;  // :127-141
  const int bs = ec->block_size > 0 ? ec->block_size : 64;
  const int dtype_size = ec->fp8_kvcache ? 1 : 2;  // kvcache_allocator.rs:188-193
  return (int64_t)bs * hkv * mc->head_dim * dtype_size * 2 * mc->num_layers;
}
// plan_allocation (:616-707): num_gpu_blocks = floor(free * kv_fraction / per_block_bytes);
// kv_fraction default 0.5, 0.95 when max_model_len is given (:196-202,311-315).
extern "C" int64_t vra_kv_plan_num_blocks(const vra_model_config* mc, const vra_engine_config* ec, int64_t free_bytes) {
  if (ec->num_gpu_blocks > 0) return ec->num_gpu_blocks;
  double frac = ec->kv_fraction > 0 ? ec->kv_fraction : (ec->max_model_len > 0 ? 0.95 : 0.5);
  const int64_t per = vra_kv_per_block_bytes(mc, ec);
  if (per <= 0) return 0;
  return (int64_t)((double)free_bytes * frac) / per;
}

// Rows of the cos / sin tables the reference builds (rotary_emb.rs:44-46 default / linear / llama3: max_position_embeddings;
// :296-312 dynamic: max_position_embeddings with `alpha`, else (u32)(original_max * factor); :519-526 yarn:
// (u32)(max_position_embeddings as f32 * factor)) — the longest position a sequence may reach.
extern "C" int32_t vra_rope_table_rows(const vra_model_config* mc) {
  if (mc->rope_scaling_type == 4) return (int32_t)(uint32_t)((float)mc->max_position_embeddings * (float)mc->rope_factor);
  if (mc->rope_scaling_type == 3 && !mc->rope_dynamic_alpha) {
    const double omax = mc->rope_original_max_position_f > 0.0 ? mc->rope_original_max_position_f
                                                                 : (mc->rope_original_max_position > 0 ? mc->rope_original_max_position : mc->max_position_embeddings);
    return (int32_t)(uint32_t)(omax * mc->rope_factor);
  }
  return mc->max_position_embeddings;
}

// RotaryEmbedding::new / ScalingRotaryEmbedding::new (src/models/layers/rotary_emb.rs:32-73,143-415,435-541):
// inv_freq = 1f32 / (theta^(i/d) in f64 -> f32); linear: * (f32)(1/factor); llama3: wavelength smoothing in f32; dynamic
// (NTK): the default table of a rescaled theta (f64); yarn: interpolation / extrapolation blend of two f32 frequency sets
// behind a linear ramp, cos / sin scaled by mscale; freqs = pos(f32) * inv_freq; cos/sin in f32.
extern "C" void vra_rope_tables_f32(const vra_model_config* mc, int32_t n_pos, float* h_cos, float* h_sin) {
  const int rot = mc->head_dim, half = rot / 2;
  std::vector<float> inv(half);
  const double omax = mc->rope_original_max_position_f > 0.0 ? mc->rope_original_max_position_f
                                                               : (mc->rope_original_max_position > 0 ? mc->rope_original_max_position : mc->max_position_embeddings);
  double theta = mc->rope_theta;
  float mscale = 1.0f;
  if (mc->rope_scaling_type == 3) {  // "dynamic" (rotary_emb.rs:281-333)
    const double f = mc->rope_factor;
    if (mc->rope_dynamic_alpha) {
      theta = pow(theta * f, (double)rot / (double)(rot - 2));
    } else {
      const uint32_t max_len = (uint32_t)(omax * f);
      theta = pow(theta * ((f * (double)max_len / omax) - (f - 1.0)), (double)rot / (double)(rot - 2));
    }
  }
  for (int i = 0; i < half; i++) inv[i] = 1.0f / (float)pow(theta, (double)(2 * i) / (double)rot);
  if (mc->rope_scaling_type == 1) {
    for (int i = 0; i < half; i++) inv[i] = inv[i] * (float)(1.0 / mc->rope_factor);
  } else if (mc->rope_scaling_type == 2) {
    const float low_wl = (float)(omax / mc->rope_low_freq_factor), high_wl = (float)(omax / mc->rope_high_freq_factor);
    for (int i = 0; i < half; i++) {
      const float freq = inv[i];
      const float wavelen = 2.0f * 3.14159265358979323846f / freq;
      if (wavelen < high_wl) {
      } else if (wavelen > low_wl) {
        inv[i] = freq / (float)mc->rope_factor;
      } else {
        const float smooth = ((float)omax / wavelen - (float)mc->rope_low_freq_factor) /
                             (float)(mc->rope_high_freq_factor - mc->rope_low_freq_factor);
        inv[i] = (1.0f - smooth) * freq / (float)mc->rope_factor + smooth * freq;
      }
    }
  } else if (mc->rope_scaling_type == 4) {  // "yarn" (YarnRotaryEmbedding::new_yarn, rotary_emb.rs:482-540): everything in f32
    const float base = (float)mc->rope_theta, factor = (float)mc->rope_factor;
    // (0 in a field = the reference's default, unless the caller marked the field as explicitly given: rope_yarn_explicit)
    const int ex = mc->rope_yarn_explicit;
    const float beta_fast = (ex & 1) || mc->rope_yarn_beta_fast != 0.0 ? (float)mc->rope_yarn_beta_fast : 32.0f;
    const float beta_slow = (ex & 2) || mc->rope_yarn_beta_slow != 0.0 ? (float)mc->rope_yarn_beta_slow : 1.0f;
    const float attn_factor = (ex & 4) || mc->rope_yarn_attn_factor != 0.0 ? (float)mc->rope_yarn_attn_factor : 1.0f;
    const float extrapolation = (ex & 8) || mc->rope_yarn_extrapolation_factor != 0.0 ? (float)mc->rope_yarn_extrapolation_factor : 1.0f;
    auto corr_dim = [&](float num_rot) {  // yarn_find_correction_dim
      return ((float)rot * logf((float)(size_t)omax / (num_rot * 2.0f * 3.14159265358979323846f))) / (2.0f * logf(base));
    };
    float low = floorf(corr_dim(beta_fast)), high = ceilf(corr_dim(beta_slow));
    low = fmaxf(low, 0.0f), high = fminf(high, (float)rot - 1.0f);
    if (low == high) high += 0.001f;
    const float ramp_mul = (float)(1.0 / ((double)high - (double)low));  // Tensor / f64 = affine(1/rhs): the reciprocal in f64, applied in f32
    for (int i = 0; i < half; i++) {
      const float p = powf(base, (float)(2 * i) / (float)rot);
      const float extra = 1.0f / p, inter = 1.0f / (factor * p);
      float ramp = ((float)i + (float)(-(double)low)) * ramp_mul;   // (arange - min) then / (max - min)
      ramp = fminf(fmaxf(ramp, 0.0f), 1.0f);
      const float mask = ((ramp * -1.0f) + 1.0f) * extrapolation;   // (1 - ramp) * extrapolation_factor
      inv[i] = inter * ((mask * -1.0f) + 1.0f) + extra * mask;
    }
    mscale = (factor <= 1.0f ? 1.0f : 0.1f * 1.0f * logf(factor) + 1.0f) * attn_factor;  // yarn_get_mscale(factor, 1.0) * attn_factor
  }
  for (int p = 0; p < n_pos; p++)
    for (int i = 0; i < half; i++) {
      const float ang = (float)p * inv[i];
      h_cos[(size_t)p * half + i] = mc->rope_scaling_type == 4 ? cosf(ang) * mscale : cosf(ang);
      h_sin[(size_t)p * half + i] = mc->rope_scaling_type == 4 ? sinf(ang) * mscale : sinf(ang);
    }
}

// marlin_permute_scales (src/models/layers/wna16.rs:180-218)
extern "C" void vra_marlin_permute_scales_u16(const uint16_t* in, uint16_t* out, int32_t rows, int32_t n, int32_t grouped) {
  const int64_t total = (int64_t)rows * n;
  if (grouped) {
    int perm[64];
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) perm[i * 8 + j] = i + 8 * j;
    for (int64_t c = 0; c < total / 64; c++)
      for (int t = 0; t < 64; t++) out[c * 64 + t] = in[c * 64 + perm[t]];
  } else {
    static const int base[8] = {0, 1, 8, 9, 16, 17, 24, 25};
    int perm[32];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) perm[i * 8 + j] = 2 * i + base[j];
    for (int64_t c = 0; c < total / 32; c++)
      for (int t = 0; t < 32; t++) out[c * 32 + t] = in[c * 32 + perm[t]];
  }
}
