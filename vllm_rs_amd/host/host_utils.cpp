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

// RotaryEmbedding::new / ScalingRotaryEmbedding::new (src/models/layers/rotary_emb.rs:32-73,143-278):
// inv_freq = 1f32 / (theta^(i/d) in f64 → f32); linear: * (f32)(1/factor); llama3: wavelength
// smoothing in f32; freqs = pos(f32) * inv_freq; cos/sin in f32.
extern "C" void vra_rope_tables_f32(const vra_model_config* mc, int32_t n_pos, float* h_cos, float* h_sin) {
  const int rot = mc->head_dim, half = rot / 2;
  std::vector<float> inv(half);
  for (int i = 0; i < half; i++) inv[i] = 1.0f / (float)pow(mc->rope_theta, (double)(2 * i) / (double)rot);
  if (mc->rope_scaling_type == 1) {
    for (int i = 0; i < half; i++) inv[i] = inv[i] * (float)(1.0 / mc->rope_factor);
  } else if (mc->rope_scaling_type == 2) {
    const double omax = mc->rope_original_max_position > 0 ? mc->rope_original_max_position : mc->max_position_embeddings;
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
  }
  for (int p = 0; p < n_pos; p++)
    for (int i = 0; i < half; i++) {
      const float ang = (float)p * inv[i];
      h_cos[(size_t)p * half + i] = cosf(ang);
      h_sin[(size_t)p * half + i] = sinf(ang);
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
