// host_utils.h — small host-side numeric helpers shared by the runtime.
#pragma once
#include <stdint.h>
#include <string.h>

static inline uint16_t host_f32_to_bf16(float f) {  // RNE (same as csrc/common.cuh BF16::from_f32)
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline uint16_t host_f32_to_f16(float f) {
  _Float16 h = (_Float16)f;
  uint16_t r;
  memcpy(&r, &h, 2);
  return r;
}
