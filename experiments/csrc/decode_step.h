// decode_step.h — host-visible description of "kernel P", the persistent decode step (decode_step.hip).
//
// One launch runs ALL decoder layers of a decode step for 1..2 sequences (llama.rs:107-131 op order per layer:
// norm -> q/k/v -> rope + cache + attention -> o_proj + residual -> norm -> gate/up -> SiLU*mul -> down + residual).
// One workgroup per CU: a LOADER wave streams this CU's share of every weight tensor, in program order and across phase
// boundaries, into an LDS ring by LDS-DMA; eight CONSUMER waves run the phases.  While a phase waits for its input (grid
// barrier, x round trip, norm) the ring keeps filling, so HBM never idles behind a dependency edge — which the
// launch-per-GEMV form cannot do.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define DP_NC 8                       // consumer waves (two per SIMD: a lone wave cannot issue fast enough to keep up with HBM)
#define DP_WAVES (DP_NC + 1)          // + the loader (wave 0)
#define DP_THREADS (DP_WAVES * 64)
#define DP_SLOT_W 16384               // 16 tiles of 1 KiB: k-tiles 16*ti .. 16*ti+15 of one unit (one contiguous run)
#define DP_SLOT_S 512                 // their scales (up to 16 groups x 16 columns x 2 bytes, unit-major)
#define DP_SLOT_Z 128                 // AWQ: their zero words (16 groups x 2 words)
#define DP_SLOT_BYTES (DP_SLOT_W + DP_SLOT_S + DP_SLOT_Z)
#define DP_MAX_SLOTS 8
#define DP_CTL_BYTES 512
#define DP_PHASES_PER_LAYER 5         // q/k/v | attention | o_proj | gate/up | down
#define DP_MAX_ROWS 2

// one decode GEMV (kernel E's contract, gemv_q4s.cuh: same units, same per-unit summation order => bit-identical outputs)
struct DPGemv {
  const void* w[2];          // tiled int4 words of stream 0 (and 1: the up tensor of a gate/up pair), unit-major
  const void* sc[2];         // unit-major scales [units][G][16]
  const uint32_t* zr[2];     // AWQ: unit-major zero words [units][G][2], else null
  const void* x;             // [M, x_ld]
  const void* norm_w;        // fused RMSNorm weight or null
  const void* residual;      // [M, res_ld] or null
  const void* bias[3];       // per output segment (pair: [0] gate, [1] up)
  void* out[3];
  int out_ld[3];
  int unit_start[3];         // first unit of segment 1, 2 (segment 0 starts at 0)
  int nseg;
  int x_ld, res_ld;
  int K, KT, TPW, gsh, G;    // KT = K/128, TPW = ceil(KT/16), scale group of k = k >> gsh, G = groups per unit
  int NS;                    // 1 | 2 (gate/up pair + SiLU*mul)
  int n_units, units_q, units_r, rot;  // workgroup b has rank (b + rot) % grid; rank r owns units_q (+1 if r < units_r) units from r*units_q + min(r, units_r)
  int sc_bytes, zr_bytes;    // bytes of one scale / zero stream (the last slot's DMA is clamped into it)
};
struct DPLayer {
  DPGemv g[4];               // 0 q/k/v, 1 o_proj, 2 gate/up, 3 down
  void* kc;                  // K cache of the layer [NB, Hkv, BS, D]
  void* vc;                  // V cache [NB, Hkv, D, BS]
};
struct DPStepArgs {
  const DPLayer* layers;     // device memory, built once
  int n_layers;
  int M;                     // sequences (rows), 1..DP_MAX_ROWS
  int ph0, ph1;              // phases [ph0, ph1) of the step (whole step: 0 .. 5*n_layers)
  float eps;
  // attention
  const void* q;             // [M, Hq, D]   (outputs of phase 0)
  const void* k;             // [M, Hkv, D]
  const void* v;
  void* attn;                // [M, Hq, D]
  const void* cosv;
  const void* sinv;
  const int64_t* positions;
  const int64_t* slots;
  const uint32_t* block_tables;
  const uint32_t* context_lens;
  int Hq, Hkv, BS, bs_shift, max_blocks;
  float scale_log2e;
  // LDS plan
  int nslot;                 // ring slots
  int ring_off, x_off, red_off;
  int xt;                    // bytes of the x region per k-tile: M rows x 272 + 16 (sums)
  int zero_off;              // an all-zero k-tile (x rows and sums): what a played wave without a k-tile multiplies, as in kernel E
  // grid synchronisation (device memory, zeroed once; monotonic: replayable from a hipGraph)
  uint32_t* counters;        // 8 shards x 16 words (one line each)
  uint32_t* count;           // phases completed by earlier launches
  uint32_t* err;             // device error word (a wait timed out)
  unsigned long long* ts;    // VRA_GEMV_TS builds: [grid][phases][8] stamps, then [grid][64 slots][4] of phase ts_phase
  int ts_phase;
  int dbg;                   // timing experiments (VRA_DP_DBG; results are garbage): 1 = the loader never waits for a free slot, 2 = consumers never wait for a landed one
};

struct DPPlan {
  int nslot, ring_off, x_off, red_off, xt, lds_bytes, zero_off;
};
bool vra_decode_step_init();  // per device, outside graph capture
// LDS plan for a model: max_kt = largest K/128 of a GEMV, max_red = largest (units per workgroup x NS) of a GEMV,
// group = q heads per kv head, D = head size; false = does not fit (the caller keeps the launch-per-op path)
bool vra_decode_step_plan(int M, int max_kt, int max_red, int group, int D, DPPlan* plan);
bool vra_decode_step_enabled();  // off unless VRA_DECODE_STEP=1 / vra_debug_set_decode_step(1)
int vra_decode_step_grid();   // workgroups of a launch (= CUs)
void vra_decode_step_sync_ptrs(uint32_t** counters, uint32_t** count, uint32_t** err);
uint32_t* vra_decode_step_error_word();  // device word (null before init): non-zero = a wait of some launch timed out
void vra_decode_step_reset();            // host, no launch in flight: clear the error word and the barrier state
void vra_launch_decode_step(const DPStepArgs& a, int dtype, bool awq, bool kv8, int head_dim, int64_t stream);
