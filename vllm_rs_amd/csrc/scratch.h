// scratch.h — per-process device scratch for split-K slabs / arrival counters (allocated once,
// outside any graph capture, by vra_scratch_init(); kernels never allocate).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
bool vra_scratch_init();            // idempotent; false if allocation failed or stream is capturing
float* vra_scratch_slabs();         // nullptr until initialised
uint32_t* vra_scratch_counters();   // zeroed at init, every kernel leaves them zero
size_t vra_scratch_slab_bytes();
size_t vra_scratch_counter_count();  // flag words usable by kernels (one more word behind them is the error word)
uint32_t* vra_scratch_error_word();  // device word set by a kernel whose split-K wait timed out (a lost slice)
int vra_scratch_take_error();        // host: read and clear that word (synchronises the device); 1 = a wait timed out (all flags are re-zeroed then)
void vra_scratch_reset_after_error(hipStream_t st);  // flags + error word back to zero (a timed-out exchange leaves them undefined)
void* vra_scratch_scales(int which);  // two regions for row-major copies of Marlin-permuted scale tensors
size_t vra_scratch_scale_bytes();
