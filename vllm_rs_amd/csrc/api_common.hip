// api_common.hip — error side channel, device plumbing, scratch (include/vllm_rs_amd.h §B tail).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.cuh"
#include "scratch.h"

static thread_local char g_err[512] = {0};
void vra_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  if (getenv("VRA_DEBUG")) fprintf(stderr, "[vllm_rs_amd] error: %s\n", g_err);
}
extern "C" const char* vra_last_error(void) { return g_err; }
extern "C" void vra_clear_error(void) { g_err[0] = 0; }
extern "C" int32_t vra_take_device_error(void) { return vra_scratch_take_error(); }
extern "C" const char* vra_version(void) { return "vllm_rs_amd 0.1.0 (gfx950, hip)"; }

static int hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    vra_set_error("%s: %s", what, hipGetErrorString(e));
    return -1;
  }
  return 0;
}
extern "C" int32_t vra_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int32_t vra_set_device(int32_t d) { return hip_ok(hipSetDevice(d), "hipSetDevice"); }
extern "C" void* vra_malloc(size_t bytes) {
  void* p = nullptr;
  if (hip_ok(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) return nullptr;
  return p;
}
extern "C" void vra_free(void* p) {
  if (p) (void)hipFree(p);
}
extern "C" void* vra_malloc_host(size_t bytes) {
  void* p = nullptr;
  if (hip_ok(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
  return p;
}
extern "C" void vra_free_host(void* p) {
  if (p) (void)hipHostFree(p);
}
extern "C" int32_t vra_memcpy_h2d(void* dst, const void* src, size_t bytes, int64_t stream) {
  return hip_ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)), "memcpy h2d");
}
extern "C" int32_t vra_memcpy_d2h(void* dst, const void* src, size_t bytes, int64_t stream) {
  return hip_ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)), "memcpy d2h");
}
extern "C" int32_t vra_memcpy_d2d(void* dst, const void* src, size_t bytes, int64_t stream) {
  return hip_ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)), "memcpy d2d");
}
extern "C" int32_t vra_memset(void* dst, int32_t value, size_t bytes, int64_t stream) {
  return hip_ok(hipMemsetAsync(dst, value, bytes, as_stream(stream)), "memset");
}
extern "C" int32_t vra_stream_sync(int64_t stream) { return hip_ok(hipStreamSynchronize(as_stream(stream)), "stream sync"); }
extern "C" int32_t vra_device_sync(void) { return hip_ok(hipDeviceSynchronize(), "device sync"); }
extern "C" int64_t vra_stream_create(void) {
  hipStream_t s = nullptr;
  if (hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream create")) return 0;
  return reinterpret_cast<int64_t>(s);
}
extern "C" void vra_stream_destroy(int64_t s) {
  if (s) (void)hipStreamDestroy(as_stream(s));
}
extern "C" int32_t vra_mem_info(size_t* h_free, size_t* h_total) { return hip_ok(hipMemGetInfo(h_free, h_total), "mem info"); }
extern "C" void* vra_event_create(void) {
  hipEvent_t e = nullptr;
  if (hip_ok(hipEventCreate(&e), "event create")) return nullptr;
  return e;
}
extern "C" void vra_event_destroy(void* e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}
extern "C" int32_t vra_event_record(void* e, int64_t stream) { return hip_ok(hipEventRecord((hipEvent_t)e, as_stream(stream)), "event record"); }
extern "C" float vra_event_elapsed_ms(void* a, void* b) {
  float ms = -1.f;
  if (hip_ok(hipEventSynchronize((hipEvent_t)b), "event sync")) return -1.f;
  if (hip_ok(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b), "event elapsed")) return -1.f;
  return ms;
}

// ---------------------------------------------------------------- scratch
// One scratch set PER DEVICE (keyed by the device current at the call, like every launcher that
// consumes it): a second engine on another device of the same process gets its own slabs, flags and
// error word.  Within one device the reference's threading contract applies: one forward at a time
// per process (engine.rs:844 holds `runners.write()`), so one set per device is enough; two streams
// of ONE device running split-K GEMMs concurrently are outside that contract.
static const size_t kSlabBytes = (size_t)192 << 20;  // fp32 split-K partials
static const size_t kCounters = 1 << 16;
static const size_t kScaleBytes = (size_t)8 << 20;  // per region; two regions (gate/up)
static const int kMaxDevices = 64;
struct ScratchSet {
  float* slabs = nullptr;
  unsigned char* scales = nullptr;
  uint32_t* counters = nullptr;
};
static ScratchSet g_scratch[kMaxDevices];
static std::mutex g_scratch_mu;
static ScratchSet* scratch_for_current_device(bool create) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  ScratchSet& s = g_scratch[dev];
  if (s.slabs && s.counters && s.scales) return &s;
  if (!create) return nullptr;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (s.slabs && s.counters && s.scales) return &s;
  void* p = nullptr;
  if (!s.slabs) {
    if (hipMalloc(&p, kSlabBytes) != hipSuccess) return nullptr;
    {
      const char* poison = getenv("VRA_POISON_ALLOC");  // debugging aid (host/model.cpp): slabs start as NaN instead of whatever was there
      if (poison && poison[0] == '1') (void)hipMemset(p, 0xFF, kSlabBytes);
    }
    s.slabs = (float*)p;
  }
  if (!s.counters) {
    if (hipMalloc(&p, (kCounters + 16) * sizeof(uint32_t)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, (kCounters + 16) * sizeof(uint32_t)) != hipSuccess) return nullptr;
    s.counters = (uint32_t*)p;
  }
  if (!s.scales) {
    if (hipMalloc(&p, 2 * kScaleBytes) != hipSuccess) return nullptr;
    s.scales = (unsigned char*)p;
  }
  return &s;
}
bool vra_scratch_init() { return scratch_for_current_device(true) != nullptr; }
float* vra_scratch_slabs() {
  ScratchSet* s = scratch_for_current_device(true);
  return s ? s->slabs : nullptr;
}
uint32_t* vra_scratch_counters() {
  ScratchSet* s = scratch_for_current_device(true);
  return s ? s->counters : nullptr;
}
void* vra_scratch_scales(int which) {
  ScratchSet* s = scratch_for_current_device(true);
  return s ? s->scales + (size_t)(which & 1) * kScaleBytes : nullptr;
}
size_t vra_scratch_scale_bytes() { return kScaleBytes; }
size_t vra_scratch_slab_bytes() { return kSlabBytes; }
size_t vra_scratch_counter_count() { return kCounters; }
uint32_t* vra_scratch_error_word() {
  ScratchSet* s = scratch_for_current_device(true);
  return s ? s->counters + kCounters : nullptr;
}
int vra_scratch_take_error() {
  ScratchSet* s = scratch_for_current_device(false);
  if (!s) return 0;
  uint32_t v = 0;
  if (hipMemcpy(&v, s->counters + kCounters, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  // a wait that gave up leaves the exchange state of that launch undefined: the owner cleared a flag it never saw raised and the
  // late slice may still raise it afterwards — the next launch would then consume stale slabs without waiting.  The
  // device is idle here (the copy above synchronised): every flag goes back to zero together with the error word.
  if (v) (void)hipMemset(s->counters, 0, (kCounters + 1) * sizeof(uint32_t));
  return v != 0;
}
// the same for callers that read the error word with their own copy (host/engine.cpp): queue the reset of flags + error word
void vra_scratch_reset_after_error(hipStream_t st) {
  ScratchSet* s = scratch_for_current_device(false);
  if (s) (void)hipMemsetAsync(s->counters, 0, (kCounters + 1) * sizeof(uint32_t), st);
}
