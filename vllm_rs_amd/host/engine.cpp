// engine.cpp — ModelRunner + LLMEngine loop + C API (include/vllm_rs_amd.h §C).
//   ModelRunner::{prepare_prefill, prepare_decode, prepare_block_tables, run, sample}
//     src/core/runner.rs:743-896, 952-1388, 1390-1570 (greedy only)
//   GraphCapturer::{capture, replay} src/utils/graph.rs:448-834 → hipGraph, lazily per (batch bucket,
//     context bucket); padded lanes use slot -1 / context 0 (fix of Appendix A6)
//   LLMEngine three-phase step src/core/engine.rs:812-1128, TTFT per :1004-1012
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <thread>
#include <memory>

#include "../csrc/scratch.h"
#include "core.h"
#include "model.h"

// " [slice j, peer r: expected epoch e, flag read f]" of the first one-shot wait that gave up (comm.hip)
static std::string comm_timeout_detail(void* comm) {
  uint32_t d[4] = {0, 0, 0, 0};
  if (!comm || vra_comm_error_detail(comm, d) != 0) return "";
  return " [slice " + std::to_string(d[0]) + ", waiting for rank " + std::to_string(d[1]) + ": expected epoch " + std::to_string(d[2]) +
         ", its flag read " + std::to_string(d[3]) + "]";
}

// Optional profiler ranges (SURVEY §5 aux: the reference marks its prefill / decode steps for nsys; here roctx, shown by
// `rocprofv3 --marker-trace`): VRA_ROCTX=1 loads the roctx library at run time — no link-time dependency, nothing on the hot path otherwise.
#include <dlfcn.h>
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* on = getenv("VRA_ROCTX");
    if (!on || !atoi(on)) return;
    // (the rocprofiler-sdk library is the one rocprofv3 intercepts; roctracer's libroctx64 for older tools)
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
struct RoctxRange {
  static Roctx& api() {
    static Roctx r;
    return r;
  }
  bool on;
  RoctxRange(const char* what, int tokens, int seqs) : on(api().push != nullptr) {
    if (!on) return;
    char buf[96];
    snprintf(buf, sizeof buf, "vra %s tokens=%d seqs=%d", what, tokens, seqs);
    api().push(buf);
  }
  ~RoctxRange() {
    if (on) api().pop();
  }
};
}  // namespace

namespace vra {

static inline uint32_t vra_hash32_host(uint64_t seed, uint64_t idx) {  // == vra_hash32 of csrc/common.cuh
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

struct RequestResult {
  std::vector<uint32_t> output;
  double created_ms = 0, first_token_ms = 0, finished_ms = 0;
  bool finished = false, aborted = false;
};

class Engine {
 public:
  Engine(const vra_model_config& mc, const vra_engine_config& ec) : mc_(mc), ec_(ec), model_(mc, ec) {
    if (ec_.block_size <= 0) ec_.block_size = 64;
    if (ec_.prefill_chunk <= 0) ec_.prefill_chunk = 8192;
    // the staging buffers and activations hold kMaxStepTokens rows: a larger chunk is clamped HERE so that the scheduler
    // (chunking, A13) and prepare_prefill use the same value
    if (ec_.prefill_chunk > kMaxStepTokens) ec_.prefill_chunk = kMaxStepTokens;
  }
  static constexpr int kMaxStepTokens = 16384;
  static constexpr int kMinScheduledReqs = 5;  // scheduler.rs:44 — the scheduler batches up to max(max_num_seqs, 5) sequences
  ~Engine() {
    if (ec_.device < 0) {  // host-only engine: nothing was allocated through HIP
      free(h_meta_);
      return;
    }
    for (auto& g : graphs_) (void)hipGraphExecDestroy(g.second);
    if (h_meta_) (void)hipHostFree(h_meta_);
    if (d_meta_) (void)hipFree(d_meta_);
    if (h_tokens_) (void)hipHostFree(h_tokens_);
    if (d_tokens_) (void)hipFree(d_tokens_);
    if (h_err_) (void)hipHostFree(h_err_);
    if (h_pen_ctx_) (void)hipHostFree(h_pen_ctx_);
    if (d_pen_) (void)hipFree(d_pen_);
    if (stream_) (void)hipStreamDestroy(stream_);
    // comm_ is caller-owned (vra_engine_set_comm): like every other buffer the caller hands over, it is not destroyed here
  }
  std::string error;
  vra_model_config mc_;
  vra_engine_config ec_;
  Model model_;
  std::unique_ptr<BlockManager> bm_;
  std::unique_ptr<Scheduler> sched_;
  std::map<int64_t, RequestResult> results_;
  hipStream_t stream_ = nullptr;
  void* comm_ = nullptr;
  int max_seqs_ = 0, max_model_len_ = 0, max_blocks_per_seq_ = 0, max_step_tokens_ = 0;

  // packed metadata staging: one pinned buffer, one device buffer, one H2D copy per step
  uint8_t* h_meta_ = nullptr;
  uint8_t* d_meta_ = nullptr;
  size_t meta_bytes_ = 0;
  size_t off_ids_, off_pos_, off_slots_, off_bt_, off_ctx_, off_cuq_, off_last_, off_dids_, off_dpos_, off_dslots_;
  uint32_t* d_tokens_ = nullptr;
  uint32_t* h_tokens_ = nullptr;
  uint32_t* h_err_ = nullptr;  // pinned copy of the device error words (split-K exchange, one-shot all-reduce), read every step
  std::map<int64_t, hipGraphExec_t> graphs_;
  hipGraphExec_t last_graph_ = nullptr;  // the decode graph of the most recent step (vra_engine_bench_replay)
  bool prepared_ = false;
  int64_t planned_blocks_ = 0;
  // ---- sampling (ModelRunner::sample, runner.rs:1390-1570): strategy cached at prefill from the first sequence (A3)
  struct CachedSampling {
    int kind = 0;  // 0 ArgMax, 1 stochastic (All / TopK / TopP / TopKThenTopP by k and p)
    int k = 0;
    float p = -1.f, temperature = 1.f;
    bool has_freq = false, has_pres = false;
    float freq = 0.f, pres = 0.f;
  } cached_sampling_;
  uint64_t sample_calls_ = 0;
  uint32_t* h_pen_ctx_ = nullptr;  // pinned [max_seqs, 128] tokens + [max_seqs] lengths + 2 x [max_seqs] penalties
  uint8_t* d_pen_ = nullptr;
  static constexpr int kPenaltyWindow = 128;

  bool fail(const std::string& m) {
    error = m;
    return false;
  }

  // host-only engine (device = -1): scheduler, block manager and metadata staging exactly as in the real
  // engine, the forward pass replaced by caller-supplied tokens (vra_engine_dry_schedule / _dry_commit)
  bool dry() const { return ec_.device < 0; }
  std::vector<int> pend_ids_;
  bool pend_prefill_ = false, pend_valid_ = false;
  int num_cpu_blocks_ = 0;
  std::vector<void*> h_swap_k_, h_swap_v_;
  int64_t swap_out_blocks_ = 0, swap_in_blocks_ = 0;  // statistics (vra_engine_swap_stats)
  // the block copies the scheduler decided, on the engine stream, before the forward that follows (cache::swap_blocks,
  // runner.rs:1641-1645).  A host-only engine only counts them.
  bool execute_swaps() {
    for (auto& op : sched_->take_swap_ops()) {
      (op.to_gpu ? swap_in_blocks_ : swap_out_blocks_) += (int64_t)op.pairs.size();
      if (dry() || op.pairs.empty()) continue;
      std::vector<int64_t> pairs;
      for (auto& pr : op.pairs) pairs.push_back(pr.first), pairs.push_back(pr.second);
      const int64_t bb = (int64_t)model_.kv_block_bytes();
      for (int l = 0; l < mc_.num_layers; l++) {
        if (op.to_gpu) {
          vra_swap_blocks(h_swap_k_[l], model_.k_cache(l), pairs.data(), (int)op.pairs.size(), bb, 2, (int64_t)stream_);
          vra_swap_blocks(h_swap_v_[l], model_.v_cache(l), pairs.data(), (int)op.pairs.size(), bb, 2, (int64_t)stream_);
        } else {
          vra_swap_blocks(model_.k_cache(l), h_swap_k_[l], pairs.data(), (int)op.pairs.size(), bb, 1, (int64_t)stream_);
          vra_swap_blocks(model_.v_cache(l), h_swap_v_[l], pairs.data(), (int)op.pairs.size(), bb, 1, (int64_t)stream_);
        }
      }
      const char* e = vra_last_error();
      if (e && e[0]) return fail(std::string("swap: ") + e);
    }
    return true;
  }
  // nothing was scheduled: drop a sequence if that is the only way forward; a swapped-out sequence inside its cooling
  // period is progress of its own (the caller's step loop polls)
  void nothing_scheduled() {
    collect();
    if (!sched_->has_unfinished()) return;
    if (sched_->only_swapped_left()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
      return;
    }
    sched_->abort_one(now_ms());
    collect();
  }
  InputMetadata pend_md_;
  bool finalize_dry() {
    if (ec_.num_gpu_blocks < 2) return fail("dry engine needs an explicit num_gpu_blocks");
    max_model_len_ = ec_.max_model_len > 0 ? ec_.max_model_len : mc_.max_position_embeddings;
    if (max_model_len_ > vra_rope_table_rows(&mc_)) max_model_len_ = vra_rope_table_rows(&mc_);  // the rotary table's rows (yarn / dynamic: > max_position_embeddings)
    max_seqs_ = std::max(ec_.max_num_seqs > 0 ? ec_.max_num_seqs : 32, kMinScheduledReqs);
    max_step_tokens_ = kMaxStepTokens;
    const int64_t nb = ec_.num_gpu_blocks;
    setup_host(nb);
    h_meta_ = (unsigned char*)calloc(1, meta_bytes_);
    d_meta_ = h_meta_;  // bind() hands out pointers into the (host) staging buffer
    return h_meta_ != nullptr;
  }
  void setup_host(int64_t nb) {
    max_blocks_per_seq_ = (max_model_len_ + ec_.block_size - 1) / ec_.block_size;
    // CPU swap space (kvcache_allocator.rs:673): num_cpu_blocks = gpu blocks x cpu_mem_fold
    num_cpu_blocks_ = ec_.cpu_mem_fold > 0.f ? (int)((double)nb * ec_.cpu_mem_fold) : 0;
    bm_.reset(new BlockManager((int)nb, ec_.block_size, ec_.enable_prefix_cache != 0, ec_.prefix_cache_fraction, num_cpu_blocks_));
    SchedulerConfig sc;
    if (ec_.swap_cooling_ms) sc.swap_cooling_ms = std::max(0, ec_.swap_cooling_ms);
    if (ec_.min_tokens_left_for_swap) sc.min_tokens_left_for_swap = std::max(0, ec_.min_tokens_left_for_swap);
    sc.max_num_seqs = ec_.max_num_seqs > 0 ? ec_.max_num_seqs : 32;  // the scheduler applies its own floor of 5 (scheduler.rs:44)
    sc.max_num_batched_tokens = (int)std::min<int64_t>(nb * ec_.block_size, 1 << 30);
    sc.block_size = ec_.block_size;
    sc.prefill_chunk = ec_.prefill_chunk;
    sc.max_step_tokens = max_step_tokens_;
    sc.max_model_len = max_model_len_;
    sched_.reset(new Scheduler(bm_.get(), sc));
    // ---- metadata staging: the per-sequence arrays and the block tables come first, so that a decode step uploads only
    // [0, off_bt_ + n_seqs * stride * 4); the token-sized arrays of prefill steps follow
    const size_t T = std::max(max_step_tokens_, max_seqs_), B = max_seqs_;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    off_ctx_ = o, o += al(B * 4);
    off_cuq_ = o, o += al((B + 1) * 4);
    off_last_ = o, o += al(B * 4);
    off_dids_ = o, o += al(B * 4);
    off_dpos_ = o, o += al(B * 8);
    off_dslots_ = o, o += al(B * 8);
    off_bt_ = o, o += al(B * (size_t)max_blocks_per_seq_ * 4);
    off_ids_ = o, o += al(T * 4);
    off_pos_ = o, o += al(T * 8);
    off_slots_ = o, o += al(T * 8);
    meta_bytes_ = o;
  }

  // weights are in place: activation buffers, then the KV plan of THIS rank (KVCacheAllocator::plan_allocation,
  // kvcache_allocator.rs:564-707).  Under tensor parallelism every rank must end up with the SAME block count (the
  // schedulers run in lock step): the launcher takes the minimum over the ranks' plans and sets it before finalize —
  // the reference's engine process does the same through MessageType::UsableMemoryLeft (runner/mod.rs:277).
  bool prepare() {
    if (prepared_) return true;
    if (hipSetDevice(ec_.device) != hipSuccess) return fail("hipSetDevice failed");
    if (!stream_ && hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return fail("stream create failed");
    max_model_len_ = ec_.max_model_len > 0 ? ec_.max_model_len : mc_.max_position_embeddings;
    if (max_model_len_ > vra_rope_table_rows(&mc_)) max_model_len_ = vra_rope_table_rows(&mc_);  // the rotary table's rows (yarn / dynamic: > max_position_embeddings)
    // every per-sequence buffer is sized for what the scheduler may batch: max(max_num_seqs, 5)
    max_seqs_ = std::max(ec_.max_num_seqs > 0 ? ec_.max_num_seqs : 32, kMinScheduledReqs);
    max_step_tokens_ = kMaxStepTokens;  // a prefill step batches prompts up to this many tokens (1.2 GB of activations for Llama-3-8B)
    // reserve activations first, then give kv_fraction of what is left to the cache
    if (!model_.init_buffers(std::max(max_step_tokens_, max_seqs_), max_seqs_)) return fail("activation buffers: " + model_.error);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    planned_blocks_ = vra_kv_plan_num_blocks(&mc_, &ec_, (int64_t)free_b);
    if (planned_blocks_ > (1 << 24)) planned_blocks_ = 1 << 24;
    prepared_ = true;
    return true;
  }
  bool finalize() {
    if (dry()) return finalize_dry();
    if (model_.world() > 1 && !model_.has_comm()) return fail("tensor parallel world_size > 1 without a communicator (vra_engine_set_comm before finalize)");
    if (model_.world() > 1 && ec_.num_gpu_blocks <= 0)
      return fail("tensor parallel: num_gpu_blocks must be set explicitly and identically on every rank (vra_engine_plan_kv_blocks on each rank, take the minimum, vra_engine_set_num_gpu_blocks)");
    if (!prepare()) return false;
    int64_t nb = ec_.num_gpu_blocks > 0 ? ec_.num_gpu_blocks : planned_blocks_;
    if (nb < 2) return fail("not enough memory for the KV cache");
    if (nb > (1 << 24)) nb = 1 << 24;
    if (!model_.init_kv_cache((int)nb)) return fail("kv cache: " + model_.error);
    setup_host(nb);
    if (num_cpu_blocks_ > 0) {  // pinned host copies of whole blocks, per layer: K then V (kvcache_allocator.rs:905-930)
      const size_t bytes = (size_t)num_cpu_blocks_ * model_.kv_block_bytes();
      h_swap_k_.assign(mc_.num_layers, nullptr);
      h_swap_v_.assign(mc_.num_layers, nullptr);
      for (int l = 0; l < mc_.num_layers; l++)
        if (hipHostMalloc(&h_swap_k_[l], bytes, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&h_swap_v_[l], bytes, hipHostMallocDefault) != hipSuccess)
          return fail("pinned alloc for the CPU swap space failed");
    }
    const size_t B = max_seqs_;
    if (hipHostMalloc((void**)&h_meta_, meta_bytes_, hipHostMallocDefault) != hipSuccess) return fail("pinned alloc failed");
    if (hipMalloc((void**)&d_meta_, meta_bytes_) != hipSuccess) return fail("meta alloc failed");
    if (hipMalloc((void**)&d_tokens_, B * 4) != hipSuccess) return fail("token alloc failed");
    if (hipHostMalloc((void**)&h_tokens_, B * 4, hipHostMallocDefault) != hipSuccess) return fail("pinned alloc failed");
    if (hipHostMalloc((void**)&h_err_, 64, hipHostMallocDefault) != hipSuccess) return fail("pinned alloc failed");
    const size_t pen_bytes = B * (kPenaltyWindow * 4 + 4 + 4 + 4);
    if (hipHostMalloc((void**)&h_pen_ctx_, pen_bytes, hipHostMallocDefault) != hipSuccess) return fail("pinned alloc failed");
    if (hipMalloc((void**)&d_pen_, pen_bytes) != hipSuccess) return fail("penalty staging alloc failed");
    memset(h_meta_, 0, meta_bytes_);
    memset(h_err_, 0, 64);
    if (ec_.use_graph && !warmup_capture()) return false;
    return true;
  }

  // ---- ModelRunner::prepare_prefill (runner.rs:978-1241)
  InputMetadata prepare_prefill(const std::vector<int>& ids, int* n_copy_bytes) {
    auto& run = sched_->running();
    uint32_t* h_ids = (uint32_t*)(h_meta_ + off_ids_);
    int64_t* h_pos = (int64_t*)(h_meta_ + off_pos_);
    int64_t* h_slots = (int64_t*)(h_meta_ + off_slots_);
    uint32_t* h_bt = (uint32_t*)(h_meta_ + off_bt_);
    uint32_t* h_ctx = (uint32_t*)(h_meta_ + off_ctx_);
    uint32_t* h_cuq = (uint32_t*)(h_meta_ + off_cuq_);
    uint32_t* h_last = (uint32_t*)(h_meta_ + off_last_);
    const int BS = ec_.block_size, CHUNK = ec_.prefill_chunk;
    int T = 0, max_q = 0, max_ctx = 0, max_bt = 0;
    InputMetadata md;
    {  // the token count is checked BEFORE anything is written to the staging buffers
      int64_t total = 0;
      for (int id : ids) total += std::min(CHUNK, run[id].len() - run[id].num_cached_tokens);
      if (total > std::max(max_step_tokens_, max_seqs_) || (int)ids.size() > max_seqs_) {
        md.n_tokens = -1;
        return md;
      }
    }
    for (int id : ids) max_bt = std::max(max_bt, (int)run[id].block_table.size());
    h_cuq[0] = 0;
    for (size_t b = 0; b < ids.size(); b++) {
      const Sequence& s = run[ids[b]];
      const int n = std::min(CHUNK, s.len() - s.num_cached_tokens);
      for (int i = 0; i < n; i++) {
        h_ids[T + i] = s.token_ids[s.num_cached_tokens + i];
        h_pos[T + i] = s.num_cached_tokens + i;
      }
      // slot mapping walks blocks from num_cached_blocks() with the in-block offset
      // num_cached_tokens % BS on the first one (runner.rs:1020-1038)
      int done = 0;
      for (int i = s.num_cached_blocks(); i < s.num_blocks() && done < n; i++) {
        int64_t start = (int64_t)s.block_table[i] * BS;
        int room = BS;
        if (i == s.num_cached_blocks()) {
          start += s.num_cached_tokens % BS;
          room = BS - s.num_cached_tokens % BS;
        }
        const int take = std::min(n - done, room);
        for (int j = 0; j < take; j++) h_slots[T + done + j] = start + j;
        done += take;
      }
      h_ctx[b] = (uint32_t)(s.num_cached_tokens + n);
      for (int j = 0; j < max_bt; j++) h_bt[b * (size_t)max_bt + j] = j < (int)s.block_table.size() ? s.block_table[j] : 0;
      T += n;
      h_cuq[b + 1] = (uint32_t)T;
      h_last[b] = (uint32_t)(T - 1);
      max_q = std::max(max_q, n);
      max_ctx = std::max(max_ctx, s.num_cached_tokens + n);
    }
    md.is_prefill = true;
    md.n_tokens = T;
    md.n_seqs = (int)ids.size();
    md.max_blocks = max_bt;
    md.max_seqlen_q = max_q;
    md.max_context_len = max_ctx;
    bind(md);
    *n_copy_bytes = (int)meta_bytes_;
    return md;
  }
  // ---- ModelRunner::prepare_decode (runner.rs:1243-1388); `bucket` >= batch pads the static buffers
  InputMetadata prepare_decode(const std::vector<int>& ids, int bucket) {
    auto& run = sched_->running();
    uint32_t* h_ids = (uint32_t*)(h_meta_ + off_dids_);
    int64_t* h_pos = (int64_t*)(h_meta_ + off_dpos_);
    int64_t* h_slots = (int64_t*)(h_meta_ + off_dslots_);
    uint32_t* h_bt = (uint32_t*)(h_meta_ + off_bt_);
    uint32_t* h_ctx = (uint32_t*)(h_meta_ + off_ctx_);
    const int BS = ec_.block_size, stride = max_blocks_per_seq_;
    int max_ctx = 0;
    for (int b = 0; b < bucket; b++) {
      if (b < (int)ids.size()) {
        const Sequence& s = run[ids[b]];
        h_ids[b] = s.last_token;
        h_pos[b] = s.len() - 1;
        h_ctx[b] = (uint32_t)s.len();
        // slot = block_table.last()*BS + last_block_tokens - 1 (runner.rs:1259-1262)
        h_slots[b] = (int64_t)s.block_table.back() * BS + s.last_block_num_tokens() - 1;
        for (size_t j = 0; j < s.block_table.size(); j++) h_bt[(size_t)b * stride + j] = s.block_table[j];
        max_ctx = std::max(max_ctx, s.len());
      } else {  // padded lane of a graph bucket: writes nothing, attends to nothing (Appendix A6 fix)
        h_ids[b] = 0;
        h_pos[b] = 0;
        h_ctx[b] = 0;
        h_slots[b] = -1;
      }
    }
    InputMetadata md;
    md.is_prefill = false;
    md.n_tokens = bucket;
    md.n_seqs = bucket;
    md.max_blocks = stride;
    md.max_seqlen_q = 1;
    md.max_context_len = max_ctx;
    bind(md);
    return md;
  }
  void bind(InputMetadata& md) {  // decode steps read the compact per-sequence arrays, prefill steps the token-sized ones
    md.input_ids = (const uint32_t*)(d_meta_ + (md.is_prefill ? off_ids_ : off_dids_));
    md.positions = (const int64_t*)(d_meta_ + (md.is_prefill ? off_pos_ : off_dpos_));
    md.slot_mapping = (const int64_t*)(d_meta_ + (md.is_prefill ? off_slots_ : off_dslots_));
    md.block_tables = (const uint32_t*)(d_meta_ + off_bt_);
    md.context_lens = (const uint32_t*)(d_meta_ + off_ctx_);
    md.cu_seqlens_q = (const uint32_t*)(d_meta_ + off_cuq_);
    md.last_token_rows = (const uint32_t*)(d_meta_ + off_last_);
  }
  // one H2D copy per step (the reference does five, runner.rs:1222-1238): a decode step only needs the head of the staging
  // buffer (per-sequence arrays + the block-table rows in use), a prefill step everything up to its last token
  bool upload_meta(const InputMetadata& md) {
    size_t bytes = off_bt_ + (size_t)md.n_seqs * (md.is_prefill ? md.max_blocks : max_blocks_per_seq_) * 4;
    if (md.is_prefill) bytes = off_slots_ + (size_t)md.n_tokens * 8;
    bytes = std::min(meta_bytes_, (bytes + 255) & ~(size_t)255);
    return hipMemcpyAsync(d_meta_, h_meta_, bytes, hipMemcpyHostToDevice, stream_) == hipSuccess;
  }

  static int batch_bucket(int n) {  // planned_graph_capture_batches (graph.rs:370-377): {1..15, 16, 32}; beyond 32: powers of two
    if (n <= 15) return n;
    int b = 16;
    while (b < n) b *= 2;
    return b;
  }
  // context buckets 256, then 512 * 4^k: the split-KV decision of the decode attention is taken for the bucket, and below
  // 256 tokens splitting (plus its merge launch) is a loss at every batch size (bs = 32 at ctx ~200: 3.61 ms per step with
  // the 512 bucket's two splits, 3.49 ms eager with one)
  static int ctx_bucket(int c) {
    if (c <= 256) return 256;
    int b = 512;
    while (b < c) b *= 4;
    return b;
  }
  static int next_ctx_bucket(int cb) { return cb < 512 ? 512 : cb * 4; }

  // ---- GraphCapturer::capture (graph.rs:267-308, 448-560): the decode forward of `bucket` lanes with contexts up to `cb`,
  // static buffers, relaxed mode.  Returns null (error left empty) when capture is unavailable: the caller runs eagerly.
  hipGraphExec_t graph_for(int bucket, int cb) {
    const int64_t key = ((int64_t)bucket << 32) | (uint32_t)cb;
    auto it = graphs_.find(key);
    if (it != graphs_.end()) return it->second;
    InputMetadata cmd;
    cmd.is_prefill = false;
    cmd.n_tokens = cmd.n_seqs = bucket;
    cmd.max_blocks = max_blocks_per_seq_;
    cmd.max_seqlen_q = 1;
    cmd.max_context_len = cb;
    bind(cmd);
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed) != hipSuccess) return nullptr;
    const bool ok = model_.forward(cmd, (int64_t)stream_, d_tokens_);
    const hipError_t e = hipStreamEndCapture(stream_, &g);
    if (ok && e == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) graphs_[key] = ge;
    else ge = nullptr;
    if (g) (void)hipGraphDestroy(g);
    if (!ok) fail("graph capture: " + model_.error);
    return ge;
  }
  // warmup_capture (engine.rs:108-503 → graph.rs:448-560): all decode graphs are captured at init, so that no request
  // pays a capture inside its TTFT.  Batch buckets as the reference plans them; one graph per context bucket (the
  // split-KV decision of the decode attention depends on it).
  bool warmup_capture() {
    std::vector<int> bs;
    for (int b = 1; b <= std::min(max_seqs_, 15); b++) bs.push_back(b);
    for (int b = 16; b <= max_seqs_; b *= 2) bs.push_back(b);
    for (int b : bs)
      for (int cb = 256;; cb = next_ctx_bucket(cb)) {
        if (!graph_for(b, cb) && !error.empty()) return false;
        if (cb >= max_model_len_) break;
      }
    return true;
  }

  // ---- ModelRunner::run (runner.rs:743-896) + sample (argmax, logits_processor.rs:67-70)
  bool run(const std::vector<int>& ids, bool is_prefill, std::vector<uint32_t>* tokens) {
    const int B = (int)ids.size();
    if (is_prefill) {
      int nb = 0;
      InputMetadata md = prepare_prefill(ids, &nb);
      if (md.n_tokens < 0) return fail("prefill step exceeds max_step_tokens / max_num_seqs");
      if (!upload_meta(md)) return fail("metadata upload failed");
      last_graph_ = nullptr;  // (vra_engine_bench_replay replays the decode graph of the MOST RECENT step only: ADVICE r4)
      RoctxRange range("prefill", md.n_tokens, B);
      if (!model_.forward(md, (int64_t)stream_, d_tokens_)) return fail(model_.error);
    } else {
      const int bucket = std::min(batch_bucket(B), max_seqs_);
      InputMetadata md = prepare_decode(ids, std::max(bucket, B));
      if (!upload_meta(md)) return fail("metadata upload failed");
      RoctxRange range("decode", md.n_tokens, B);
      bool launched = false;
      if (ec_.use_graph) {
        const int cb = ctx_bucket(md.max_context_len);
        hipGraphExec_t ge = graph_for(md.n_tokens, cb);  // captured at init (warmup_capture); a miss captures now
        if (!ge && !error.empty()) return false;
        if (ge) {
          if (hipGraphLaunch(ge, stream_) != hipSuccess) return fail("hipGraphLaunch failed");
          launched = true;
          last_graph_ = ge;
        }
      }
      if (!launched) {
        last_graph_ = nullptr;
        if (!model_.forward(md, (int64_t)stream_, d_tokens_)) return fail(model_.error);
      }
    }
    if (!sample_stochastic(ids, is_prefill)) return false;
    if (hipMemcpyAsync(h_tokens_, d_tokens_, (size_t)B * 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("token download failed");
    // the split-K exchange's error word rides along with the tokens: EVERY step is checked before its tokens are committed
    uint32_t* dev_err = vra_scratch_error_word();
    if (dev_err && hipMemcpyAsync(h_err_, dev_err, 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("error-word download failed");
    uint32_t* comm_err = comm_ ? vra_comm_error_word(comm_) : nullptr;
    if (comm_err && hipMemcpyAsync(h_err_ + 1, comm_err, 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("error-word download failed");
    if (hipStreamSynchronize(stream_) != hipSuccess) return fail(std::string("stream error: ") + hipGetErrorString(hipGetLastError()));
    if (h_err_[1]) {
      h_err_[1] = 0;
      (void)hipMemsetAsync(comm_err, 0, 4, stream_);
      return fail("one-shot all-reduce timed out waiting for a peer (results of this step are invalid)" + comm_timeout_detail(comm_));
    }
    if (h_err_[0]) {
      h_err_[0] = 0;
      vra_scratch_reset_after_error(stream_);  // flags too: a timed-out exchange leaves them undefined
      return fail("split-K exchange timed out on the device (results of this step are invalid)");
    }
    tokens->assign(h_tokens_, h_tokens_ + B);
    return true;
  }

  // ---- ModelRunner::sample beyond argmax (runner.rs:1390-1570).  The forward (or its graph) has left f32 logits
  // [B, V] and the argmax tokens; a stochastic strategy overwrites the tokens.
  bool sample_stochastic(const std::vector<int>& ids, bool is_prefill) {
    auto& run = sched_->running();
    const int B = (int)ids.size();
    if (is_prefill) {  // strategy and penalties of the batch = those of its first sequence, cached for decode (A3)
      const Sequence& s0 = run[ids[0]];
      CachedSampling c;
      const bool has_t = s0.temperature >= 0.f;
      const bool greedy = has_t && s0.temperature == 0.f;
      const bool has_user = has_t || s0.top_k > 0 || (s0.top_p > 0.f && s0.top_p < 1.f);
      if (greedy) {
        c.kind = 0;
      } else if (has_user) {  // LogitsProcessor::get_strategy (logits_processor.rs:48-65)
        const bool t_ok = has_t && s0.temperature >= 1e-7f;
        c.kind = t_ok ? 1 : 0;
        c.temperature = t_ok ? s0.temperature : 1.f;
        c.k = s0.top_k > 0 ? s0.top_k : 0;
        c.p = s0.top_p >= 0.f ? s0.top_p : -1.f;
      } else {  // no user config and no generation_config: top-k 32, top-p 0.95, temperature 0.7 (runner.rs:1475-1486, A4)
        c.kind = 1, c.k = 32, c.p = 0.95f, c.temperature = 0.7f;
      }
      c.has_freq = s0.has_freq_penalty, c.freq = s0.freq_penalty;
      c.has_pres = s0.has_pres_penalty, c.pres = s0.pres_penalty;
      cached_sampling_ = c;
    }
    const CachedSampling& c = cached_sampling_;
    const bool any_pen = c.has_freq || c.has_pres;
    if (c.kind == 0 && !any_pen) return true;  // the argmax tokens stand
    if (c.k > 256) return fail("top_k > 256 is not supported by the device sampler");
    if (!is_prefill && any_pen) {  // runner.rs:1519-1541: last 128 sampled tokens once more than 128 were sampled
      int32_t* h_len = (int32_t*)(h_pen_ctx_ + (size_t)max_seqs_ * kPenaltyWindow);
      float* h_f = (float*)(h_len + max_seqs_);
      float* h_p = h_f + max_seqs_;
      bool any = false;
      for (int b = 0; b < B; b++) {
        const std::vector<uint32_t>& t = run[ids[b]].sampled;
        const int n = (int)t.size() > kPenaltyWindow ? kPenaltyWindow : 0;
        for (int i = 0; i < n; i++) h_pen_ctx_[(size_t)b * kPenaltyWindow + i] = t[t.size() - n + i];
        h_len[b] = n;
        h_f[b] = c.has_freq ? c.freq : 0.f;
        h_p[b] = c.has_pres ? c.pres : 0.f;
        any = any || n > 0;
      }
      if (any) {
        const size_t bytes = (size_t)max_seqs_ * (kPenaltyWindow * 4 + 12);
        if (hipMemcpyAsync(d_pen_, h_pen_ctx_, bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail("penalty upload failed");
        const uint8_t* d_len = d_pen_ + (size_t)max_seqs_ * kPenaltyWindow * 4;
        vra_apply_penalties(model_.logits(), (const uint32_t*)d_pen_, (const int32_t*)d_len, B, kPenaltyWindow, mc_.vocab_size,
                            (const float*)(d_len + (size_t)max_seqs_ * 4), (const float*)(d_len + (size_t)max_seqs_ * 8), (int64_t)stream_);
      }
    }
    if (c.kind == 0) {  // greedy with penalties
      vra_argmax_f32(model_.logits(), d_tokens_, B, mc_.vocab_size, (int64_t)stream_);
    } else {
      // one fresh seed per call, as `self.rng.lock().next_u64()` (logits_processor.rs:223-226)
      ++sample_calls_;
      const uint64_t seed = (uint64_t)vra_hash32_host(ec_.seed ? ec_.seed : 1234, sample_calls_) << 32 | vra_hash32_host(ec_.seed + 1, sample_calls_);
      vra_sample(model_.logits(), d_tokens_, B, mc_.vocab_size, c.k, c.p, c.temperature, seed, nullptr, nullptr, (int64_t)stream_);
    }
    const char* e = vra_last_error();
    if (e && e[0]) return fail(std::string("sampler: ") + e);
    return true;
  }

  // ---- dry engine: the two halves of step() around the (absent) forward pass
  int dry_schedule(int* is_prefill_out) {
    pend_valid_ = false;
    bool is_prefill = false;
    pend_ids_ = sched_->schedule(&is_prefill);
    pend_prefill_ = is_prefill;
    if (is_prefill_out) *is_prefill_out = is_prefill ? 1 : 0;
    if (!execute_swaps()) return -1;
    if (pend_ids_.empty()) {
      nothing_scheduled();
      return 0;
    }
    int nb = 0;
    pend_md_ = is_prefill ? prepare_prefill(pend_ids_, &nb) : prepare_decode(pend_ids_, (int)pend_ids_.size());
    pend_valid_ = true;
    return (int)pend_ids_.size();
  }
  int dry_commit(const uint32_t* toks, int n) {
    if (!pend_valid_ || n != (int)pend_ids_.size()) {
      fail("dry_commit: no pending step or wrong token count");
      return -1;
    }
    pend_valid_ = false;
    std::vector<uint32_t> tokens(toks, toks + n);
    finish_step(pend_ids_, pend_prefill_, tokens);
    return n;
  }
  void finish_step(const std::vector<int>& ids, bool is_prefill, const std::vector<uint32_t>& tokens) {
    const double now = now_ms();
    if (cached_sampling_.has_freq || cached_sampling_.has_pres) {  // runner.rs:1549-1563: every sampled token is tracked
      auto& run = sched_->running();
      for (size_t i = 0; i < ids.size() && i < tokens.size(); i++)
        if (ids[i] >= 0 && ids[i] < (int)run.size()) run[ids[i]].sampled.push_back(tokens[i]);
    }
    if (is_prefill) {
      std::vector<int> keep, ridx;
      sched_->filter_prefill_finished(ids, &keep, &ridx);  // only fully-prefilled prompts keep their token (engine.rs:906-916)
      std::vector<uint32_t> kept;
      for (int p : keep) kept.push_back(tokens[p]);
      sched_->postprocess(ridx, kept, now);
    } else {
      sched_->postprocess(ids, tokens, now);
    }
    collect();
  }

  // ---- one engine step (engine.rs:1693-1757)
  int step(int* is_prefill_out) {
    if (dry()) {
      fail("step() on a host-only engine: use vra_engine_dry_schedule / vra_engine_dry_commit");
      return -1;
    }
    bool is_prefill = false;
    std::vector<int> ids = sched_->schedule(&is_prefill);
    if (is_prefill_out) *is_prefill_out = is_prefill ? 1 : 0;
    if (!execute_swaps()) return -1;
    if (ids.empty()) {
      nothing_scheduled();
      return 0;
    }
    std::vector<uint32_t> tokens;
    if (!run(ids, is_prefill, &tokens)) return -1;
    finish_step(ids, is_prefill, tokens);
    return (int)ids.size();
  }
  void collect() {
    for (auto& s : sched_->clear_finished()) {
      RequestResult& r = results_[s.id];
      r.output = s.output_ids;
      r.created_ms = s.created_ms;
      r.first_token_ms = s.first_token_ms;
      r.finished_ms = s.finished_ms;
      r.finished = true;
      r.aborted = s.aborted;
    }
  }
  const Sequence* find_live(int64_t id) const {
    for (auto& s : sched_->running())
      if (s.id == id) return &s;
    for (auto& s : sched_->waiting())
      if (s.id == id) return &s;
    return nullptr;
  }
};

}  // namespace vra

using vra::Engine;

// ================================================================================================
// C API — block manager
// ================================================================================================
namespace {
struct BmHandle {
  vra::BlockManager bm;
  std::map<int64_t, vra::Sequence> seqs;
  int64_t next = 1;
  BmHandle(int nb, int bs, bool pc, float frac) : bm(nb, bs, pc, frac) {}
};
}  // namespace
extern "C" void* vra_bm_create(int32_t num_blocks, int32_t block_size, int32_t enable_prefix_cache, float prefix_cache_fraction) {
  return new BmHandle(num_blocks, block_size, enable_prefix_cache != 0, prefix_cache_fraction);
}
extern "C" void vra_bm_destroy(void* bm) { delete static_cast<BmHandle*>(bm); }
extern "C" int32_t vra_bm_num_free_blocks(const void* bm) { return static_cast<const BmHandle*>(bm)->bm.num_free_blocks(); }
extern "C" int64_t vra_bm_seq_create(void* bm, const uint32_t* h_tokens, int32_t n) {
  auto* h = static_cast<BmHandle*>(bm);
  vra::Sequence s;
  s.id = h->next++;
  s.block_size = h->bm.block_size();
  s.token_ids.assign(h_tokens, h_tokens + n);
  s.prompt_len = n;
  s.last_token = n ? h_tokens[n - 1] : 0;
  h->seqs[s.id] = s;
  return s.id;
}
extern "C" void vra_bm_seq_free(void* bm, int64_t seq) { static_cast<BmHandle*>(bm)->seqs.erase(seq); }
extern "C" int32_t vra_bm_can_allocate(const void* bm, int64_t seq) {
  auto* h = const_cast<BmHandle*>(static_cast<const BmHandle*>(bm));
  return h->bm.can_allocate(h->seqs.at(seq)) ? 1 : 0;
}
extern "C" int32_t vra_bm_allocate(void* bm, int64_t seq) {
  auto* h = static_cast<BmHandle*>(bm);
  vra::Sequence& s = h->seqs.at(seq);
  if (!s.block_table.empty()) return -1;
  if (!h->bm.allocate(s)) return -1;
  return s.num_cached_tokens;
}
extern "C" int32_t vra_bm_can_append(const void* bm, int64_t seq) {
  auto* h = static_cast<const BmHandle*>(bm);
  return h->bm.can_append(h->seqs.at(seq)) ? 1 : 0;
}
extern "C" int32_t vra_bm_may_append(void* bm, int64_t seq) {
  auto* h = static_cast<BmHandle*>(bm);
  return h->bm.may_append(h->seqs.at(seq)) ? 0 : -1;
}
extern "C" void vra_bm_append_token(void* bm, int64_t seq, uint32_t token) { static_cast<BmHandle*>(bm)->seqs.at(seq).append_token(token); }
extern "C" void vra_bm_deallocate(void* bm, int64_t seq) {
  auto* h = static_cast<BmHandle*>(bm);
  vra::Sequence& s = h->seqs.at(seq);
  h->bm.cache_sequence(s);  // scheduler.rs:619-621: cache_sequence then deallocate
  h->bm.deallocate(s);
  s.block_table.clear();
}
extern "C" int32_t vra_bm_seq_len(const void* bm, int64_t seq) { return static_cast<const BmHandle*>(bm)->seqs.at(seq).len(); }
extern "C" int32_t vra_bm_seq_num_cached_tokens(const void* bm, int64_t seq) {
  return static_cast<const BmHandle*>(bm)->seqs.at(seq).num_cached_tokens;
}
extern "C" int32_t vra_bm_seq_block_table(const void* bm, int64_t seq, uint32_t* h_out, int32_t cap) {
  const auto& bt = static_cast<const BmHandle*>(bm)->seqs.at(seq).block_table;
  for (int i = 0; i < (int)bt.size() && i < cap; i++) h_out[i] = bt[i];
  return (int)bt.size();
}
extern "C" int32_t vra_bm_prefix_cached_blocks(const void* bm) { return static_cast<const BmHandle*>(bm)->bm.prefix_cache_blocks(); }
extern "C" int32_t vra_bm_evict_prefix(void* bm, int32_t n) { return static_cast<BmHandle*>(bm)->bm.evict_prefix_cache(n); }

// ---- prefix cache on its own (the reference's unit tests, prefix_cache.rs:362-403, drive it directly)
extern "C" void* vra_pc_create(int32_t block_size, int32_t max_cached_blocks) { return new vra::PrefixCache(block_size, true, max_cached_blocks); }
extern "C" void vra_pc_destroy(void* pc) { delete static_cast<vra::PrefixCache*>(pc); }
extern "C" int32_t vra_pc_insert_prefix(void* pc, const uint32_t* h_tokens, int32_t n_tokens, const int32_t* h_blocks, int32_t n_blocks,
                                        int32_t* h_evicted, int32_t cap, int32_t* h_n_evicted) {
  std::vector<int> blocks(h_blocks, h_blocks + n_blocks);
  auto up = static_cast<vra::PrefixCache*>(pc)->insert_prefix(h_tokens, n_tokens, blocks);
  for (int i = 0; i < (int)up.evicted.size() && i < cap; i++) h_evicted[i] = up.evicted[i];
  if (h_n_evicted) *h_n_evicted = (int)up.evicted.size();
  return (int)up.inserted.size();
}
extern "C" int32_t vra_pc_match_prefix(void* pc, const uint32_t* h_tokens, int32_t n_tokens, int32_t* h_blocks, int32_t cap) {
  auto* c = static_cast<vra::PrefixCache*>(pc);
  auto m = c->match_prefix(h_tokens, n_tokens);
  if (m.has_hash && h_blocks) {
    auto b = c->blocks_for_match(m.last_hash);
    for (int i = 0; i < (int)b.size() && i < cap; i++) h_blocks[i] = b[i];
  }
  return m.matched_blocks;
}
extern "C" int32_t vra_pc_cached_blocks(const void* pc) { return static_cast<const vra::PrefixCache*>(pc)->cached_blocks(); }
extern "C" int32_t vra_pc_evict_blocks(void* pc, int32_t n, int32_t* h_evicted, int32_t cap) {
  auto ev = static_cast<vra::PrefixCache*>(pc)->evict_blocks(n);
  for (int i = 0; i < (int)ev.size() && i < cap; i++) h_evicted[i] = ev[i];
  return (int)ev.size();
}

// ================================================================================================
// C API — engine
// ================================================================================================
extern "C" void* vra_engine_create(const vra_model_config* mc, const vra_engine_config* ec) {
  if (!mc || !ec) return nullptr;
  if (ec->device >= 0 && hipSetDevice(ec->device) != hipSuccess) return nullptr;
  return new Engine(*mc, *ec);
}
extern "C" void vra_engine_destroy(void* e) { delete static_cast<Engine*>(e); }
extern "C" const char* vra_engine_last_error(const void* e) { return static_cast<const Engine*>(e)->error.c_str(); }
extern "C" int32_t vra_engine_init_synthetic(void* e) {
  auto* en = static_cast<Engine*>(e);
  if (!en->model_.init_synthetic(en->ec_.seed ? en->ec_.seed : 1234)) {
    en->error = en->model_.error;
    return -1;
  }
  return 0;
}
extern "C" int32_t vra_engine_load_tensor(void* e, const char* name, const void* h_data, const int64_t* shape, int32_t ndim, int32_t elem_bytes) {
  auto* en = static_cast<Engine*>(e);
  if (!en->model_.load_tensor(name, h_data, shape, ndim, elem_bytes)) {
    en->error = en->model_.error;
    return -1;
  }
  return 0;
}
extern "C" int32_t vra_engine_set_comm(void* e, void* comm) {
  auto* en = static_cast<Engine*>(e);
  en->comm_ = comm;
  en->model_.set_comm(comm);
  return 0;
}
extern "C" int32_t vra_engine_finalize_weights(void* e) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry()) return en->finalize() ? 0 : -1;
  if (!en->model_.finalize_weights()) {  // repack of explicit tensors; a no-op after synthetic init (idempotent)
    en->error = en->model_.error;
    return -1;
  }
  return en->finalize() ? 0 : -1;
}
extern "C" int32_t vra_engine_copy_logits(void* e, float* h_out, int32_t n_seqs) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry() || !en->sched_ || !h_out || n_seqs < 1 || n_seqs > en->max_seqs_) {
    en->error = "vra_engine_copy_logits: finalised GPU engine and 1..max_num_seqs rows";
    return -1;
  }
  if (hipMemcpyAsync(h_out, en->model_.logits(), (size_t)n_seqs * en->mc_.vocab_size * 4, hipMemcpyDeviceToHost, en->stream_) != hipSuccess ||
      hipStreamSynchronize(en->stream_) != hipSuccess) {
    en->error = "vra_engine_copy_logits: copy failed";
    return -1;
  }
  return 0;
}
// which fused-norm launches of a step of `rows` rows use the deferred order (model.h norm_deferred_mask): what the parity tests ask
// the engine so that the oracle restates the order the engine actually runs
extern "C" int32_t vra_engine_norm_deferred(void* e, int32_t rows, int32_t layer) {
  auto* en = static_cast<Engine*>(e);
  return en->dry() ? 0 : en->model_.norm_deferred_mask(rows, layer);
}
// parity instrumentation of the tensor-parallel forward (model.h `set_tp_snapshots`): stage copies of one layer (`on` = 1 + the layer, 0 = off), read back per stage
extern "C" void vra_engine_debug_tp_snapshots(void* e, int32_t on) { static_cast<Engine*>(e)->model_.set_tp_snapshots(on); }
extern "C" int64_t vra_engine_debug_read_tp_snapshot(void* e, int32_t idx, void* h_out, int64_t max_bytes) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry()) return -1;
  return en->model_.read_tp_snapshot(idx, h_out, max_bytes, (int64_t)en->stream_);
}
extern "C" int32_t vra_engine_finalize_model(void* e) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry()) return 0;
  if (!en->model_.finalize_weights()) {
    en->error = en->model_.error;
    return -1;
  }
  return 0;
}
extern "C" int32_t vra_engine_update_config(void* e, const vra_engine_config* cfg) {
  auto* en = static_cast<Engine*>(e);
  if (!cfg || en->prepared_ || en->sched_) {
    en->error = "vra_engine_update_config: only before the engine has allocated its buffers (before finalize / plan_kv_blocks)";
    return -1;
  }
  en->ec_.num_gpu_blocks = cfg->num_gpu_blocks;
  if (cfg->max_num_seqs > 0) en->ec_.max_num_seqs = cfg->max_num_seqs;
  if (cfg->max_model_len > 0) en->ec_.max_model_len = cfg->max_model_len;
  en->ec_.cpu_mem_fold = cfg->cpu_mem_fold;
  if (cfg->kv_fraction > 0.f) en->ec_.kv_fraction = cfg->kv_fraction;
  return 0;
}
// KVCacheAllocator plan of THIS rank before the cache is allocated (weights repacked, activation buffers reserved):
// the block count `finalize` would choose from free memory x kv_fraction.  Tensor-parallel launchers call it on every
// rank, take the minimum and hand it to vra_engine_set_num_gpu_blocks, so that all schedulers see the same cache.
extern "C" int64_t vra_engine_plan_kv_blocks(void* e) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry()) return en->ec_.num_gpu_blocks;
  if (!en->model_.finalize_weights()) {
    en->error = en->model_.error;
    return -1;
  }
  if (!en->prepare()) return -1;
  return en->planned_blocks_;
}
extern "C" int32_t vra_engine_set_num_gpu_blocks(void* e, int32_t n) {
  auto* en = static_cast<Engine*>(e);
  if (n < 2 || en->sched_) {
    en->error = "vra_engine_set_num_gpu_blocks: need n >= 2, before finalize";
    return -1;
  }
  en->ec_.num_gpu_blocks = n;
  return 0;
}
// cache::swap_blocks as the runner executes it for the reference's engine (runner.rs:1626-1670, MessageType::KVCacheSwap):
// whole blocks (source id, destination id) between the GPU cache and this engine's pinned swap space, every layer, K and V,
// on the engine stream.  The block ids are the CALLER's bookkeeping (the engine process owns the block manager there).
extern "C" int32_t vra_engine_swap_blocks(void* e, const int64_t* h_pairs, int32_t n_pairs, int32_t swap_in) {
  auto* en = static_cast<Engine*>(e);
  if (en->dry() || !en->sched_ || en->num_cpu_blocks_ <= 0 || en->h_swap_k_.empty()) {
    en->error = "vra_engine_swap_blocks: no CPU swap space (vra_engine_config.cpu_mem_fold = 0, or a host-only engine)";
    return -1;
  }
  const int64_t ngpu = en->model_.num_blocks(), ncpu = en->num_cpu_blocks_;
  for (int i = 0; i < n_pairs; i++) {
    const int64_t src = h_pairs[2 * i], dst = h_pairs[2 * i + 1];
    if (src < 0 || dst < 0 || src >= (swap_in ? ncpu : ngpu) || dst >= (swap_in ? ngpu : ncpu)) {
      en->error = "vra_engine_swap_blocks: block id out of range";
      return -1;
    }
  }
  const int64_t bb = (int64_t)en->model_.kv_block_bytes();
  for (int l = 0; l < en->mc_.num_layers; l++) {
    if (swap_in) {
      vra_swap_blocks(en->h_swap_k_[l], en->model_.k_cache(l), h_pairs, n_pairs, bb, 2, (int64_t)en->stream_);
      vra_swap_blocks(en->h_swap_v_[l], en->model_.v_cache(l), h_pairs, n_pairs, bb, 2, (int64_t)en->stream_);
    } else {
      vra_swap_blocks(en->model_.k_cache(l), en->h_swap_k_[l], h_pairs, n_pairs, bb, 1, (int64_t)en->stream_);
      vra_swap_blocks(en->model_.v_cache(l), en->h_swap_v_[l], h_pairs, n_pairs, bb, 1, (int64_t)en->stream_);
    }
  }
  (swap_in ? en->swap_in_blocks_ : en->swap_out_blocks_) += n_pairs;
  const char* err = vra_last_error();
  if ((err && err[0]) || hipStreamSynchronize(en->stream_) != hipSuccess) {
    en->error = std::string("vra_engine_swap_blocks: ") + (err && err[0] ? err : "stream error");
    return -1;
  }
  return 0;
}
extern "C" void vra_engine_swap_stats(const void* e, int64_t* out4) {
  const Engine* en = static_cast<const Engine*>(e);
  out4[0] = en->num_cpu_blocks_;
  out4[1] = en->bm_ ? en->bm_->num_free_cpu_blocks() : 0;
  out4[2] = en->swap_out_blocks_;
  out4[3] = en->swap_in_blocks_;
}
extern "C" int32_t vra_engine_num_gpu_blocks(const void* e) { return static_cast<const Engine*>(e)->model_.num_blocks(); }
extern "C" int64_t vra_engine_add_request(void* e, const uint32_t* h_prompt, int32_t n_prompt, int32_t max_tokens, int32_t ignore_eos,
                                          const uint32_t* h_eos, int32_t n_eos) {
  auto* en = static_cast<Engine*>(e);
  if (!en->sched_) {
    en->error = "engine not finalised";
    return -1;
  }
  vra::Sequence s;
  s.token_ids.assign(h_prompt, h_prompt + n_prompt);
  s.max_tokens = max_tokens > 0 ? max_tokens : 16384;
  s.ignore_eos = ignore_eos != 0;
  if (h_eos) s.eos.assign(h_eos, h_eos + n_eos);
  int64_t id = en->sched_->add(std::move(s));
  if (id < 0) en->error = en->sched_->last_error;
  return id;
}
extern "C" int64_t vra_engine_add_request_ex(void* e, const uint32_t* h_prompt, int32_t n_prompt, int32_t max_tokens, int32_t ignore_eos,
                                             const uint32_t* h_eos, int32_t n_eos, const vra_sampling_params* sp) {
  auto* en = static_cast<Engine*>(e);
  if (!en->sched_) {
    en->error = "engine not finalised";
    return -1;
  }
  vra::Sequence s;
  s.token_ids.assign(h_prompt, h_prompt + n_prompt);
  s.max_tokens = max_tokens > 0 ? max_tokens : 16384;
  s.ignore_eos = ignore_eos != 0;
  if (h_eos) s.eos.assign(h_eos, h_eos + n_eos);
  if (sp) {
    s.temperature = sp->temperature;
    s.top_k = sp->top_k;
    s.top_p = sp->top_p;
    s.has_freq_penalty = sp->has_frequency_penalty != 0, s.freq_penalty = sp->frequency_penalty;
    s.has_pres_penalty = sp->has_presence_penalty != 0, s.pres_penalty = sp->presence_penalty;
  }
  int64_t id = en->sched_->add(std::move(s));
  if (id < 0) en->error = en->sched_->last_error;
  return id;
}
extern "C" int32_t vra_engine_step(void* e, int32_t* h_is_prefill) { return static_cast<Engine*>(e)->step(h_is_prefill); }
extern "C" int32_t vra_engine_dry_schedule(void* e, int32_t* h_is_prefill, vra_step_meta* out) {
  auto* en = static_cast<Engine*>(e);
  if (!en->dry() || !en->sched_) {
    en->error = "vra_engine_dry_schedule: not a finalised host-only engine (device = -1)";
    return -1;
  }
  const int n = en->dry_schedule(h_is_prefill);
  if (out) {
    memset(out, 0, sizeof(*out));
    if (n > 0) {
      const vra::InputMetadata& md = en->pend_md_;
      out->n_tokens = md.n_tokens;
      out->n_seqs = md.n_seqs;
      out->max_blocks = md.max_blocks;
      out->max_seqlen_q = md.max_seqlen_q;
      out->max_context_len = md.max_context_len;
      out->input_ids = md.input_ids;
      out->positions = md.positions;
      out->slot_mapping = md.slot_mapping;
      out->block_tables = md.block_tables;
      out->context_lens = md.context_lens;
      out->cu_seqlens_q = md.cu_seqlens_q;
      for (int i = 0; i < n && i < 64; i++) out->request_ids[i] = en->sched_->running()[en->pend_ids_[i]].id;
    }
  }
  return n;
}
extern "C" int32_t vra_engine_dry_commit(void* e, const uint32_t* h_tokens, int32_t n) { return static_cast<Engine*>(e)->dry_commit(h_tokens, n); }
extern "C" int32_t vra_engine_has_unfinished(const void* e) {
  auto* en = static_cast<const Engine*>(e);
  return en->sched_ && en->sched_->has_unfinished() ? 1 : 0;
}
extern "C" int32_t vra_engine_request_finished(const void* e, int64_t req) {
  auto* en = static_cast<const Engine*>(e);
  auto it = en->results_.find(req);
  return it != en->results_.end() && it->second.finished ? 1 : 0;
}
extern "C" int32_t vra_engine_request_output(const void* e, int64_t req, uint32_t* h_out, int32_t cap) {
  auto* en = static_cast<const Engine*>(e);
  const std::vector<uint32_t>* out = nullptr;
  auto it = en->results_.find(req);
  if (it != en->results_.end()) out = &it->second.output;
  else if (const vra::Sequence* s = en->find_live(req)) out = &s->output_ids;
  if (!out) return -1;
  for (int i = 0; i < (int)out->size() && i < cap; i++) h_out[i] = (*out)[i];
  return (int)out->size();
}
extern "C" int32_t vra_engine_request_times(const void* e, int64_t req, double h_times[3]) {
  auto* en = static_cast<const Engine*>(e);
  auto it = en->results_.find(req);
  if (it != en->results_.end()) {
    h_times[0] = it->second.created_ms;
    h_times[1] = it->second.first_token_ms;
    h_times[2] = it->second.finished_ms;
    return 0;
  }
  if (const vra::Sequence* s = en->find_live(req)) {
    h_times[0] = s->created_ms;
    h_times[1] = s->first_token_ms;
    h_times[2] = 0;
    return 0;
  }
  return -1;
}
extern "C" void vra_engine_release_request(void* e, int64_t req) { static_cast<Engine*>(e)->results_.erase(req); }
extern "C" int64_t vra_engine_stream(const void* e) { return (int64_t) static_cast<const Engine*>(e)->stream_; }

// One forward with caller-built metadata (the runner process' path).  h_logits_out: the f32 logits [n_seqs, vocab] come back to the host
// (vra_engine_forward_raw).  h_tokens_out: only the sampled token ids do (vra_engine_forward_tokens): the decode step replays the
// captured hipGraph of its batch / context bucket when the engine has graphs, greedy tokens come out of the lm_head launch itself and
// a stochastic strategy runs on the device logits (vra_sample) — nothing of vocabulary size crosses PCIe.
struct RawSampling {
  int kind = 0;  // 0 greedy (first maximal index), 1 temperature / top-k / top-p
  int k = 0;
  float p = -1.f, t = 1.f;
  uint64_t seed = 0;
};
static int32_t forward_raw_impl(Engine* en, const uint32_t* h_ids, const int64_t* h_positions, const int64_t* h_slot_mapping,
                                int32_t n_tokens, int32_t is_prefill, const uint32_t* h_block_tables, int32_t max_blocks,
                                const uint32_t* h_context_lens, const uint32_t* h_cu_seqlens_q, int32_t n_seqs,
                                float* h_logits_out, uint32_t* h_tokens_out, const RawSampling* smp) {
  if (!en->sched_) {
    en->error = "engine not finalised";
    return -1;
  }
  if (n_tokens > std::max(en->max_step_tokens_, en->max_seqs_) || n_seqs > en->max_seqs_ || max_blocks > en->max_blocks_per_seq_) {
    en->error = "forward_raw: batch exceeds engine limits";
    return -1;
  }
  if (!is_prefill && n_tokens != n_seqs) {
    en->error = "forward_raw: a decode step carries one token per sequence";
    return -1;
  }
  // The metadata comes from a peer (the runner process hands over what the engine process sent): nothing reaches the device
  // that could index outside the cache, the block tables, the embedding or the logits rows.
  if (n_tokens <= 0 || n_seqs <= 0 || max_blocks <= 0 || !h_ids || !h_positions || !h_slot_mapping || !h_block_tables || !h_context_lens ||
      (!h_logits_out && !h_tokens_out) || (is_prefill && !h_cu_seqlens_q)) {
    en->error = "forward_raw: empty batch or null argument";
    return -1;
  }
  {
    const int64_t n_slots = (int64_t)en->model_.num_blocks() * en->ec_.block_size;
    const int64_t max_ctx = (int64_t)max_blocks * en->ec_.block_size;
    for (int t = 0; t < n_tokens; t++) {
      if (h_ids[t] >= (uint32_t)en->mc_.vocab_size) return en->error = "forward_raw: token id " + std::to_string(h_ids[t]) + " outside the vocabulary", -1;
      if (h_slot_mapping[t] >= n_slots) return en->error = "forward_raw: slot " + std::to_string(h_slot_mapping[t]) + " outside the KV cache", -1;
      if (is_prefill ? h_slot_mapping[t] < 0 : h_slot_mapping[t] < -1)  // (-1: a padded decode lane writes nothing)
        return en->error = "forward_raw: negative slot", -1;
      if (h_positions[t] < 0 || h_positions[t] >= vra_rope_table_rows(&en->mc_))
        return en->error = "forward_raw: position " + std::to_string(h_positions[t]) + " outside the rotary table", -1;
    }
    for (int b = 0; b < n_seqs; b++) {
      if ((int64_t)h_context_lens[b] > max_ctx) return en->error = "forward_raw: context length beyond the block table", -1;
      const int used = (int)(((int64_t)h_context_lens[b] + en->ec_.block_size - 1) / en->ec_.block_size);
      for (int k = 0; k < used; k++)
        if (h_block_tables[(size_t)b * max_blocks + k] >= (uint32_t)en->model_.num_blocks())
          return en->error = "forward_raw: block id " + std::to_string(h_block_tables[(size_t)b * max_blocks + k]) + " outside the KV cache", -1;
    }
    if (is_prefill) {
      if (h_cu_seqlens_q[0] != 0 || h_cu_seqlens_q[n_seqs] != (uint32_t)n_tokens) return en->error = "forward_raw: cu_seqlens_q does not cover the tokens", -1;
      for (int b = 0; b < n_seqs; b++) {
        if (h_cu_seqlens_q[b + 1] <= h_cu_seqlens_q[b]) return en->error = "forward_raw: a sequence without tokens in a prefill step", -1;
        if (h_cu_seqlens_q[b + 1] - h_cu_seqlens_q[b] > h_context_lens[b]) return en->error = "forward_raw: more query tokens than context", -1;
      }
    }
  }
  // decode with graphs (tokens wanted): the static layout of Engine::prepare_decode — block-table rows max_blocks_per_seq_ apart, the
  // batch padded to its bucket with lanes that write nothing and attend to nothing — so that the captured graph can be replayed
  const bool graph_decode = !is_prefill && h_tokens_out && en->ec_.use_graph;
  const int bucket = graph_decode ? std::max(std::min(Engine::batch_bucket(n_seqs), en->max_seqs_), (int)n_seqs) : n_tokens;
  const int bt_stride = graph_decode ? en->max_blocks_per_seq_ : max_blocks;
  memcpy(en->h_meta_ + (is_prefill ? en->off_ids_ : en->off_dids_), h_ids, (size_t)n_tokens * 4);
  memcpy(en->h_meta_ + (is_prefill ? en->off_pos_ : en->off_dpos_), h_positions, (size_t)n_tokens * 8);
  memcpy(en->h_meta_ + (is_prefill ? en->off_slots_ : en->off_dslots_), h_slot_mapping, (size_t)n_tokens * 8);
  if (bt_stride == max_blocks) {
    memcpy(en->h_meta_ + en->off_bt_, h_block_tables, (size_t)n_seqs * max_blocks * 4);
  } else {
    for (int b = 0; b < n_seqs; b++) memcpy(en->h_meta_ + en->off_bt_ + (size_t)b * bt_stride * 4, h_block_tables + (size_t)b * max_blocks, (size_t)max_blocks * 4);
  }
  memcpy(en->h_meta_ + en->off_ctx_, h_context_lens, (size_t)n_seqs * 4);
  for (int b = n_seqs; b < bucket && graph_decode; b++) {
    ((uint32_t*)(en->h_meta_ + en->off_dids_))[b] = 0;
    ((int64_t*)(en->h_meta_ + en->off_dpos_))[b] = 0;
    ((int64_t*)(en->h_meta_ + en->off_dslots_))[b] = -1;
    ((uint32_t*)(en->h_meta_ + en->off_ctx_))[b] = 0;
  }
  uint32_t* h_last = (uint32_t*)(en->h_meta_ + en->off_last_);
  vra::InputMetadata md;
  md.is_prefill = is_prefill != 0;
  md.n_tokens = graph_decode ? bucket : n_tokens;
  md.n_seqs = graph_decode ? bucket : n_seqs;
  md.max_blocks = bt_stride;
  md.max_seqlen_q = 1;
  md.max_context_len = 0;
  for (int b = 0; b < n_seqs; b++) md.max_context_len = std::max(md.max_context_len, (int)h_context_lens[b]);
  if (is_prefill) {
    memcpy(en->h_meta_ + en->off_cuq_, h_cu_seqlens_q, (size_t)(n_seqs + 1) * 4);
    for (int b = 0; b < n_seqs; b++) {
      h_last[b] = h_cu_seqlens_q[b + 1] - 1;
      md.max_seqlen_q = std::max(md.max_seqlen_q, (int)(h_cu_seqlens_q[b + 1] - h_cu_seqlens_q[b]));
    }
  }
  en->bind(md);
  if (!en->upload_meta(md)) return -1;
  bool launched = false;
  if (graph_decode) {
    hipGraphExec_t ge = en->graph_for(md.n_tokens, Engine::ctx_bucket(md.max_context_len));  // captured at init; a miss captures now
    if (!ge && !en->error.empty()) return -1;
    if (ge) {
      if (hipGraphLaunch(ge, en->stream_) != hipSuccess) return en->error = "hipGraphLaunch failed", -1;
      launched = true;
    }
  }
  if (!launched && !en->model_.forward(md, (int64_t)en->stream_, h_tokens_out ? en->d_tokens_ : nullptr)) {
    en->error = en->model_.error;
    return -1;
  }
  en->last_graph_ = nullptr;
  if (h_logits_out &&
      hipMemcpyAsync(h_logits_out, en->model_.logits(), (size_t)n_seqs * en->mc_.vocab_size * 4, hipMemcpyDeviceToHost, en->stream_) != hipSuccess)
    return -1;
  if (h_tokens_out) {
    if (smp && smp->kind == 1) {  // LogitsProcessor::sample_with_strategy on the device logits (logits_processor.rs:199-271)
      if (smp->k > 256) return en->error = "top_k > 256 is not supported by the device sampler", -1;
      vra_sample(en->model_.logits(), en->d_tokens_, n_seqs, en->mc_.vocab_size, smp->k, smp->p, smp->t, smp->seed, nullptr, nullptr, (int64_t)en->stream_);
      const char* se = vra_last_error();
      if (se && se[0]) return en->error = std::string("sampler: ") + se, -1;
    }
    if (hipMemcpyAsync(en->h_tokens_, en->d_tokens_, (size_t)n_seqs * 4, hipMemcpyDeviceToHost, en->stream_) != hipSuccess) return en->error = "token download failed", -1;
  }
  // the device error words (a split-K slice or a tensor-parallel peer that never arrived) ride along, as in step()
  uint32_t* dev_err = vra_scratch_error_word();
  uint32_t* comm_err = en->comm_ ? vra_comm_error_word(en->comm_) : nullptr;
  if (dev_err) (void)hipMemcpyAsync(en->h_err_, dev_err, 4, hipMemcpyDeviceToHost, en->stream_);
  if (comm_err) (void)hipMemcpyAsync(en->h_err_ + 1, comm_err, 4, hipMemcpyDeviceToHost, en->stream_);
  if (hipStreamSynchronize(en->stream_) != hipSuccess) {
    en->error = "stream error in forward_raw";
    return -1;
  }
  if (comm_err && en->h_err_[1]) {
    en->h_err_[1] = 0;
    (void)hipMemsetAsync(comm_err, 0, 4, en->stream_);
    en->error = "one-shot all-reduce timed out waiting for a peer (results of this forward are invalid)" + comm_timeout_detail(en->comm_);
    return -1;
  }
  if (dev_err && en->h_err_[0]) {
    en->h_err_[0] = 0;
    vra_scratch_reset_after_error(en->stream_);  // flags too: a timed-out exchange leaves them undefined
    en->error = "split-K exchange timed out on the device (results of this forward are invalid)";
    return -1;
  }
  if (h_tokens_out) memcpy(h_tokens_out, en->h_tokens_, (size_t)n_seqs * 4);
  return 0;
}
extern "C" int32_t vra_engine_forward_raw(void* e, const uint32_t* h_ids, const int64_t* h_positions, const int64_t* h_slot_mapping,
                                          int32_t n_tokens, int32_t is_prefill, const uint32_t* h_block_tables, int32_t max_blocks,
                                          const uint32_t* h_context_lens, const uint32_t* h_cu_seqlens_q, int32_t n_seqs,
                                          float* h_logits_out) {
  if (!h_logits_out) return static_cast<Engine*>(e)->error = "forward_raw: empty batch or null argument", -1;
  return forward_raw_impl(static_cast<Engine*>(e), h_ids, h_positions, h_slot_mapping, n_tokens, is_prefill, h_block_tables, max_blocks, h_context_lens,
                          h_cu_seqlens_q, n_seqs, h_logits_out, nullptr, nullptr);
}
extern "C" int32_t vra_engine_forward_tokens(void* e, const uint32_t* h_ids, const int64_t* h_positions, const int64_t* h_slot_mapping,
                                             int32_t n_tokens, int32_t is_prefill, const uint32_t* h_block_tables, int32_t max_blocks,
                                             const uint32_t* h_context_lens, const uint32_t* h_cu_seqlens_q, int32_t n_seqs,
                                             int32_t stochastic, int32_t top_k, float top_p, float temperature, uint64_t seed,
                                             uint32_t* h_tokens_out) {
  if (!h_tokens_out) return static_cast<Engine*>(e)->error = "forward_tokens: null token buffer", -1;
  RawSampling smp;
  smp.kind = stochastic ? 1 : 0, smp.k = top_k, smp.p = top_p, smp.t = temperature, smp.seed = seed;
  return forward_raw_impl(static_cast<Engine*>(e), h_ids, h_positions, h_slot_mapping, n_tokens, is_prefill, h_block_tables, max_blocks, h_context_lens,
                          h_cu_seqlens_q, n_seqs, nullptr, h_tokens_out, &smp);
}

extern "C" double vra_engine_timed_decode(void* e, int32_t steps) {
  auto* en = static_cast<Engine*>(e);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, en->stream_);
  for (int i = 0; i < steps; i++) {
    int pf = 0;
    int n = en->step(&pf);
    if (n <= 0) break;
  }
  (void)hipEventRecord(b, en->stream_);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ms;
}

// Measurement aid (bench.py `step_overhead`): the decode graph of the most recent step launched `steps` times back to back with the
// metadata of that step left in place — no upload, no download, no host work between replays (each replay rewrites the same KV slot
// and the same token: engine state is untouched).  ms per replay = what the GPU needs for a step; vra_engine_timed_decode minus this
// is what the host loop (upload, launch, two downloads, wake-up, scheduler) adds per step: 9–13 us at bs 1
// (profiles/r04_ab_step_host_loop.txt — where a single download and a polling wait were also measured: no gain, not kept).
extern "C" double vra_engine_bench_replay(void* e, int32_t steps) {
  auto* en = static_cast<Engine*>(e);
  if (!en->last_graph_ || steps <= 0) return -1.0;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipGraphLaunch(en->last_graph_, en->stream_);
  (void)hipEventRecord(a, en->stream_);
  for (int i = 0; i < steps; i++) (void)hipGraphLaunch(en->last_graph_, en->stream_);
  (void)hipEventRecord(b, en->stream_);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return (double)ms / steps;
}

// roofline leg of bench.py: average launch duration of one decode-shaped GEMM kernel family,
// rotating over all layers' weights so nothing stays in L2/MALL, measured with HIP events on the
// engine stream. which: 0 qkv 1 o_proj 2 gate_up 3 down.
extern "C" double vra_engine_bench_gemm(void* e, int32_t which, int32_t m, int32_t iters) {
  auto* en = static_cast<Engine*>(e);
  const int L = en->mc_.num_layers;
  for (int i = 0; i < L; i++)
    if (!en->model_.launch_gemm(which, i % L, m, (int64_t)en->stream_)) return -1.0;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, en->stream_);
  for (int i = 0; i < iters; i++) en->model_.launch_gemm(which, i % L, m, (int64_t)en->stream_);
  (void)hipEventRecord(b, en->stream_);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return (double)ms / iters;
}
extern "C" int64_t vra_engine_gemm_bytes(const void* e, int32_t which, int32_t m) {
  return static_cast<const Engine*>(e)->model_.gemm_algorithmic_bytes(which, m);
}
extern "C" int64_t vra_engine_weight_bytes(const void* e) { return (int64_t) static_cast<const Engine*>(e)->model_.weight_bytes(); }
