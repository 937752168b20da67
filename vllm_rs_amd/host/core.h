// core.h — host-side restatement of vllm.rs src/core/{sequence,prefix_cache,block_manager,scheduler}.rs
// for the quantized-forward hot path.  Pure C++17, no HIP.
//
// Deliberate differences from the reference:
//   A15  the free list is an intrusive doubly-linked FIFO (O(1) removal) instead of VecDeque::retain;
//        FIFO order of reuse is preserved.
//   A16  the prefix hash is FNV-1a/splitmix over (parent, tokens) instead of SipHash — only matching
//        behaviour is observable.
//   A1   `ignore_eos` is honoured (the reference declares it and never reads it) so that synthetic-
//        weight benchmarks generate a fixed number of tokens.
//   PD transfer / mamba / images / tool-call stop are out of scope (§8 "next").  CPU swap follows
//   block_manager.rs:870-1010 and scheduler.rs:303-338,826-955; the scheduler only decides and records the block copies,
//   the engine executes them on its stream before the step that follows (SwapOp).
#pragma once
#include <stdint.h>

#include <deque>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace vra {

enum class SeqStatus { Waiting, Running, Finished, Swapped };

// src/core/sequence.rs:140-238
struct Sequence {
  int64_t id = 0;
  double created_ms = 0;       // sequence.rs:163-167 — TTFT start
  double first_token_ms = 0;   // engine.rs:1004-1012 decode_start_time
  double finished_ms = 0;
  double swapped_ms = -1;      // sequence.rs swapped_time: last swap-out / swap-in (cooling period)
  SeqStatus status = SeqStatus::Waiting;
  std::vector<uint32_t> token_ids;
  std::vector<uint32_t> output_ids;
  std::vector<uint32_t> block_table;
  int num_cached_tokens = 0;
  int block_size = 64;
  uint32_t last_token = 0;
  int prompt_len = 0;
  // SamplingParams (config.rs:476-520) as ModelRunner::sample reads them (runner.rs:1405-1497); "None" = unset:
  // temperature < 0, top_k <= 0, top_p < 0, penalties not set
  float temperature = 0.f;  // the plain request API is greedy (parity runs pass temperature = 0 explicitly, Appendix A4)
  int top_k = 0;
  float top_p = -1.f;
  bool has_freq_penalty = false, has_pres_penalty = false;
  float freq_penalty = 0.f, pres_penalty = 0.f;
  std::vector<uint32_t> sampled;  // tokens sampled for this sequence so far (runner.rs:1549-1563 `seq_tokens`): penalty context
  int max_tokens = 16384;  // scheduler.rs:598 default
  bool ignore_eos = false;
  std::vector<uint32_t> eos;
  bool aborted = false;

  int len() const { return (int)token_ids.size(); }
  int output_len() const { return (int)output_ids.size(); }
  int num_blocks() const { return (len() + block_size - 1) / block_size; }
  int last_block_num_tokens() const { return len() - (num_blocks() - 1) * block_size; }
  int num_cached_blocks() const { return num_cached_tokens / block_size; }
  void append_token(uint32_t t) {
    token_ids.push_back(t);
    output_ids.push_back(t);
    last_token = t;
  }
};

// src/core/prefix_cache.rs
class PrefixCache {
 public:
  PrefixCache(int block_size, bool enabled, int max_cached_blocks)
      : block_size_(block_size), enabled_(enabled), max_cached_blocks_(max_cached_blocks) {}
  bool enabled() const { return enabled_ && max_cached_blocks_ > 0; }
  int cached_blocks() const { return (int)entries_.size(); }
  struct Match {
    int matched_blocks = 0;
    bool has_hash = false;
    uint64_t last_hash = 0;
  };
  Match match_prefix(const uint32_t* tokens, int n);                                   // :72-117
  std::vector<int> blocks_for_match(uint64_t last_hash) const;                         // :119-132
  struct Update {
    std::vector<int> inserted, evicted;
  };
  Update insert_prefix(const uint32_t* tokens, int n, const std::vector<int>& blocks);  // :176-259
  std::vector<int> evict_blocks(int num_blocks);                                       // :261-293

 private:
  struct Entry {
    bool has_parent;
    uint64_t parent;
    int block_id;
    int children;
    uint64_t access_id;
  };
  static uint64_t hash_block(uint64_t parent, const uint32_t* tokens, int n);  // :343-348
  void touch(uint64_t h);
  void touch_leaf(uint64_t h);
  void compact_lru_if_needed();
  uint64_t next_access_id() { return ++access_counter_; }
  int block_size_;
  bool enabled_;
  int max_cached_blocks_;
  std::unordered_map<uint64_t, Entry> entries_;
  std::unordered_set<uint64_t> leaf_set_;
  std::deque<std::pair<uint64_t, uint64_t>> leaf_lru_;
  uint64_t access_counter_ = 0;
};

// src/core/block_manager.rs (GPU blocks only)
class BlockManager {
 public:
  BlockManager(int num_blocks, int block_size, bool prefix_cache, float prefix_fraction, int num_cpu_blocks = 0);
  int num_blocks() const { return (int)ref_.size(); }
  int num_free_blocks() const { return free_count_; }
  int block_size() const { return block_size_; }
  int required_blocks(const Sequence& s);       // :178-202
  bool can_allocate(const Sequence& s);         // :204-206
  bool allocate(Sequence& s);                   // :212-223 (false = no free blocks)
  void deallocate(const Sequence& s);           // :230-234
  bool can_append(const Sequence& s) const;     // :236-242
  bool may_append(Sequence& s);                 // :244-256
  void cache_sequence(const Sequence& s);       // :552-604
  bool prefix_cache_enabled() const { return cache_.enabled(); }
  int prefix_cache_blocks() const { return cache_.cached_blocks(); }
  int evict_prefix_cache(int n);                // scheduler.rs evict_prefix_cache_under_pressure
  float usage() const { return 1.0f - (float)free_count_ / (float)ref_.size(); }
  // ---- CPU swap space (block_manager.rs:870-1010).  Pairs are (source block, destination block).
  int num_cpu_blocks() const { return num_cpu_blocks_; }
  int num_free_cpu_blocks() const { return (int)free_cpu_.size(); }
  bool can_swap_out(const Sequence& s) const;   // :876-893: no shared block, more free CPU blocks than the sequence has
  bool can_swap_in(const Sequence& s) const;    // :897-906
  bool ensure_allocate(Sequence& s);            // :255-272: block table brought up to num_blocks()
  bool swap_out(const Sequence& s, std::vector<std::pair<int, int>>* gpu_to_cpu);  // :908-955 (caller deallocates)
  bool swap_in(const Sequence& s, std::vector<std::pair<int, int>>* cpu_to_gpu);   // :957-995 (blocks pre-allocated)
  void free_cpu_swap_for_seq(int64_t seq_id);   // :997-1007
  int evict_prefix_cache_until_free(int required_free);

 private:
  int pop_front();
  void push_back(int id);
  void unlink(int id);
  void allocate_block(int id);                        // :113-120
  void increment_ref(int id);                         // :274-281
  void decrement_ref(int id);                         // :283-289
  int adjusted_matched_blocks(int tokens_len, int m) const;  // :291-299
  bool allocate_fresh(Sequence& s);
  bool allocate_with_prefix(Sequence& s);
  int block_size_;
  std::vector<int> ref_;
  // intrusive FIFO free list: next_/prev_ indexed by block id, -1 = none, in_free_ flags membership
  std::vector<int> next_, prev_;
  std::vector<char> in_free_;
  int head_ = -1, tail_ = -1, free_count_ = 0;
  PrefixCache cache_;
  int num_cpu_blocks_ = 0;
  std::deque<int> free_cpu_;
  std::unordered_map<int64_t, std::vector<int>> swapped_map_;
};

// one batch of whole-block copies the engine has to run (vra_swap_blocks per layer and K/V) before its next forward
struct SwapOp {
  bool to_gpu = false;
  int64_t seq_id = 0;
  std::vector<std::pair<int, int>> pairs;  // (source block, destination block)
};

struct SchedulerConfig {
  int max_num_seqs = 32;
  int max_num_batched_tokens = 0;  // = num_blocks * block_size (kvcache_allocator.rs:668)
  int block_size = 64;
  int prefill_chunk = 8192;        // scheduler.rs:203
  int max_step_tokens = 16384;     // practical cap on tokens per prefill step (activation buffers)
  int max_model_len = 0;
  int swap_cooling_ms = 5000;       // scheduler.rs:49 SWAP_COOLING_PERIOD
  int min_tokens_left_for_swap = 1000;  // scheduler.rs:50
};

// src/core/scheduler.rs
class Scheduler {
 public:
  Scheduler(BlockManager* bm, const SchedulerConfig& cfg) : bm_(bm), cfg_(cfg) {}
  int64_t add(Sequence&& s);                                                     // :160-196
  // returns indexes into running() and whether this is a prefill step           // :200-380
  std::vector<int> schedule(bool* is_prefill);
  void postprocess(const std::vector<int>& ids, const std::vector<uint32_t>& tokens, double now_ms);  // :500-629
  // after a prefill step: (positions in `ids` whose prompt is complete, their new running indexes)   // :718-785
  void filter_prefill_finished(const std::vector<int>& ids, std::vector<int>* keep_pos, std::vector<int>* run_idx);
  std::vector<Sequence> clear_finished();                                        // :631-660
  std::vector<Sequence>& running() { return running_; }
  const std::deque<Sequence>& waiting() const { return waiting_; }
  const std::vector<Sequence>& swapped() const { return swapped_; }
  bool has_unfinished() const { return !running_.empty() || !waiting_.empty() || !swapped_.empty(); }
  // true when the only thing left to do is to wait for a swapped-out sequence's cooling period
  bool only_swapped_left() const { return running_.empty() && waiting_.empty() && !swapped_.empty(); }
  std::vector<SwapOp> take_swap_ops() {  // the copies decided by the last schedule(), in order
    std::vector<SwapOp> r;
    r.swap(swap_ops_);
    return r;
  }
  bool try_swap_out(int running_idx, double now_ms);  // :904-954
  void try_swap_in(double now_ms);                    // :830-901
  // drops the most recently admitted running sequence when nothing can make progress (engine.rs:1103-1120)
  bool abort_one(double now_ms);
  std::string last_error;

 private:
  BlockManager* bm_;
  SchedulerConfig cfg_;
  std::deque<Sequence> waiting_;
  std::vector<Sequence> running_;
  std::vector<Sequence> swapped_;  // the reference keeps these in `cached` with status Swapped
  std::vector<SwapOp> swap_ops_;
  int64_t next_id_ = 1;
  bool is_last_prefill_ = false;
};

double now_ms();

}  // namespace vra
