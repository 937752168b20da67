// core.cpp — see core.h.  Each function cites the reference lines it restates.
#include "core.h"

#include <algorithm>
#include <chrono>

namespace vra {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// PrefixCache (src/core/prefix_cache.rs)
// ---------------------------------------------------------------------------------------------
uint64_t PrefixCache::hash_block(uint64_t parent, const uint32_t* tokens, int n) {
  // prefix_cache.rs:343-348 hashes (parent_hash, tokens) with DefaultHasher; any 64-bit hash with
  // the same chaining is conformant (Appendix A16).
  uint64_t h = 0xcbf29ce484222325ull ^ (parent * 0x9E3779B97F4A7C15ull);
  for (int i = 0; i < n; i++) {
    h ^= tokens[i];
    h *= 0x100000001b3ull;
    h ^= h >> 29;
  }
  h ^= (uint64_t)n << 48;
  h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
  h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
  return h ^ (h >> 31);
}

PrefixCache::Match PrefixCache::match_prefix(const uint32_t* tokens, int n) {
  Match m;
  if (!enabled()) return m;
  const int full_blocks = n / block_size_;
  uint64_t parent = 0;
  for (int i = 0; i < full_blocks; i++) {
    uint64_t h = hash_block(parent, tokens + (size_t)i * block_size_, block_size_);
    if (entries_.count(h)) {
      m.matched_blocks++;
      parent = h;
      m.has_hash = true;
      m.last_hash = h;
      touch(h);
    } else {
      break;
    }
  }
  return m;
}

std::vector<int> PrefixCache::blocks_for_match(uint64_t last_hash) const {
  std::vector<int> blocks;
  bool has = true;
  uint64_t cur = last_hash;
  while (has) {
    auto it = entries_.find(cur);
    if (it == entries_.end()) break;
    blocks.push_back(it->second.block_id);
    has = it->second.has_parent;
    cur = it->second.parent;
  }
  std::reverse(blocks.begin(), blocks.end());
  return blocks;
}

PrefixCache::Update PrefixCache::insert_prefix(const uint32_t* tokens, int n, const std::vector<int>& blocks) {
  Update u;
  if (!enabled()) return u;
  const int full_blocks = n / block_size_;
  const int max_blocks = std::min(full_blocks, (int)blocks.size());
  bool has_parent = false;
  uint64_t parent = 0;
  for (int i = 0; i < max_blocks; i++) {
    const uint64_t base = has_parent ? parent : 0;
    const uint64_t h = hash_block(base, tokens + (size_t)i * block_size_, block_size_);
    auto it = entries_.find(h);
    if (it != entries_.end()) {
      it->second.access_id = next_access_id();
      touch_leaf(h);
    } else {
      if (has_parent) {
        auto pit = entries_.find(parent);
        if (pit != entries_.end()) {
          if (pit->second.children == 0) leaf_set_.erase(parent);
          pit->second.children++;
        }
      }
      const uint64_t aid = next_access_id();
      entries_[h] = Entry{has_parent, parent, blocks[i], 0, aid};
      leaf_set_.insert(h);
      leaf_lru_.push_back({h, aid});
      u.inserted.push_back(blocks[i]);
    }
    has_parent = true;
    parent = h;
  }
  const int excess = (int)entries_.size() - max_cached_blocks_;
  if (excess > 0) u.evicted = evict_blocks(excess);
  return u;
}

std::vector<int> PrefixCache::evict_blocks(int num_blocks) {
  std::vector<int> evicted;
  while (num_blocks > 0 && !leaf_lru_.empty()) {
    auto [h, aid] = leaf_lru_.front();
    leaf_lru_.pop_front();
    if (!leaf_set_.count(h)) continue;
    auto it = entries_.find(h);
    if (it == entries_.end()) continue;
    if (it->second.access_id != aid || it->second.children > 0) continue;
    Entry e = it->second;
    entries_.erase(it);
    leaf_set_.erase(h);
    evicted.push_back(e.block_id);
    num_blocks--;
    if (e.has_parent) {
      auto pit = entries_.find(e.parent);
      if (pit != entries_.end()) {
        if (pit->second.children > 0) pit->second.children--;
        if (pit->second.children == 0) {
          leaf_set_.insert(e.parent);
          leaf_lru_.push_back({e.parent, pit->second.access_id});
        }
      }
    }
  }
  return evicted;
}

void PrefixCache::touch(uint64_t h) {
  auto it = entries_.find(h);
  if (it == entries_.end()) return;
  it->second.access_id = next_access_id();
  touch_leaf(h);
}
void PrefixCache::touch_leaf(uint64_t h) {
  if (leaf_set_.count(h)) {
    auto it = entries_.find(h);
    if (it != entries_.end()) leaf_lru_.push_back({h, it->second.access_id});
  }
  compact_lru_if_needed();
}
void PrefixCache::compact_lru_if_needed() {
  const size_t threshold = std::max<size_t>(entries_.size(), 64) * 4;
  if (leaf_lru_.size() <= threshold) return;
  std::deque<std::pair<uint64_t, uint64_t>> keep;
  for (auto& p : leaf_lru_) {
    if (!leaf_set_.count(p.first)) continue;
    auto it = entries_.find(p.first);
    if (it != entries_.end() && it->second.access_id == p.second) keep.push_back(p);
  }
  leaf_lru_.swap(keep);
}

// ---------------------------------------------------------------------------------------------
// BlockManager (src/core/block_manager.rs)
// ---------------------------------------------------------------------------------------------
BlockManager::BlockManager(int num_blocks, int block_size, bool prefix_cache, float prefix_fraction, int num_cpu_blocks)
    : block_size_(block_size),
      ref_(num_blocks, 0),
      next_(num_blocks, -1),
      prev_(num_blocks, -1),
      in_free_(num_blocks, 0),
      cache_(block_size, prefix_cache, (int)((double)num_blocks * (prefix_fraction > 0 ? prefix_fraction : 0.65f))) {
  for (int i = 0; i < num_blocks; i++) push_back(i);  // block_manager.rs:62-68 — FIFO 0..n-1
  num_cpu_blocks_ = std::max(0, num_cpu_blocks);
  for (int i = 0; i < num_cpu_blocks_; i++) free_cpu_.push_back(i);
}
void BlockManager::push_back(int id) {
  prev_[id] = tail_;
  next_[id] = -1;
  if (tail_ >= 0) next_[tail_] = id;
  else head_ = id;
  tail_ = id;
  in_free_[id] = 1;
  free_count_++;
}
void BlockManager::unlink(int id) {
  if (!in_free_[id]) return;
  const int p = prev_[id], n = next_[id];
  if (p >= 0) next_[p] = n;
  else head_ = n;
  if (n >= 0) prev_[n] = p;
  else tail_ = p;
  in_free_[id] = 0;
  free_count_--;
}
int BlockManager::pop_front() {
  const int id = head_;
  if (id >= 0) unlink(id);
  return id;
}
void BlockManager::allocate_block(int id) {  // :113-120 (ref_count must be 0)
  ref_[id] = 1;
  unlink(id);
}
void BlockManager::increment_ref(int id) {
  if (ref_[id] == 0) unlink(id);
  ref_[id]++;
}
void BlockManager::decrement_ref(int id) {
  if (ref_[id] > 0) ref_[id]--;
  if (ref_[id] == 0 && !in_free_[id]) push_back(id);  // :139-144 deallocate_block
}
int BlockManager::adjusted_matched_blocks(int tokens_len, int m) const {
  // :291-299 — a fully cached, block-aligned prompt still recomputes its last block (Appendix A17)
  const int full_blocks = tokens_len / block_size_;
  if (m == full_blocks && tokens_len % block_size_ == 0 && m > 0) return m - 1;
  return m;
}
int BlockManager::required_blocks(const Sequence& s) {
  if (cache_.enabled()) {
    auto m = cache_.match_prefix(s.token_ids.data(), s.len());
    const int matched = adjusted_matched_blocks(s.len(), m.matched_blocks);
    return std::max(0, s.num_blocks() - matched);
  }
  return s.num_blocks();
}
bool BlockManager::can_allocate(const Sequence& s) { return free_count_ >= required_blocks(s); }
bool BlockManager::allocate_fresh(Sequence& s) {
  s.num_cached_tokens = 0;
  const int need = s.num_blocks();
  if (free_count_ < need) return false;
  for (int i = 0; i < need; i++) {
    const int id = pop_front();
    ref_[id] = 1;
    s.block_table.push_back((uint32_t)id);
  }
  return true;
}
bool BlockManager::allocate_with_prefix(Sequence& s) {  // :346-442
  int matched = 0;
  PrefixCache::Match m = cache_.match_prefix(s.token_ids.data(), s.len());
  matched = adjusted_matched_blocks(s.len(), m.matched_blocks);
  if (matched > 0 && m.has_hash) {
    std::vector<int> cached = cache_.blocks_for_match(m.last_hash);
    if ((int)cached.size() > matched) cached.resize(matched);
    if (free_count_ < s.num_blocks() - (int)cached.size()) return false;
    for (int id : cached) {
      increment_ref(id);
      s.block_table.push_back((uint32_t)id);
    }
    matched = (int)cached.size();
  } else {
    matched = 0;
    if (free_count_ < s.num_blocks()) return false;
  }
  s.num_cached_tokens = matched * block_size_;
  for (int i = (int)s.block_table.size(); i < s.num_blocks(); i++) {
    const int id = pop_front();
    if (id < 0) return false;
    ref_[id] = 1;
    s.block_table.push_back((uint32_t)id);
  }
  return true;
}
bool BlockManager::allocate(Sequence& s) {
  if (cache_.enabled()) return allocate_with_prefix(s);
  return allocate_fresh(s);
}
void BlockManager::deallocate(const Sequence& s) {
  for (auto it = s.block_table.rbegin(); it != s.block_table.rend(); ++it) decrement_ref((int)*it);
}
bool BlockManager::can_append(const Sequence& s) const {
  int need = 1;
  if (s.len() % block_size_ != 0) need++;
  return free_count_ >= need;
}
bool BlockManager::may_append(Sequence& s) {
  // approaching next block (:245-253).  Deliberate fix of a reference defect: the reference pushes a block
  // whenever len % BS == 1 — also for a sequence that was swapped out at such a length before it was scheduled: swap-out and
  // swap-in both run ensure_allocate (scheduler.rs:886,917-921), so its table already HAS the block, the second one makes
  // `block_table.last()` (the decode slot, runner.rs:1259-1262) point at a block the attention never reads.  A block is
  // allocated only while the table is shorter than the sequence needs — identical in every other situation.
  if (s.len() % block_size_ == 1 && (int)s.block_table.size() < s.num_blocks()) {
    const int id = pop_front();
    if (id < 0) return false;
    ref_[id] = 1;
    s.block_table.push_back((uint32_t)id);
  }
  return true;
}
void BlockManager::cache_sequence(const Sequence& s) {
  if (!cache_.enabled()) return;
  const int full_blocks = s.len() / block_size_;
  if (full_blocks == 0 || (int)s.block_table.size() < full_blocks) return;
  std::vector<int> blocks(s.block_table.begin(), s.block_table.begin() + full_blocks);
  auto u = cache_.insert_prefix(s.token_ids.data(), s.len(), blocks);
  for (int id : u.inserted) increment_ref(id);
  for (int id : u.evicted) decrement_ref(id);
}
int BlockManager::evict_prefix_cache(int n) {
  auto ev = cache_.evict_blocks(n);
  for (int id : ev) decrement_ref(id);
  return (int)ev.size();
}

// ---- CPU swap space (block_manager.rs:870-1010)
bool BlockManager::can_swap_out(const Sequence& s) const {
  for (uint32_t id : s.block_table)
    if (ref_[id] > 1) return false;  // a block shared with the prefix cache or another sequence stays on the GPU
  return (int)free_cpu_.size() > s.num_blocks();
}
bool BlockManager::can_swap_in(const Sequence& s) const { return free_count_ > s.num_blocks(); }
bool BlockManager::ensure_allocate(Sequence& s) {
  while ((int)s.block_table.size() < s.num_blocks()) {
    const int id = pop_front();
    if (id < 0) return false;
    ref_[id] = 1;
    s.block_table.push_back((uint32_t)id);
  }
  return true;
}
bool BlockManager::swap_out(const Sequence& s, std::vector<std::pair<int, int>>* gpu_to_cpu) {
  if (free_cpu_.size() < s.block_table.size()) return false;
  std::vector<int> cpu_ids;
  cpu_ids.reserve(s.block_table.size());
  for (uint32_t g : s.block_table) {
    const int c = free_cpu_.front();
    free_cpu_.pop_front();
    cpu_ids.push_back(c);
    gpu_to_cpu->push_back({(int)g, c});
  }
  swapped_map_[s.id] = std::move(cpu_ids);
  return true;
}
bool BlockManager::swap_in(const Sequence& s, std::vector<std::pair<int, int>>* cpu_to_gpu) {
  auto it = swapped_map_.find(s.id);
  if (it == swapped_map_.end()) return false;
  if (it->second.size() > s.block_table.size()) {  // :964-969: not enough GPU blocks behind the table: give the space back
    free_cpu_swap_for_seq(s.id);
    return false;
  }
  for (size_t i = 0; i < it->second.size(); i++) cpu_to_gpu->push_back({it->second[i], (int)s.block_table[i]});
  for (int c : it->second) free_cpu_.push_back(c);
  swapped_map_.erase(it);
  return true;
}
void BlockManager::free_cpu_swap_for_seq(int64_t seq_id) {
  auto it = swapped_map_.find(seq_id);
  if (it == swapped_map_.end()) return;
  for (int c : it->second) free_cpu_.push_back(c);
  swapped_map_.erase(it);
}
int BlockManager::evict_prefix_cache_until_free(int required_free) {
  int evicted = 0;
  while (free_count_ < required_free && cache_.cached_blocks() > 0) {
    const int n = evict_prefix_cache(std::max(1, required_free - free_count_));
    if (n == 0) break;
    evicted += n;
  }
  return evicted;
}

// ---------------------------------------------------------------------------------------------
// Scheduler (src/core/scheduler.rs)
// ---------------------------------------------------------------------------------------------
static const int kMinScheduledReqs = 5;      // scheduler.rs:44
static const float kSwapThreshold = 0.95f;   // scheduler.rs:48

int64_t Scheduler::add(Sequence&& s) {
  // engine.rs:520-528 — the prompt must leave room for at least one generated token
  if (cfg_.max_model_len > 0 && s.len() > cfg_.max_model_len - 1) {
    last_error = "prompt longer than max_model_len - 1";
    return -1;
  }
  if (s.len() == 0) {
    last_error = "empty prompt";
    return -1;
  }
  s.id = next_id_++;
  s.status = SeqStatus::Waiting;
  s.block_size = cfg_.block_size;
  s.prompt_len = s.len();
  s.last_token = s.token_ids.back();
  s.created_ms = now_ms();
  waiting_.push_back(std::move(s));
  return waiting_.back().id;
}

std::vector<int> Scheduler::schedule(bool* is_prefill) {
  std::vector<int> scheduled;
  int num_tokens = 0;
  const int CHUNK = cfg_.prefill_chunk;
  const int pre_existing_running = (int)running_.size();
  const int max_seqs_limit = std::max(cfg_.max_num_seqs, kMinScheduledReqs);
  int step_tokens = 0;  // tokens this step really carries (prefix-cache hits known): what the activation buffers must hold
  while (!waiting_.empty()) {
    Sequence& seq = waiting_.front();
    const int effective = std::min(CHUNK, seq.len() - seq.num_cached_tokens);
    if ((int)running_.size() >= max_seqs_limit || (int)scheduled.size() >= max_seqs_limit ||
        num_tokens + effective >= cfg_.max_num_batched_tokens - 1 ||
        (seq.block_table.empty() && !bm_->can_allocate(seq)) ||
        (is_last_prefill_ && pre_existing_running > 0)) {  // interleave prefill/decode (:262-264)
      break;
    }
    Sequence s = std::move(waiting_.front());
    waiting_.pop_front();
    if (s.block_table.empty()) {
      if (!bm_->allocate(s)) {
        waiting_.push_front(std::move(s));
        break;
      }
    }
    // NOT in the reference: one step carries at most max_step_tokens tokens (the activation buffers are sized for that,
    // the reference's only bound is max_num_batched_tokens = blocks x block_size).  Counted AFTER allocation, when the
    // prefix-cache hit is known: eight prompts behind a cached 16k prefix are 8 x 1024 tokens, not 8 x 8192.  A sequence
    // that does not fit waits at the front of the queue keeping its blocks, like an unfinished chunked prefill (A13).
    const int actual = std::min(CHUNK, s.len() - s.num_cached_tokens);
    if (step_tokens > 0 && step_tokens + actual > cfg_.max_step_tokens) {
      waiting_.push_front(std::move(s));
      break;
    }
    s.status = SeqStatus::Running;
    num_tokens += effective;
    step_tokens += actual;
    running_.push_back(std::move(s));
    scheduled.push_back((int)running_.size() - 1);
  }
  if (!scheduled.empty()) {
    is_last_prefill_ = true;
    *is_prefill = true;
    return scheduled;
  }
  // ---- decode phase (:285-379)
  std::vector<int> decode_ids;
  std::vector<int> preempt_ids;  // running sequences that cannot get their next slot (:288-296)
  for (int idx = 0; idx < (int)running_.size(); idx++)
    if (!bm_->can_append(running_[idx])) preempt_ids.push_back(idx);
  const double now = now_ms();
  const bool swap_space = bm_->num_cpu_blocks() > 0;
  // :303-338.  Swap a sequence back in while there is room; under pressure evict a tenth of the prefix cache, and only when
  // there is nothing to evict swap the OLDEST preempted sequence out (with a single running sequence that makes no sense)
  const bool try_in = swap_space && preempt_ids.empty() &&
                      (bm_->usage() < kSwapThreshold * 0.9f || (running_.empty() && bm_->usage() <= 0.3f));
  if (try_in) {
    try_swap_in(now);
  } else if (!preempt_ids.empty() || bm_->usage() > kSwapThreshold) {
    const int cached = bm_->prefix_cache_blocks();
    const int evicted = cached > 0 ? bm_->evict_prefix_cache(std::max(1, cached / 10)) : 0;
    if (evicted == 0 && swap_space && !preempt_ids.empty() && running_.size() > 1) {
      int oldest = preempt_ids[0];
      for (int i : preempt_ids)
        if (running_[i].id < running_[oldest].id) oldest = i;
      try_swap_out(oldest, now);
    }
  }
  const int decode_max = std::max(cfg_.max_num_seqs, kMinScheduledReqs);
  for (int idx = 0; idx < (int)running_.size(); idx++) {
    if ((int)decode_ids.size() >= decode_max) break;
    Sequence& seq = running_[idx];
    if (seq.status != SeqStatus::Running) continue;  // a failed swap-in parks its (finished) sequence here until collected
    if (!bm_->can_append(seq)) continue;  // unable to acquire resources this step
    if (!bm_->may_append(seq)) continue;
    decode_ids.push_back(idx);
  }
  is_last_prefill_ = false;
  *is_prefill = false;
  return decode_ids;
}

void Scheduler::postprocess(const std::vector<int>& ids, const std::vector<uint32_t>& tokens, double now) {
  for (size_t i = 0; i < ids.size(); i++) {
    const int idx = ids[i];
    if (idx < 0 || idx >= (int)running_.size()) continue;
    Sequence& seq = running_[idx];
    const uint32_t token = tokens[i];
    if (seq.first_token_ms == 0) seq.first_token_ms = now;  // engine.rs:1004-1012
    const bool is_eos = !seq.ignore_eos && std::find(seq.eos.begin(), seq.eos.end(), token) != seq.eos.end();
    if (is_eos || seq.output_len() >= seq.max_tokens || seq.len() > cfg_.max_num_batched_tokens ||
        (cfg_.max_model_len > 0 && seq.len() >= cfg_.max_model_len)) {
      // :596-627 — the final sampled token is NOT appended (Appendix A2)
      seq.status = SeqStatus::Finished;
      seq.finished_ms = now;
      bm_->cache_sequence(seq);
      bm_->deallocate(seq);
    } else {
      seq.append_token(token);
    }
  }
}

void Scheduler::filter_prefill_finished(const std::vector<int>& ids, std::vector<int>* keep_pos, std::vector<int>* run_idx) {
  const int CHUNK = cfg_.prefill_chunk;
  std::vector<std::pair<int, int64_t>> finished;  // (position in ids, seq id)
  std::vector<int64_t> remove_ids;
  for (size_t i = 0; i < ids.size(); i++) {
    const int id = ids[i];
    if (id >= (int)running_.size()) continue;
    Sequence& seq = running_[id];
    if (seq.len() < CHUNK || seq.num_cached_tokens + CHUNK >= seq.len()) {
      finished.push_back({(int)i, seq.id});
    } else {
      // chunk progress shares num_cached_tokens with the prefix-cache hit (Appendix A13); the
      // sequence goes back to `waiting` KEEPING its block table.
      remove_ids.push_back(seq.id);
      Sequence copy = seq;
      copy.num_cached_tokens += CHUNK;
      copy.status = SeqStatus::Waiting;
      waiting_.push_back(std::move(copy));
    }
  }
  if (!remove_ids.empty()) {
    running_.erase(std::remove_if(running_.begin(), running_.end(),
                                  [&](const Sequence& s) { return std::find(remove_ids.begin(), remove_ids.end(), s.id) != remove_ids.end(); }),
                   running_.end());
  }
  keep_pos->clear();
  run_idx->clear();
  for (auto& f : finished) {
    for (int r = 0; r < (int)running_.size(); r++)
      if (running_[r].id == f.second) {
        keep_pos->push_back(f.first);
        run_idx->push_back(r);
        break;
      }
  }
}

std::vector<Sequence> Scheduler::clear_finished() {
  std::vector<Sequence> done;
  std::vector<Sequence> keep;
  keep.reserve(running_.size());
  for (auto& s : running_) {
    if (s.status == SeqStatus::Finished) done.push_back(std::move(s));
    else keep.push_back(std::move(s));
  }
  running_.swap(keep);
  return done;
}

// :904-954 — one sequence a time; its blocks go back to the free list at once, the copy runs before the next forward
bool Scheduler::try_swap_out(int idx, double now) {
  if (idx < 0 || idx >= (int)running_.size()) return false;
  Sequence& seq = running_[idx];
  if (seq.block_table.empty() || seq.status != SeqStatus::Running || !bm_->can_swap_out(seq)) return false;
  if (!bm_->ensure_allocate(seq)) return false;  // the same number of blocks on the way back in
  SwapOp op;
  op.to_gpu = false;
  op.seq_id = seq.id;
  if (!bm_->swap_out(seq, &op.pairs)) return false;
  swap_ops_.push_back(std::move(op));
  Sequence s = std::move(running_[idx]);
  running_.erase(running_.begin() + idx);
  s.status = SeqStatus::Swapped;
  s.swapped_ms = now;
  bm_->deallocate(s);
  s.block_table.clear();  // reallocated at swap-in (:944)
  swapped_.push_back(std::move(s));
  return true;
}

// :830-901 — at most one sequence per step, and only after the cooling period
void Scheduler::try_swap_in(double now) {
  for (size_t i = 0; i < swapped_.size(); i++) {
    Sequence& c = swapped_[i];
    if (now - c.swapped_ms < (double)cfg_.swap_cooling_ms) continue;
    const long available = (long)bm_->num_free_blocks() * bm_->block_size();
    if (!bm_->can_swap_in(c) || available - c.len() < cfg_.min_tokens_left_for_swap) {
      if (!running_.empty()) continue;  // wait for the running sequences to finish
      if (bm_->evict_prefix_cache_until_free(c.num_blocks() + 1) > 0) break;
      // nothing running, nothing to evict and still no room: the sequence can never come back (:871-876)
      Sequence s = std::move(swapped_[i]);
      swapped_.erase(swapped_.begin() + i);
      bm_->free_cpu_swap_for_seq(s.id);
      s.status = SeqStatus::Finished;
      s.aborted = true;
      s.finished_ms = now;
      last_error = "no KV cache left to swap a sequence back in";
      running_.push_back(std::move(s));
      break;
    }
    Sequence s = std::move(swapped_[i]);
    swapped_.erase(swapped_.begin() + i);
    s.swapped_ms = now;
    s.block_table.clear();
    if (!bm_->ensure_allocate(s)) {  // (the reference drops the sequence here; it goes back to the swapped list instead)
      bm_->deallocate(s);
      s.block_table.clear();
      swapped_.insert(swapped_.begin() + i, std::move(s));
      continue;
    }
    SwapOp op;
    op.to_gpu = true;
    op.seq_id = s.id;
    if (bm_->swap_in(s, &op.pairs)) {
      swap_ops_.push_back(std::move(op));
      s.status = SeqStatus::Running;
    } else {
      bm_->deallocate(s);
      s.status = SeqStatus::Finished;
      s.aborted = true;
      s.finished_ms = now;
      last_error = "swap-in failed";
    }
    running_.push_back(std::move(s));
    break;
  }
}

bool Scheduler::abort_one(double now) {
  if (running_.empty()) {
    if (waiting_.empty() && !swapped_.empty()) {  // a swapped-out sequence that will never fit
      Sequence s = std::move(swapped_.back());
      swapped_.pop_back();
      bm_->free_cpu_swap_for_seq(s.id);
      s.status = SeqStatus::Finished;
      s.aborted = true;
      s.finished_ms = now;
      running_.push_back(std::move(s));
      return true;
    }
    if (waiting_.empty()) return false;
    Sequence s = std::move(waiting_.back());
    waiting_.pop_back();
    s.status = SeqStatus::Finished;
    s.aborted = true;
    s.finished_ms = now;
    if (!s.block_table.empty()) bm_->deallocate(s);
    running_.push_back(std::move(s));
    return true;
  }
  // newest running sequence gives its blocks back
  Sequence& s = running_.back();
  s.status = SeqStatus::Finished;
  s.aborted = true;
  s.finished_ms = now;
  bm_->deallocate(s);
  return true;
}

}  // namespace vra
