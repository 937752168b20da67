// sampler.hip — device-side stochastic sampling and repetition penalties.
// Restates LogitsProcessor::{sample_with_strategy, sample_topk_topp, sample_topk, sample_topp, apply_penalties}
// (src/utils/logits_processor.rs:199-345) and the call sequence of ModelRunner::sample (src/core/runner.rs:1519-1547):
//   probabilities = softmax(logits / temperature) over the full vocabulary (f32);
//   top-k: the k most probable tokens in descending order (ties: lower token id first, as a stable sort gives), k <= 256
//          — the bound of the reference's own device sampler (`sampler.sample_cuda`, k = 256 stands in for "top-p only");
//   top-p: walking that order, a candidate is KEPT while the mass accumulated BEFORE it is < p
//          (`if cumsum >= top_p { prs[i] = 0 } else { cumsum += prs[i] }`), all kept when p <= 0 or p >= Σ top-k mass;
//   draw:  one uniform u in [0, Σ kept mass), the first candidate whose running sum exceeds u (WeightedIndex).
// The uniform comes from the counter hash of common.cuh (seed, row) — the reference draws from StdRng (ChaCha12) on the
// host and hands a fresh u64 to its device sampler per call; only the distribution is contract, not the stream.
// One workgroup per row.  Top-k by exact radix select on order-preserving keys (4 passes of 8 bits over the row, which is
// L2 resident: V * 4 bytes = 0.5 MB), candidates gathered in token order, then a bitonic sort of 256 entries in LDS.
#include "common.cuh"

#define SMP_THREADS 1024
#define SMP_MAXK 256

__device__ __forceinline__ uint32_t smp_key(float v) {  // order-preserving float -> u32 (larger value = larger key)
  const uint32_t u = __float_as_uint(v + 0.0f);  // -0.0 and +0.0 compare equal in a float sort: one key for both
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// block-wide f32 max / sum through LDS (fixed order => deterministic)
__device__ __forceinline__ float smp_block_reduce(float v, float* red, bool is_max) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < SMP_THREADS / 64; w++) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

struct SampleArgs {
  const float* logits;  // [B, V]
  uint32_t* out;        // [B]
  int V, k;             // k = 0: sample from the whole distribution (Sampling::All)
  float top_p, inv_temperature;
  uint64_t seed;
  uint32_t* dbg_idx;  // [B, SMP_MAXK] candidate token ids in order (kept ones), 0xffffffff beyond; may be null
  float* dbg_prob;    // [B, SMP_MAXK] their probabilities (0 for the ones top-p dropped); may be null
};

__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const SampleArgs a) {
  __shared__ float red[SMP_THREADS / 64];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_remaining, s_count;
  __shared__ uint32_t c_key[SMP_MAXK], c_idx[SMP_MAXK];
  __shared__ float c_prob[SMP_MAXK];
  __shared__ float chunk_sum[SMP_THREADS];
  const int tid = threadIdx.x, row = blockIdx.x, V = a.V;
  const float* lg = a.logits + (size_t)row * V;
  const float it = a.inv_temperature;
  // ---- softmax statistics of logits / temperature
  float mx = -INFINITY;
  for (int i = tid; i < V; i += SMP_THREADS) mx = fmaxf(mx, lg[i] * it);
  mx = smp_block_reduce(mx, red, true);
  float sm = 0.f;
  for (int i = tid; i < V; i += SMP_THREADS) sm += expf(lg[i] * it - mx);
  sm = smp_block_reduce(sm, red, false);
  const float inv_sum = 1.0f / sm;
  const float u01 = vra_hash_unit(a.seed, (uint64_t)row);  // [0, 1)

  if (a.k <= 0) {
    // ---- Sampling::All: inverse CDF over the whole vocabulary in token order
    const int per = (V + SMP_THREADS - 1) / SMP_THREADS;
    const int i0 = min(tid * per, V), i1 = min(i0 + per, V);
    float cs = 0.f;
    for (int i = i0; i < i1; i++) cs += expf(lg[i] * it - mx) * inv_sum;
    chunk_sum[tid] = cs;
    __syncthreads();
    if (tid == 0) {
      float total = 0.f;
      for (int t = 0; t < SMP_THREADS; t++) total += chunk_sum[t];
      const float u = u01 * total;
      float run = 0.f;
      int t = 0;
      for (; t < SMP_THREADS - 1 && run + chunk_sum[t] <= u; t++) run += chunk_sum[t];
      const int j0 = min(t * per, V), j1 = min(j0 + per, V);
      int pick = j1 > j0 ? j1 - 1 : V - 1;
      for (int i = j0; i < j1; i++) {
        run += expf(lg[i] * it - mx) * inv_sum;
        if (run > u) {
          pick = i;
          break;
        }
      }
      a.out[row] = (uint32_t)pick;
    }
    return;
  }

  // ---- top-k: the k-th largest key by radix select (keys of the SCALED logits: softmax is monotone)
  const int k = min(a.k, min(V, SMP_MAXK));
  if (tid == 0) {
    s_prefix = 0u;
    s_remaining = (uint32_t)k;
  }
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const uint32_t prefix = s_prefix, mask_hi = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < V; i += SMP_THREADS) {
      const uint32_t key = smp_key(lg[i] * it);
      if ((key & mask_hi) == prefix) atomicAdd(&hist[(key >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (tid == 0) {  // walk the bins from the top: the bin holding the remaining-th largest
      uint32_t rem = s_remaining;
      int b = 255;
      for (; b > 0; b--) {
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      s_prefix = prefix | ((uint32_t)b << shift);
      s_remaining = rem;
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;       // key of the k-th largest element
  const uint32_t need_ties = s_remaining;  // how many elements equal to it belong to the top k (lowest token ids first)
  // ---- gather: everything above the k-th key, and the first `need_ties` elements equal to it in token order
  if (tid == 0) s_count = 0u;
  __syncthreads();
  {
    // ties must be taken in token order: one thread walks them per chunk in order — chunks are processed by increasing
    // token id with a running count, so do the (rare) tie bookkeeping serially per chunk boundary
    const int per = (V + SMP_THREADS - 1) / SMP_THREADS;
    const int i0 = min(tid * per, V), i1 = min(i0 + per, V);
    uint32_t my_ties = 0u;
    for (int i = i0; i < i1; i++) my_ties += smp_key(lg[i] * it) == kth ? 1u : 0u;
    reinterpret_cast<uint32_t*>(chunk_sum)[tid] = my_ties;
    __syncthreads();
    uint32_t before = 0u;  // ties in chunks of lower token ids (fixed order)
    for (int t = 0; t < tid; t++) before += reinterpret_cast<uint32_t*>(chunk_sum)[t];
    for (int i = i0; i < i1; i++) {
      const float sv = lg[i] * it;
      const uint32_t key = smp_key(sv);
      bool take = key > kth;
      if (key == kth) {
        take = before < need_ties;
        before++;
      }
      if (take) {
        const uint32_t slot = atomicAdd(&s_count, 1u);
        if (slot < SMP_MAXK) {
          c_key[slot] = key;
          c_idx[slot] = (uint32_t)i;
          c_prob[slot] = expf(sv - mx) * inv_sum;
        }
      }
    }
  }
  __syncthreads();
  const int n = (int)min(s_count, (uint32_t)SMP_MAXK);  // == k
  for (int i = n + tid; i < SMP_MAXK; i += SMP_THREADS) {  // pad for the bitonic network: sorts last
    c_key[i] = 0u;
    c_idx[i] = 0xFFFFFFFFu;
    c_prob[i] = 0.f;
  }
  __syncthreads();
  // ---- bitonic sort of 256 entries: descending key, ascending token id among equal keys
  for (int size = 2; size <= SMP_MAXK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < SMP_MAXK / 2) {
        const int lo = (tid / stride) * (stride << 1) + (tid % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint32_t ka = c_key[lo], kb = c_key[hi], ia = c_idx[lo], ib = c_idx[hi];
        const bool a_first = ka > kb || (ka == kb && ia < ib);  // "a before b" in the final order
        if (a_first != desc) {
          c_key[lo] = kb, c_key[hi] = ka;
          c_idx[lo] = ib, c_idx[hi] = ia;
          const float pa = c_prob[lo];
          c_prob[lo] = c_prob[hi], c_prob[hi] = pa;
        }
      }
      __syncthreads();
    }
  }
  // ---- top-p over the sorted candidates + the draw (sequential, k <= 256: the reference's loops verbatim)
  if (tid == 0) {
    float sum_p = 0.f;
    for (int i = 0; i < n; i++) sum_p += c_prob[i];
    float total = sum_p;
    if (!(a.top_p <= 0.f || a.top_p >= sum_p)) {
      float cumsum = 0.f;
      total = 0.f;
      for (int i = 0; i < n; i++) {
        if (cumsum >= a.top_p) c_prob[i] = 0.f;
        else cumsum += c_prob[i];
        total += c_prob[i];
      }
    }
    const float u = u01 * total;
    float run = 0.f;
    int pick = 0;
    for (int i = 0; i < n; i++) {
      if (c_prob[i] > 0.f) pick = i;  // the last kept candidate catches u == total after rounding
      run += c_prob[i];
      if (run > u) {
        pick = i;
        break;
      }
    }
    a.out[row] = c_idx[pick];
  }
  __syncthreads();
  if (a.dbg_idx) {
    for (int i = tid; i < SMP_MAXK; i += SMP_THREADS) {
      a.dbg_idx[(size_t)row * SMP_MAXK + i] = i < n ? c_idx[i] : 0xFFFFFFFFu;
      a.dbg_prob[(size_t)row * SMP_MAXK + i] = i < n ? c_prob[i] : 0.f;
    }
  }
}

// LogitsProcessor::sample_with_strategy for the stochastic strategies (logits_processor.rs:199-271):
//   top_k > 0, top_p in (0,1): TopKThenTopP; top_k > 0, top_p <= 0 or >= 1: TopK; top_k == 0 with top_p in (0,1): TopP (k = 256);
//   top_k == 0 and no top_p: All.  temperature must be > 0 (temperature 0 is ArgMax: vra_argmax_f32).
extern "C" void vra_sample(const float* logits, uint32_t* out, int32_t rows, int32_t vocab, int32_t top_k, float top_p, float temperature,
                           uint64_t seed, uint32_t* dbg_idx, float* dbg_prob, int64_t stream) {
  VRA_CHECK_ARG(logits && out, "vra_sample: null pointer");
  VRA_CHECK_ARG(rows >= 0 && vocab > 0, "vra_sample: bad shape");
  VRA_CHECK_ARG(temperature > 0.f, "vra_sample: temperature must be > 0 (greedy decoding is vra_argmax_f32)");
  VRA_CHECK_ARG(top_k <= SMP_MAXK, "vra_sample: top_k <= 256 (the bound of the reference's device sampler)");
  if (rows == 0) return;
  SampleArgs a;
  a.logits = logits, a.out = out, a.V = vocab;
  const bool has_p = top_p > 0.f && top_p < 1.f;
  a.k = top_k > 0 ? top_k : (has_p ? SMP_MAXK : 0);
  a.top_p = has_p ? top_p : 1.0f;
  a.inv_temperature = 1.0f / temperature;
  a.seed = seed;
  a.dbg_idx = dbg_idx, a.dbg_prob = dbg_prob;
  sample_kernel<<<rows, SMP_THREADS, 0, as_stream(stream)>>>(a);
}

// LogitsProcessor::apply_penalties (logits_processor.rs:288-306): logit -= count * frequency_penalty + (count > 0) * presence_penalty,
// counts over the row's context tokens (ids >= vocab ignored).  One workgroup per row, context of up to 1024 tokens.
__global__ __launch_bounds__(256) void penalties_kernel(float* logits, const uint32_t* ctx, const int32_t* ctx_len, int max_ctx, int V,
                                                        const float* freq, const float* pres) {
  const int row = blockIdx.x, n = min(ctx_len[row], max_ctx);
  const uint32_t* c = ctx + (size_t)row * max_ctx;
  float* lg = logits + (size_t)row * V;
  const float fp = freq[row], pp = pres[row];
  // apply_batch_repeat_penalty's guard: context longer than one token and a penalty that is neither 0 nor 1
  if (n <= 1 || !((fp != 1.0f && fp != 0.f) || (pp != 1.0f && pp != 0.f))) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t tok = c[i];
    if (tok >= (uint32_t)V) continue;
    int cnt = 0;
    bool first = true;
    for (int j = 0; j < n; j++) {
      if (c[j] == tok) {
        cnt++;
        if (j < i) first = false;
      }
    }
    if (first) lg[tok] = lg[tok] - (float)cnt * fp - pp;  // one writer per distinct token
  }
}
extern "C" void vra_apply_penalties(float* logits, const uint32_t* context, const int32_t* context_lens, int32_t rows, int32_t max_context,
                                    int32_t vocab, const float* frequency_penalties, const float* presence_penalties, int64_t stream) {
  VRA_CHECK_ARG(logits && context && context_lens && frequency_penalties && presence_penalties, "vra_apply_penalties: null pointer");
  VRA_CHECK_ARG(max_context > 0 && max_context <= 1024, "vra_apply_penalties: 1..1024 context tokens per row");
  if (rows <= 0) return;
  penalties_kernel<<<rows, 256, 0, as_stream(stream)>>>(logits, context, context_lens, max_context, vocab, frequency_penalties, presence_penalties);
}
