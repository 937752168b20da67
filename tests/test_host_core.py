"""CPU parity tests of the native HOST runtime (vllm_rs_amd/host/*.cpp) through the C ABI — no GPU.

* prefix cache: the reference's own two unit tests (src/core/prefix_cache.rs:362-403) replayed verbatim
* block manager: allocation / ref counting / prefix reuse traces (block_manager.rs:113-442)
* scheduler + metadata: chunked prefill, prefill/decode alternation, slot arithmetic, finish rules
  (scheduler.rs:200-380,500-629; runner.rs:978-1388) on a host-only engine (device = -1)
* KV plan, rotary tables and the Marlin scale permutation against the oracle / closed forms
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from vllm_rs_amd import _lib  # noqa: E402
from vllm_rs_amd import engine as E  # noqa: E402

L = _lib.load()


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


# ------------------------------------------------------------------------------------------------ prefix cache
class PC:
    def __init__(self, block_size, max_cached):
        self.h = L.vra_pc_create(block_size, max_cached)

    def insert(self, tokens, blocks):
        t, b = u32(tokens), np.ascontiguousarray(blocks, dtype=np.int32)
        ev = np.zeros(64, np.int32)
        n_ev = C.c_int32(0)
        ins = L.vra_pc_insert_prefix(self.h, t.ctypes.data, len(t), b.ctypes.data, len(b), ev.ctypes.data, 64, C.byref(n_ev))
        return ins, ev[:n_ev.value].tolist()

    def match(self, tokens):
        t = u32(tokens)
        out = np.zeros(64, np.int32)
        n = L.vra_pc_match_prefix(self.h, t.ctypes.data, len(t), out.ctypes.data, 64)
        return n, out[:n].tolist()

    def __del__(self):
        L.vra_pc_destroy(self.h)


def test_prefix_cache_matches_full_blocks():
    """prefix_cache.rs:362-384, same numbers."""
    c = PC(4, 8)
    ins, ev = c.insert([1, 2, 3, 4, 5, 6, 7, 8], [10, 11])
    assert ev == [] and ins == 2
    n, blocks = c.match([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
    assert n == 2 and blocks == [10, 11]


def test_prefix_cache_evicts_leaf_blocks():
    """prefix_cache.rs:386-403, same numbers."""
    c = PC(4, 1)
    ins, ev = c.insert([1, 2, 3, 4, 5, 6, 7, 8], [21, 22])
    assert ev == [22]
    n, _ = c.match([1, 2, 3, 4, 5, 6, 7, 8])
    assert n == 1


def test_prefix_cache_partial_block_and_divergence():
    c = PC(4, 16)
    c.insert(list(range(12)), [3, 4, 5])
    assert c.match(list(range(11)))[0] == 2          # only FULL blocks match
    assert c.match([0, 1, 2, 3, 9, 9, 9, 9])[0] == 1  # chain breaks at the first differing block
    assert c.match([9, 1, 2, 3])[0] == 0
    # hash chaining: the same 4 tokens under a different parent are a different entry (prefix_cache.rs:343-355)
    assert c.match([4, 5, 6, 7])[0] == 0
    # inserting a sibling branch keeps the shared parent
    ins, ev = c.insert([0, 1, 2, 3, 40, 41, 42, 43], [3, 9])
    assert ins == 1 and ev == []
    assert c.match([0, 1, 2, 3, 40, 41, 42, 43])[1] == [3, 9]
    assert L.vra_pc_cached_blocks(c.h) == 4


def test_prefix_cache_lru_evicts_leaves_only():
    c = PC(2, 16)
    c.insert([1, 2, 3, 4, 5, 6], [0, 1, 2])      # chain a: 0 <- 1 <- 2
    c.insert([1, 2, 7, 8], [0, 3])               # branch b shares block 0: leaf 3
    c.match([1, 2, 3, 4, 5, 6])                  # touch chain a -> leaf 3 is now least recent
    ev = np.zeros(8, np.int32)
    n = L.vra_pc_evict_blocks(c.h, 1, ev.ctypes.data, 8)
    assert n == 1 and ev[0] == 3                 # a leaf, never the shared interior block 0
    n = L.vra_pc_evict_blocks(c.h, 8, ev.ctypes.data, 8)
    assert sorted(ev[:n].tolist()) == [0, 1, 2]  # leaves first, parents as they become leaves
    assert ev[:n].tolist().index(2) < ev[:n].tolist().index(1) < ev[:n].tolist().index(0)


# ------------------------------------------------------------------------------------------------ block manager
class BM:
    def __init__(self, nb, bs, prefix=False, frac=0.65):
        self.h = L.vra_bm_create(nb, bs, int(prefix), frac)
        self.bs = bs

    def seq(self, toks):
        t = u32(toks)
        return L.vra_bm_seq_create(self.h, t.ctypes.data, len(t))

    def table(self, s):
        out = np.zeros(256, np.uint32)
        n = L.vra_bm_seq_block_table(self.h, s, out.ctypes.data, 256)
        return out[:n].tolist()

    def free(self):
        return L.vra_bm_num_free_blocks(self.h)

    def __del__(self):
        L.vra_bm_destroy(self.h)


def test_block_manager_fifo_allocation_and_reuse():
    """free list is FIFO: blocks come out in id order and freed blocks are reused last (block_manager.rs:62-68,126-131)."""
    bm = BM(8, 4)
    a = bm.seq(range(10))           # ceil(10/4) = 3 blocks (sequence.rs:203-213)
    assert L.vra_bm_can_allocate(bm.h, a) == 1
    assert L.vra_bm_allocate(bm.h, a) == 0
    assert bm.table(a) == [0, 1, 2] and bm.free() == 5
    b = bm.seq(range(100, 105))
    L.vra_bm_allocate(bm.h, b)
    assert bm.table(b) == [3, 4] and bm.free() == 3
    L.vra_bm_deallocate(bm.h, a)
    assert bm.free() == 6
    c = bm.seq(range(200, 216))     # 4 blocks: the never-used 5,6,7 first, then the oldest freed
    L.vra_bm_allocate(bm.h, c)
    t = bm.table(c)
    assert t[:3] == [5, 6, 7] and t[3] in (0, 1, 2)
    d = bm.seq(range(300, 340))     # 10 blocks > free
    assert L.vra_bm_can_allocate(bm.h, d) == 0 and L.vra_bm_allocate(bm.h, d) == -1


def test_block_manager_may_append_allocates_on_first_token_of_a_block():
    """may_append allocates when len % BS == 1 (block_manager.rs:244-256)."""
    bm = BM(4, 4)
    s = bm.seq(range(4))            # exactly one full block
    L.vra_bm_allocate(bm.h, s)
    assert bm.table(s) == [0]
    L.vra_bm_append_token(bm.h, s, 7)   # len 5: first token of block 1
    assert L.vra_bm_may_append(bm.h, s) == 0 and bm.table(s) == [0, 1]
    for t in range(3):                  # len 6,7,8: same block
        L.vra_bm_append_token(bm.h, s, t)
        assert L.vra_bm_may_append(bm.h, s) == 0 and bm.table(s) == [0, 1]
    L.vra_bm_append_token(bm.h, s, 9)   # len 9
    assert L.vra_bm_may_append(bm.h, s) == 0 and bm.table(s) == [0, 1, 2]
    assert L.vra_bm_seq_len(bm.h, s) == 9


def test_block_manager_prefix_reuse_and_last_block_recompute():
    bm = BM(16, 4, prefix=True, frac=1.0)
    prompt = list(range(1, 14))         # 13 tokens: 3 full blocks + 1
    a = bm.seq(prompt)
    assert L.vra_bm_allocate(bm.h, a) == 0
    ta = bm.table(a)
    L.vra_bm_deallocate(bm.h, a)        # caches the 3 full blocks (scheduler.rs:619-621)
    assert L.vra_bm_prefix_cached_blocks(bm.h) == 3
    b = bm.seq(prompt + [99, 98])
    cached = L.vra_bm_allocate(bm.h, b)
    assert cached == 12 and bm.table(b)[:3] == ta[:3]   # shared physical blocks
    assert L.vra_bm_seq_num_cached_tokens(bm.h, b) == 12
    # a fully cached, block-aligned prompt still recomputes its last full block (block_manager.rs:291-299, Appendix A17)
    c = bm.seq(prompt[:12])
    cached = L.vra_bm_allocate(bm.h, c)
    assert cached == 8 and bm.table(c)[:2] == ta[:2]
    # eviction under pressure frees only unreferenced cached blocks
    L.vra_bm_deallocate(bm.h, b)
    L.vra_bm_deallocate(bm.h, c)
    before = bm.free()
    ev = L.vra_bm_evict_prefix(bm.h, 2)
    assert ev == 2 and bm.free() == before + 2 or bm.free() >= before


# ------------------------------------------------------------------------------------------------ scheduler + metadata
TINY = dict(E.TINYLLAMA, max_position_embeddings=8192)


def run_to_completion(h, sampler, max_steps=10000):
    trace = []
    for _ in range(max_steps):
        st = h.schedule()
        if st is None:
            if not h.has_unfinished():
                break
            continue
        trace.append(st)
        h.commit([sampler(st, i) for i in range(st["n_seqs"])])
    return trace


def test_decode_slot_arithmetic_and_padding():
    """slot = block_table.last()*BS + last_block_tokens - 1 (runner.rs:1259-1262); tables right-padded with 0 (A5)."""
    h = E.HostEngine(TINY, num_gpu_blocks=32, block_size=4, max_num_seqs=4, max_model_len=64)
    r = h.add_request(list(range(10, 20)), max_tokens=7)
    trace = run_to_completion(h, lambda st, i: 500 + len(st["ids"]))
    pf = trace[0]
    assert pf["is_prefill"] and pf["ids"].tolist() == list(range(10, 20))
    assert pf["positions"].tolist() == list(range(10)) and pf["slots"].tolist() == list(range(10))
    assert pf["cu_q"].tolist() == [0, 10] and pf["context_lens"].tolist() == [10]
    bt = None
    for k, st in enumerate(trace[1:]):
        assert not st["is_prefill"]
        pos = 10 + k
        assert st["positions"].tolist() == [pos] and st["context_lens"].tolist() == [pos + 1]
        row = st["block_tables"][0]
        nblk = pos // 4 + 1
        assert (row[nblk:] == 0).all()
        assert st["slots"].tolist() == [int(row[nblk - 1]) * 4 + pos % 4]
        if bt is not None:
            assert row[:len(bt)].tolist() == bt          # tables only grow
        bt = row[:nblk].tolist()
    assert h.finished(r)
    # max_tokens = 7: seven tokens appended, the 8th sampled token finishes and is NOT appended (A2, scheduler.rs:596-627)
    assert len(h.output(r)) == 7 and len(trace) == 1 + 7


def test_eos_finishes_and_ignore_eos():
    h = E.HostEngine(TINY, num_gpu_blocks=32, block_size=4, max_num_seqs=4, max_model_len=64)
    a = h.add_request([1, 2, 3], max_tokens=50, eos=[7])
    b = h.add_request([4, 5, 6], max_tokens=5, eos=[7], ignore_eos=True)
    toks = iter([11, 12, 7, 7, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
    run_to_completion(h, lambda st, i: 7 if (len(st["ids"]) and st["context_lens"][i] >= 5) else 11)
    assert h.finished(a) and h.finished(b)
    assert 7 not in h.output(a)                 # EOS itself is not appended
    assert h.output(b).count(7) >= 1 and len(h.output(b)) == 5   # bench-harness ignore_eos (A1)


def test_chunked_prefill_schedule():
    """prompt of 2.5 chunks -> three prefill steps, progress carried by num_cached_tokens, block table kept (A13);
    only the last chunk's sampled token is kept (engine.rs:906-916)."""
    CH = 2048
    h = E.HostEngine(TINY, num_gpu_blocks=128, block_size=64, max_num_seqs=4, max_model_len=8192, prefill_chunk=CH)
    prompt = (np.arange(5000) % 1000 + 1).tolist()
    r = h.add_request(prompt, max_tokens=2)
    trace = run_to_completion(h, lambda st, i: 777)
    pre = [t for t in trace if t["is_prefill"]]
    assert [t["n_tokens"] for t in pre] == [CH, CH, 5000 - 2 * CH]
    for k, t in enumerate(pre):
        assert t["positions"][0] == k * CH and t["positions"][-1] == min(5000, (k + 1) * CH) - 1
        assert t["context_lens"].tolist() == [min(5000, (k + 1) * CH)]
        assert t["ids"].tolist() == prompt[k * CH:(k + 1) * CH]
        assert t["block_tables"][0][:79].tolist() == pre[0]["block_tables"][0][:79].tolist()
        # slots follow the block table with the in-block offset of the chunk start (runner.rs:1020-1038)
        bt = t["block_tables"][0]
        want = [int(bt[p // 64]) * 64 + p % 64 for p in range(k * CH, k * CH + t["n_tokens"])]
        assert t["slots"].tolist() == want
    dec = [t for t in trace if not t["is_prefill"]]
    assert dec[0]["ids"].tolist() == [777] and dec[0]["positions"].tolist() == [5000]
    assert h.output(r) == [777, 777]


def test_prefill_and_decode_never_share_a_step_and_alternate():
    """A14: after a prefill step, decode is forced next when something was already running."""
    h = E.HostEngine(TINY, num_gpu_blocks=64, block_size=4, max_num_seqs=8, max_model_len=64)
    a = h.add_request([1, 2, 3, 4, 5], max_tokens=6)
    kinds = []
    st = h.schedule(); kinds.append(st["is_prefill"]); h.commit([9])
    b = h.add_request([6, 7, 8], max_tokens=6)       # arrives while a is decoding
    c = h.add_request([9, 10, 11, 12], max_tokens=6)
    for _ in range(6):
        st = h.schedule()
        if st is None:
            break
        kinds.append(st["is_prefill"])
        assert st["is_prefill"] or st["n_tokens"] == st["n_seqs"]
        h.commit([9] * st["n_seqs"])
    assert kinds[0] is True
    assert kinds[1] is True or kinds[1] is False
    # the new prompts are prefilled together in one step, then everything decodes as one batch
    pf_steps = [i for i, k in enumerate(kinds) if k]
    assert len(pf_steps) == 2
    assert all(not k for k in kinds[pf_steps[1] + 1:])


def test_batch_decode_metadata_is_per_sequence():
    h = E.HostEngine(TINY, num_gpu_blocks=64, block_size=4, max_num_seqs=8, max_model_len=64)
    lens = [3, 9, 5]
    rids = [h.add_request(list(range(100 * i + 1, 100 * i + 1 + n)), max_tokens=4) for i, n in enumerate(lens)]
    st = h.schedule()
    assert st["is_prefill"] and st["n_seqs"] == 3 and st["cu_q"].tolist() == [0, 3, 12, 17]
    assert st["max_seqlen_q"] == 9 and st["max_context_len"] == 9
    used = set()
    for i, n in enumerate(lens):
        bt = st["block_tables"][i][: (n + 3) // 4].tolist()
        assert not (set(bt) & used), "sequences must not share physical blocks"
        used |= set(bt)
        sl = st["slots"][st["cu_q"][i]:st["cu_q"][i + 1]].tolist()
        assert sl == [bt[p // 4] * 4 + p % 4 for p in range(n)]
    h.commit([50, 51, 52])
    st = h.schedule()
    assert not st["is_prefill"] and st["ids"].tolist() == [50, 51, 52]
    assert st["positions"].tolist() == lens and st["context_lens"].tolist() == [n + 1 for n in lens]
    assert st["requests"] == rids


def test_admission_waits_for_blocks_and_prompt_limit():
    """A18: a request needs blocks for prompt + decode reserve; too long a prompt is rejected up front."""
    h = E.HostEngine(TINY, num_gpu_blocks=6, block_size=4, max_num_seqs=4, max_model_len=24)
    with pytest.raises(RuntimeError):
        h.add_request(list(range(1, 40)), max_tokens=4)            # > max_model_len - 1
    a = h.add_request(list(range(1, 13)), max_tokens=3)              # 3 blocks + reserve
    b = h.add_request(list(range(21, 33)), max_tokens=3)
    st = h.schedule()
    assert st["is_prefill"] and st["n_seqs"] == 1 and st["requests"] == [a]   # b does not fit next to a
    h.commit([5])
    trace = run_to_completion(h, lambda st, i: 5)
    assert [t["requests"] for t in trace if t["is_prefill"]] == [[b]]          # admitted once a's blocks are back
    assert h.finished(a) and h.finished(b)
    assert len(h.output(a)) == 3 and len(h.output(b)) == 3


def test_prefix_cache_hit_through_the_engine():
    h = E.HostEngine(TINY, num_gpu_blocks=64, block_size=4, max_num_seqs=4, max_model_len=128, enable_prefix_cache=True)
    prompt = list(range(1, 18))        # 17 tokens = 4 full blocks + 1
    a = h.add_request(prompt, max_tokens=2)
    t1 = run_to_completion(h, lambda st, i: 300)
    b = h.add_request(prompt + [400, 401], max_tokens=2)
    st = h.schedule()
    # 4 cached blocks: the prefill step only carries tokens 16.. (positions continue at the cached length)
    assert st["is_prefill"] and st["positions"][0] == 16 and st["n_tokens"] == 3
    assert st["block_tables"][0][:4].tolist() == t1[0]["block_tables"][0][:4].tolist()
    assert st["context_lens"].tolist() == [19]
    assert st["slots"][0] == int(st["block_tables"][0][4]) * 4


# ------------------------------------------------------------------------------------------------ plan / tables / permutation
def test_kv_plan_matches_reference_formula():
    """per_block = BS * Hkv * D * 2(K,V) * elem * L (kvcache_allocator.rs:447-468); SURVEY §8 a16: 8 MiB for Llama-3-8B."""
    mc = E.model_config(E.LLAMA3_8B)
    ec = _lib.EngineConfig(block_size=64, kv_fraction=0.5, tp_world_size=1)
    assert L.vra_kv_per_block_bytes(C.byref(mc), C.byref(ec)) == 64 * 8 * 128 * 2 * 2 * 32 == 8388608
    free = 280 * 10 ** 9
    assert L.vra_kv_plan_num_blocks(C.byref(mc), C.byref(ec), free) == int(free * 0.5) // 8388608
    ec8 = _lib.EngineConfig(block_size=64, kv_fraction=0.5, tp_world_size=8)
    mc70 = E.model_config(E.LLAMA3_70B)
    assert L.vra_kv_per_block_bytes(C.byref(mc70), C.byref(ec8)) == 64 * 1 * 128 * 2 * 2 * 80   # Hkv/world = 1
    ecn = _lib.EngineConfig(block_size=64, num_gpu_blocks=1234)
    assert L.vra_kv_plan_num_blocks(C.byref(mc), C.byref(ecn), free) == 1234


@pytest.mark.parametrize("cfg", [E.TINYLLAMA, E.LLAMA3_8B, E.QWEN2_7B,
                                 dict(E.LLAMA3_8B, rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0,
                                                                     high_freq_factor=4.0, original_max_position_embeddings=8192),
                                      max_position_embeddings=131072)])
def test_rope_tables_bit_exact_vs_oracle_and_closed_form(cfg):
    mc = E.model_config(cfg)
    n_pos, half = 300, mc.head_dim // 2
    cos, sin = np.empty((n_pos, half), np.float32), np.empty((n_pos, half), np.float32)
    L.vra_rope_tables_f32(C.byref(mc), n_pos, cos.ctypes.data, sin.ctypes.data)
    ocos, osin = orc.rope_tables(mc.head_dim, mc.rope_theta, n_pos, mc.rope_scaling_type, mc.rope_factor, mc.rope_low_freq_factor,
                                 mc.rope_high_freq_factor, mc.rope_original_max_position)
    assert (cos.view(np.uint32) == ocos.view(np.uint32)).all() and (sin.view(np.uint32) == osin.view(np.uint32)).all()
    # independent restatement of rotary_emb.rs:50-62: inv_freq = 1f32 / (theta.powf(i/d) as f32), pow in f64, angle in f32
    i = np.arange(0, mc.head_dim, 2, dtype=np.float64)
    inv = (np.float32(1.0) / np.power(np.float64(mc.rope_theta), i / mc.head_dim).astype(np.float32)).astype(np.float32)
    if mc.rope_scaling_type == 2:  # llama3 smoothing in f32 (rotary_emb.rs:236-251)
        f, lo, hi, om = np.float32(mc.rope_factor), np.float32(mc.rope_low_freq_factor), np.float32(mc.rope_high_freq_factor), np.float32(mc.rope_original_max_position)
        wl = (np.float32(2 * np.pi) / inv).astype(np.float32)
        smooth = ((om / wl - lo) / (hi - lo)).astype(np.float32)
        mid = ((np.float32(1) - smooth) * inv / f + smooth * inv).astype(np.float32)
        inv = np.where(wl > om / lo, inv / f, np.where(wl < om / hi, inv, mid)).astype(np.float32)
    ang = (np.arange(n_pos, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32)
    assert np.abs(cos - np.cos(ang.astype(np.float64))).max() < 2e-6
    assert np.abs(sin - np.sin(ang.astype(np.float64))).max() < 2e-6


@pytest.mark.parametrize("grouped", [True, False])
def test_marlin_permute_scales(grouped):
    """wna16.rs:180-218: grouped rows are reshaped [-1,64] and permuted by i+8j; channel-wise [-1,32] by 2i+{0,1,8,9,16,17,24,25}."""
    rows, n = (4, 128) if grouped else (1, 128)
    s = np.arange(rows * n, dtype=np.uint16).reshape(rows, n)
    out = np.empty_like(s)
    L.vra_marlin_permute_scales_u16(s.ctypes.data, out.ctypes.data, rows, n, int(grouped))
    perm = [i + 8 * j for i in range(8) for j in range(8)] if grouped else [2 * i + j for i in range(4) for j in (0, 1, 8, 9, 16, 17, 24, 25)]
    want = s.reshape(-1, len(perm))[:, perm].reshape(rows, n)
    assert (out == want).all()
    assert (orc.marlin_permute_scales(s, grouped) == want).all()


# ------------------------------------------------------------------------------------------------ randomized stress
@pytest.mark.parametrize("seed,prefix_cache,swap", [(s, bool(s & 1), False) for s in range(1, 13)] + [(s, bool(s & 1), True) for s in range(13, 21)])
def test_engine_random_streams_keep_the_kv_invariants(seed, prefix_cache, swap):
    """random request streams (shared prefixes, long prompts that need chunking, block pressure, early EOS) on the
    host-only engine: every step's metadata must be self-consistent — the properties the kernels rely on
    (runner.rs:978-1388, block_manager.rs:113-442) — and everything must drain."""
    r = np.random.default_rng(seed)
    BS, NB, CHUNK = 16, 96, 64
    cfg = dict(E.TINYLLAMA)
    skw = dict(cpu_mem_fold=0.5, swap_cooling_ms=-1, min_tokens_left_for_swap=-1) if swap else {}
    if swap:
        NB = 56   # tighter cache: decode-time preemption (and with it swap-out / swap-in) becomes common
    h = E.HostEngine(cfg, num_gpu_blocks=NB, block_size=BS, max_num_seqs=8, max_model_len=512, prefill_chunk=CHUNK, enable_prefix_cache=prefix_cache, **skw)
    shared = [r.integers(5, 1000, size=int(n)).tolist() for n in (40, 70, 17)]
    pending, live, outputs, step_no = 40, {}, {}, 0
    def submit():
        base = shared[int(r.integers(0, len(shared)))] if r.random() < 0.6 else []
        tail = r.integers(5, 1000, size=int(r.integers(1, 150))).tolist()
        prompt = (base + tail)[:300]
        rid = h.add_request(prompt, max_tokens=int(r.integers(1, 24)), eos=(3,))
        live[rid] = dict(prompt=prompt, seen=0)
    while pending > 0 or h.has_unfinished():
        while pending > 0 and r.random() < 0.5:
            submit()
            pending -= 1
        st = h.schedule()
        if st is None:
            if pending == 0 and not h.has_unfinished():
                break
            if pending > 0:
                submit()
                pending -= 1
            continue
        step_no += 1
        assert step_no < 20000, "the engine does not drain"
        T, B = st["n_tokens"], st["n_seqs"]
        assert 1 <= B <= 8 and T >= B
        slots, bt, ctx = st["slots"], st["block_tables"], st["context_lens"]
        # every written slot is written once per step and lies in a block of the writing sequence's table
        live_slots = slots[slots >= 0]
        assert len(set(live_slots.tolist())) == len(live_slots), "two tokens of one step share a KV slot"
        assert (live_slots < NB * BS).all()
        if st["is_prefill"]:
            cu = st["cu_q"]
            assert cu[0] == 0 and cu[-1] == T and (np.diff(cu) >= 1).all() and T <= CHUNK * B
            for b in range(B):
                q0, q1 = int(cu[b]), int(cu[b + 1])
                pos = st["positions"][q0:q1]
                assert (np.diff(pos) == 1).all() and pos[-1] == ctx[b] - 1          # a contiguous span ending at the context
                for j in range(q0, q1):                                            # slot = table[pos / BS] * BS + pos % BS
                    p = int(st["positions"][j])
                    assert slots[j] == int(bt[b, p // BS]) * BS + p % BS
        else:
            assert T == B
            for b in range(B):
                p = int(st["positions"][b])
                assert p == ctx[b] - 1 and slots[b] == int(bt[b, p // BS]) * BS + p % BS
        # no block is shared by two sequences of the step except through the prefix cache (whole cached blocks only)
        used = {}
        for b in range(B):
            nblk = (int(ctx[b]) + BS - 1) // BS
            for i, blk in enumerate(bt[b, :nblk].tolist()):
                assert 0 <= blk < NB
                if blk in used and used[blk] != b:
                    assert prefix_cache and (i + 1) * BS <= int(ctx[b]), "a partially filled block is shared"
                used.setdefault(blk, b)
        h.commit([3 if r.random() < 0.05 else int(v) for v in r.integers(5, 1000, size=B)])
    for rid, info in live.items():
        assert h.finished(rid)
        outputs[rid] = h.output(rid)
        assert len(outputs[rid]) <= 24
    if swap:
        cpu, free_cpu, out_blocks, in_blocks = h.swap_stats()
        assert free_cpu == cpu and in_blocks <= out_blocks     # swap space fully returned (a dropped sequence frees its copy)
    # drained: a fresh maximal request is admitted again (no leaked blocks)
    rid = h.add_request(r.integers(5, 1000, size=300).tolist(), max_tokens=2)
    n = 0
    while h.has_unfinished():
        st = h.schedule()
        assert st is not None
        h.commit([7] * st["n_seqs"])
        n += 1
        assert n < 100
    assert h.finished(rid)
    h.close()


# ---------------------------------------------------------------------------------------------
# BASELINE config 5: 32k-token prompts — chunked prefill (8192), paged-KV block manager, prefix-cache stress
# (SURVEY §8d "Config 5"; scheduler.rs:203,718-785; block_manager.rs:291-299; runner.rs:978-1241)
# ---------------------------------------------------------------------------------------------
LONG = dict(E.LLAMA31_8B)


def test_config5_32k_prompt_takes_four_chunks_then_hits_511_blocks():
    h = E.HostEngine(LONG, num_gpu_blocks=1400, block_size=64, max_num_seqs=8, max_model_len=40960, enable_prefix_cache=True)
    r = np.random.default_rng(42)
    prompt = r.integers(1000, 127000, size=32768).tolist()
    a = h.add_request(prompt, max_tokens=3, ignore_eos=True)
    trace = run_to_completion(h, lambda st, i: 555)
    pre = [t for t in trace if t["is_prefill"]]
    assert [t["n_tokens"] for t in pre] == [8192] * 4                 # exactly four prefill steps of 8192 (scheduler.rs:203)
    for k, t in enumerate(pre):
        assert t["context_lens"].tolist() == [8192 * (k + 1)] and t["positions"][0] == 8192 * k
        bt = t["block_tables"][0]
        assert len(set(bt[:512].tolist())) == 512                   # 512 distinct blocks, table kept across chunks (A13)
        assert t["block_tables"][0][:512].tolist() == pre[0]["block_tables"][0][:512].tolist()
        assert t["slots"].tolist() == [int(bt[p // 64]) * 64 + p % 64 for p in range(8192 * k, 8192 * (k + 1))]
    first_blocks = pre[0]["block_tables"][0][:512].tolist()
    assert h.finished(a) and h.output(a) == [555] * 3
    # ---- the same prompt again: all 512 blocks are cached, the LAST full block is recomputed (block_manager.rs:291-299)
    b = h.add_request(prompt, max_tokens=3, ignore_eos=True)
    trace2 = run_to_completion(h, lambda st, i: 556)
    pre2 = [t for t in trace2 if t["is_prefill"]]
    assert [t["n_tokens"] for t in pre2] == [64]
    assert pre2[0]["positions"].tolist() == list(range(511 * 64, 512 * 64)) and pre2[0]["context_lens"].tolist() == [32768]
    assert pre2[0]["block_tables"][0][:511].tolist() == first_blocks[:511]     # the 511 cached blocks are SHARED, not copied
    assert int(pre2[0]["block_tables"][0][511]) != first_blocks[511]          # the recomputed block is a fresh one
    assert h.finished(b)


def test_config5_eight_prompts_share_a_16k_prefix():
    h = E.HostEngine(LONG, num_gpu_blocks=4096, block_size=64, max_num_seqs=8, max_model_len=40960, enable_prefix_cache=True)
    r = np.random.default_rng(7)
    prefix = r.integers(1000, 127000, size=16384).tolist()
    warm = h.add_request(prefix + r.integers(1000, 127000, size=100).tolist(), max_tokens=2, ignore_eos=True)
    run_to_completion(h, lambda st, i: 9)                            # leaves the 256 prefix blocks in the cache
    assert h.finished(warm)
    tails = [r.integers(1000, 127000, size=1024).tolist() for _ in range(8)]
    rids = [h.add_request(prefix + t, max_tokens=4, ignore_eos=True) for t in tails]
    trace = run_to_completion(h, lambda st, i: 100 + i)
    pre = [t for t in trace if t["is_prefill"]]
    # every prompt starts behind the 256 cached blocks: only its 1024 tail tokens are prefilled — all eight in ONE step
    # (the per-step token cap counts tokens after the prefix match, core.cpp Scheduler::schedule)
    assert [t["n_tokens"] for t in pre] == [8 * 1024]
    shared = None
    for t in pre:
        for i in range(t["n_seqs"]):
            assert t["cu_q"][i + 1] - t["cu_q"][i] == 1024 and t["context_lens"][i] == 16384 + 1024
            head = t["block_tables"][i][:256].tolist()
            shared = shared or head
            assert head == shared                                      # all eight read the SAME physical prefix blocks
            assert t["positions"][t["cu_q"][i]] == 16384
    assert all(h.finished(x) for x in rids)


def test_more_pending_requests_than_max_num_seqs_floor_of_five():
    """the scheduler batches up to max(max_num_seqs, 5) sequences (scheduler.rs:44): every per-sequence staging region must
    hold that many — max_num_seqs = 2 with 7 pending requests schedules 5 and corrupts nothing"""
    h = E.HostEngine(TINY, num_gpu_blocks=64, block_size=4, max_num_seqs=2, max_model_len=64)
    rids = [h.add_request([10 * i + 1, 10 * i + 2, 10 * i + 3], max_tokens=3) for i in range(7)]
    st = h.schedule()
    assert st["is_prefill"] and st["n_seqs"] == 5
    assert st["context_lens"].tolist() == [3] * 5 and st["cu_q"].tolist() == [0, 3, 6, 9, 12, 15]
    assert st["block_tables"].shape[0] == 5
    h.commit([7] * 5)
    trace = run_to_completion(h, lambda st, i: 7)
    assert all(h.finished(r) for r in rids)
    for t in trace:
        assert t["n_seqs"] <= 5
        if not t["is_prefill"]:
            assert (t["context_lens"] >= 4).all()                    # no staging region overwrote another


def test_prefill_chunk_above_the_staging_cap_is_clamped():
    """prefill_chunk > 16384 (the staging / activation cap) is clamped for scheduler AND runner alike: a 40000-token prompt
    goes through in steps of 16384 instead of overrunning the staging buffers"""
    h = E.HostEngine(LONG, num_gpu_blocks=1400, block_size=64, max_num_seqs=4, max_model_len=65536, prefill_chunk=32768)
    p = (np.arange(40000) % 5000 + 1).tolist()
    a = h.add_request(p, max_tokens=2, ignore_eos=True)
    trace = run_to_completion(h, lambda st, i: 3)
    assert [t["n_tokens"] for t in trace if t["is_prefill"]] == [16384, 16384, 40000 - 2 * 16384]
    assert h.finished(a)


# ------------------------------------------------------------------------------------------------ CPU swap
def _swap_engine(**kw):
    args = dict(num_gpu_blocks=12, block_size=4, max_num_seqs=4, max_model_len=64, cpu_mem_fold=1.0, swap_cooling_ms=-1,
                min_tokens_left_for_swap=-1)
    args.update(kw)
    return E.HostEngine(TINY, **args)


def test_preempted_sequence_is_swapped_out_and_back_in():
    """scheduler.rs:303-338,826-955 + block_manager.rs:870-1010: when a running sequence cannot get its next slot and there is
    no prefix cache to evict, the OLDEST preempted sequence goes to the CPU swap space (its GPU blocks return to the free
    list at once), the others keep decoding, and it comes back — into freshly allocated blocks — once there is room."""
    h = _swap_engine()
    a = h.add_request(list(range(1, 13)), max_tokens=20, ignore_eos=True)
    b = h.add_request(list(range(21, 33)), max_tokens=20, ignore_eos=True)
    assert h.swap_stats()[:2] == (12, 12)
    trace = run_to_completion(h, lambda st, i: 7)
    assert h.finished(a) and h.finished(b)
    assert len(h.output(a)) == 20 and len(h.output(b)) == 20          # nobody was dropped
    cpu, free_cpu, out_blocks, in_blocks = h.swap_stats()
    assert out_blocks > 0 and in_blocks == out_blocks and free_cpu == cpu
    # the swapped-out request is the older one (min id among the preempted, scheduler.rs:326-334)
    # (one step earlier request a may decode alone: it took the last free block of that step, scheduler.rs:353-358)
    solo = [t["requests"] for t in trace if not t["is_prefill"] and t["n_seqs"] == 1]
    assert solo.count([b]) >= 4 and solo[-1] == [a]                   # b runs on alone; a finishes last, after its swap-in
    # after the swap-in request a decodes again, its context continuing where it stopped, in a table of new blocks
    a_steps = [t for t in trace if not t["is_prefill"] and a in t["requests"]]
    ctx = [int(t["context_lens"][t["requests"].index(a)]) for t in a_steps]
    assert ctx == list(range(13, 13 + len(ctx)))                      # no token lost or repeated across the swap
    for t in a_steps:                                                  # slots always derive from the table of that step
        i = t["requests"].index(a)
        pos = int(t["positions"][i])
        assert int(t["slots"][i]) == int(t["block_tables"][i][pos // 4]) * 4 + pos % 4


def test_single_running_sequence_is_never_swapped_out():
    """scheduler.rs:315-324: with one running sequence swapping makes no sense — it waits (here: is dropped by the engine's
    no-progress rule, engine.rs:1103-1120)."""
    h = _swap_engine(num_gpu_blocks=6, max_model_len=64)
    a = h.add_request(list(range(1, 13)), max_tokens=40, ignore_eos=True)
    run_to_completion(h, lambda st, i: 7)
    assert h.finished(a) and h.swap_stats()[2] == 0


def test_swap_in_waits_for_the_cooling_period():
    """scheduler.rs:49,846: a sequence swapped out less than SWAP_COOLING_PERIOD ago stays where it is."""
    import time
    h = _swap_engine(swap_cooling_ms=150)
    a = h.add_request(list(range(1, 13)), max_tokens=20, ignore_eos=True)
    b = h.add_request(list(range(21, 33)), max_tokens=20, ignore_eos=True)
    t_out = None
    t_back = None
    for _ in range(100000):
        st = h.schedule()
        out_blocks, in_blocks = h.swap_stats()[2:]
        if out_blocks and t_out is None:
            t_out = time.monotonic()
        if in_blocks and t_back is None:
            t_back = time.monotonic()
        if st is None:
            if not h.has_unfinished():
                break
            continue
        h.commit([7] * st["n_seqs"])
    assert h.finished(a) and h.finished(b) and len(h.output(a)) == 20
    assert t_out is not None and t_back is not None and t_back - t_out >= 0.14


def test_shared_prefix_blocks_are_not_swapped():
    """block_manager.rs:876-893: a sequence holding a block with ref_count > 1 (prefix cache) cannot be swapped out."""
    h = _swap_engine(num_gpu_blocks=14, enable_prefix_cache=True)
    warm = h.add_request(list(range(1, 10)), max_tokens=1)
    run_to_completion(h, lambda st, i: 7)
    a = h.add_request(list(range(1, 10)) + [90, 91, 92], max_tokens=24, ignore_eos=True)   # shares 2 cached blocks
    b = h.add_request(list(range(41, 53)), max_tokens=24, ignore_eos=True)
    run_to_completion(h, lambda st, i: 7)
    assert h.finished(a) and h.finished(b)
    # pressure is relieved by evicting the prefix cache first (scheduler.rs:318-321); whatever was swapped came back
    cpu, free_cpu, out_blocks, in_blocks = h.swap_stats()
    assert in_blocks == out_blocks and free_cpu == cpu
