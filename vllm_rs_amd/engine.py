"""Python mirror of the reference's `EngineBuilder`/`LLMEngine` surface for this path
(src/api.rs:25-114, src/core/engine.rs:108,1291,1457): token ids in, token ids out.  All work is
done by the native runtime in libvllm_rs_amd.so (host/*.cpp + csrc/*.hip); no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EngineConfig, ModelConfig

BF16, F16, F32 = 0, 1, 2
QUANT = {None: 0, "": 0, "none": 0, "gptq": 1, "awq": 2}
ROPE = {None: 0, "": 0, "default": 0, "linear": 1, "llama3": 2, "dynamic": 3, "yarn": 4}


def model_config(cfg):
    """dict with HF config.json-style keys -> ModelConfig (config.rs:218-255)."""
    rs = cfg.get("rope_scaling") or {}
    head_dim = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_heads"]
    return ModelConfig(
        arch={"llama": 0, 0: 0, "qwen2": 1, 1: 1, "qwen3": 2, 2: 2}[cfg.get("arch", "llama")],
        qk_norm={None: 0, 0: 0, False: 0, "head": 1, 1: 1, True: 1, "full": 2, 2: 2}[cfg.get("qk_norm", 1 if cfg.get("arch") in ("qwen3", 2) else 0)],
        hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"], num_layers=cfg["num_layers"],
        num_heads=cfg["num_heads"], num_kv_heads=cfg["num_kv_heads"], head_dim=head_dim, vocab_size=cfg["vocab_size"],
        max_position_embeddings=cfg["max_position_embeddings"], rms_norm_eps=cfg["rms_norm_eps"],
        rope_theta=cfg["rope_theta"], rope_scaling_type=ROPE[rs.get("rope_type", rs.get("type", ""))],
        rope_factor=rs.get("alpha", rs.get("factor", 1.0)), rope_low_freq_factor=rs.get("low_freq_factor", 1.0),
        rope_high_freq_factor=rs.get("high_freq_factor", 4.0),
        # rotary_emb.rs:150-164: the key, else max_position_embeddings / factor, else max_position_embeddings
        rope_original_max_position=int(rs.get("original_max_position_embeddings") or
                                       (cfg["max_position_embeddings"] / rs["factor"] if rs.get("factor") else cfg["max_position_embeddings"])),
        rope_dynamic_alpha=int("alpha" in rs), rope_yarn_beta_fast=rs.get("beta_fast", 32.0), rope_yarn_beta_slow=rs.get("beta_slow", 1.0),
        rope_yarn_attn_factor=rs.get("attn_factor", 1.0), rope_yarn_extrapolation_factor=rs.get("extrapolation_factor", 1.0),
        # (the f64 the reference keeps, rotary_emb.rs:150-164; and which yarn fields were given explicitly: an explicit 0 is honoured)
        rope_original_max_position_f=float(rs.get("original_max_position_embeddings") or
                                           (cfg["max_position_embeddings"] / rs["factor"] if rs.get("factor") else cfg["max_position_embeddings"])),
        rope_yarn_explicit=sum(1 << i for i, k in enumerate(("beta_fast", "beta_slow", "attn_factor", "extrapolation_factor")) if k in rs),
        attention_bias=int(bool(cfg.get("attention_bias"))), quant_method=QUANT[cfg.get("quant_method")],
        bits=4, group_size=cfg.get("group_size", 128), dtype=cfg.get("dtype", BF16),
        tie_word_embeddings=int(bool(cfg.get("tie_word_embeddings"))),
        sliding_window=int(cfg.get("sliding_window") or 0))


LLAMA3_8B = dict(arch="llama", hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8,
                 head_dim=128, vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0,
                 quant_method="gptq", group_size=128, dtype=BF16)
QWEN2_7B = dict(arch="qwen2", hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4,
                head_dim=128, vocab_size=152064, max_position_embeddings=32768, rms_norm_eps=1e-6, rope_theta=1000000.0,
                attention_bias=True, quant_method="awq", group_size=128, dtype=BF16)
# Llama-3.1-8B: the same widths with the llama3 rope scaling and a 128k window — the shape of BASELINE config 5
# (32k-token prompts need max_position_embeddings >= 32768; rotary_emb.rs:208-278)
LLAMA31_8B = dict(LLAMA3_8B, max_position_embeddings=131072,
                  rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                    original_max_position_embeddings=8192))
LLAMA3_70B = dict(arch="llama", hidden_size=8192, intermediate_size=28672, num_layers=80, num_heads=64, num_kv_heads=8,
                  head_dim=128, vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0,
                  quant_method="gptq", group_size=128, dtype=BF16)
# what ONE rank of Llama-3-70B TP=8 computes (column-parallel q/k/v/gate/up, row-parallel o/down, replicated lm_head:
# SURVEY §8e), as a stand-alone single-GPU shape: the per-rank compute of BASELINE config 4 without the all-reduces
LLAMA3_70B_TP8_RANK = dict(arch="llama", hidden_size=8192, intermediate_size=28672 // 8, num_layers=80, num_heads=64 // 8,
                           num_kv_heads=1, head_dim=128, vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5,
                           rope_theta=500000.0, quant_method="gptq", group_size=128, dtype=BF16)
TINYLLAMA = dict(arch="llama", hidden_size=2048, intermediate_size=5632, num_layers=22, num_heads=32, num_kv_heads=4,
                 head_dim=64, vocab_size=32000, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0,
                 quant_method=None, dtype=BF16)


class Engine:
    def __init__(self, cfg, *, block_size=64, max_num_seqs=32, max_model_len=0, num_gpu_blocks=0, kv_fraction=0.0,
                 prefill_chunk=8192, enable_prefix_cache=False, use_graph=True, tp_rank=0, tp_world_size=1, device=0,
                 seed=1234, comm=None, fp8_kvcache=False, cpu_mem_fold=0.0, swap_cooling_ms=0, min_tokens_left_for_swap=0):
        # cpu_mem_fold: CPU swap space as a fraction of the KV blocks, PINNED per layer at finalize.  0 (no swap space: preempted
        # sequences are recomputed) unless the caller asks — the reference's default 0.2 (kvcache_allocator.rs:317) is applied where
        # the reference's ENGINE plans CPU block ids, i.e. on the runner-IPC path (runner_ipc.py, host/runner_main.cpp); as the
        # package-wide default it pinned 13 GB of host memory per 8192-block engine
        self.L = _lib.load()
        if self.L.vra_device_count() <= 0:
            raise RuntimeError("vllm_rs_amd.Engine needs a GPU: no HIP device visible (there is no CPU fallback)")
        self.cfg = cfg
        self.mc = model_config(cfg)
        self.ec = EngineConfig(block_size=block_size, max_num_seqs=max_num_seqs, max_model_len=max_model_len,
                               num_gpu_blocks=num_gpu_blocks, kv_fraction=kv_fraction, prefill_chunk=prefill_chunk,
                               enable_prefix_cache=int(enable_prefix_cache), prefix_cache_fraction=0.65,
                               use_graph=int(use_graph), tp_rank=tp_rank, tp_world_size=tp_world_size, device=device,
                               seed=seed, fp8_kvcache=int(fp8_kvcache), cpu_mem_fold=cpu_mem_fold,
                               swap_cooling_ms=swap_cooling_ms, min_tokens_left_for_swap=min_tokens_left_for_swap)
        self.h = self.L.vra_engine_create(C.byref(self.mc), C.byref(self.ec))
        if not self.h:
            raise RuntimeError("vra_engine_create failed")
        if comm is not None:
            self.L.vra_engine_set_comm(self.h, comm)

    def _check(self, rc, what):
        if rc is None or rc < 0:
            raise RuntimeError(f"{what}: {self.L.vra_engine_last_error(self.h).decode()}")
        return rc

    def init_synthetic(self, finalize=True):
        self._check(self.L.vra_engine_init_synthetic(self.h), "init_synthetic")
        return self.finalize() if finalize else self

    def load_weights(self, tensors, finalize=True):
        """tensors: name -> numpy array in checkpoint format (16-bit floats as uint16 bit patterns in model dtype).
        Under tensor parallelism every rank is handed the FULL tensors and keeps its shard (wna16.rs:35-40,
        distributed.rs:498-538)."""
        for name, a in tensors.items():
            a = np.ascontiguousarray(a)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.L.vra_engine_load_tensor(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, a.itemsize),
                        f"load_tensor({name})")
        return self.finalize() if finalize else self

    def swap_stats(self):
        """(cpu blocks, free cpu blocks, blocks swapped out so far, blocks swapped in so far)"""
        out = (C.c_int64 * 4)()
        self.L.vra_engine_swap_stats(self.h, out)
        return tuple(int(x) for x in out)

    def plan_kv_blocks(self):
        """this rank's KV plan before the cache exists; TP launchers take the minimum over ranks (vllm_rs_amd/runner.py)"""
        return self._check(self.L.vra_engine_plan_kv_blocks(self.h), "plan_kv_blocks")

    def set_num_gpu_blocks(self, n):
        self._check(self.L.vra_engine_set_num_gpu_blocks(self.h, int(n)), "set_num_gpu_blocks")
        return self

    def finalize(self):
        self._check(self.L.vra_engine_finalize_weights(self.h), "finalize")
        return self

    def last_logits(self, n_seqs=1):
        """f32 logits [n_seqs, vocab] of the step that just ran (graph replay or eager): parity instrumentation"""
        import numpy as np
        out = np.empty((n_seqs, self.cfg["vocab_size"]), np.float32)
        self._check(self.L.vra_engine_copy_logits(self.h, out.ctypes.data_as(C.c_void_p), n_seqs), "copy_logits")
        return out

    def finalize_model(self):
        """weights only (repack + decode layouts): the KV cache is sized later, from the engine process's plan"""
        self._check(self.L.vra_engine_finalize_model(self.h), "finalize_model")
        return self

    def update_config(self, num_gpu_blocks=0, max_num_seqs=0, max_model_len=0, cpu_mem_fold=0.0, kv_fraction=0.0):
        """the negotiated EngineConfig of MessageType::UsableMemoryLeft, before `finalize` allocates anything"""
        ec = self.ec
        ec.num_gpu_blocks, ec.max_num_seqs, ec.max_model_len = int(num_gpu_blocks), int(max_num_seqs), int(max_model_len)
        ec.cpu_mem_fold, ec.kv_fraction = float(cpu_mem_fold), float(kv_fraction)
        self._check(self.L.vra_engine_update_config(self.h, C.byref(ec)), "update_config")
        return self

    @classmethod
    def from_pretrained(cls, model_dir, dtype=None, finalize=True, **kw):
        """HF checkpoint directory (config.json + *.safetensors, GPTQ / AWQ int4 or dense) -> engine
        (`EngineBuilder::build` + `WNA16::new`, src/api.rs:25-114, wna16.rs:56-152)."""
        from . import checkpoint
        cfg, tensors = checkpoint.load_pretrained(model_dir, dtype)
        eng = cls(cfg, **kw)
        for name, a in tensors:
            shape = (C.c_int64 * a.ndim)(*a.shape)
            eng._check(eng.L.vra_engine_load_tensor(eng.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, a.itemsize),
                       f"load_tensor({name})")
        if finalize:
            eng._check(eng.L.vra_engine_finalize_weights(eng.h), "finalize")
        return eng

    @property
    def num_gpu_blocks(self):
        return self.L.vra_engine_num_gpu_blocks(self.h)

    def add_request(self, prompt, max_tokens=16, ignore_eos=False, eos=(), sampling=None):
        """sampling: None = greedy; dict(temperature=, top_k=, top_p=, frequency_penalty=, presence_penalty=) with missing
        keys unset (an EMPTY dict is the reference's default: top-k 32, top-p 0.95, temperature 0.7 — Appendix A4)"""
        p = np.ascontiguousarray(prompt, np.uint32)
        e = np.ascontiguousarray(list(eos), np.uint32)
        ep = e.ctypes.data_as(C.c_void_p) if len(e) else None
        if sampling is None:
            rid = self.L.vra_engine_add_request(self.h, p.ctypes.data_as(C.c_void_p), len(p), max_tokens, int(ignore_eos), ep, len(e))
        else:
            sp = _lib.SamplingParams(temperature=sampling.get("temperature", -1.0), top_k=sampling.get("top_k", 0) or 0,
                                     top_p=sampling.get("top_p", -1.0), has_frequency_penalty=int("frequency_penalty" in sampling),
                                     has_presence_penalty=int("presence_penalty" in sampling),
                                     frequency_penalty=sampling.get("frequency_penalty", 0.0),
                                     presence_penalty=sampling.get("presence_penalty", 0.0))
            rid = self.L.vra_engine_add_request_ex(self.h, p.ctypes.data_as(C.c_void_p), len(p), max_tokens, int(ignore_eos), ep, len(e),
                                                   C.byref(sp))
        return self._check(rid, "add_request")

    def step(self):
        pf = C.c_int32(0)
        n = self.L.vra_engine_step(self.h, C.byref(pf))
        self._check(n, "step")
        return n, bool(pf.value)

    def has_unfinished(self):
        return bool(self.L.vra_engine_has_unfinished(self.h))

    def finished(self, rid):
        return bool(self.L.vra_engine_request_finished(self.h, rid))

    def output(self, rid, cap=1 << 16):
        buf = np.empty(cap, np.uint32)
        n = self.L.vra_engine_request_output(self.h, rid, buf.ctypes.data_as(C.c_void_p), cap)
        return buf[:max(n, 0)].copy()

    def times(self, rid):
        t = (C.c_double * 3)()
        self.L.vra_engine_request_times(self.h, rid, t)
        return dict(created_ms=t[0], first_token_ms=t[1], finished_ms=t[2])

    def generate(self, prompts, max_tokens=16, ignore_eos=False, eos=(), sampling=None):
        """LLMEngine::generate_sync (engine.rs:1291): run to completion, return outputs in request order."""
        rids = [self.add_request(p, max_tokens, ignore_eos, eos, sampling) for p in prompts]
        while self.has_unfinished():
            self.step()
        return [self.output(r) for r in rids]

    def forward_raw(self, ids, positions, slot_mapping, block_tables, context_lens, cu_q=None):
        """`forward(input_ids, positions, kv_caches, input_metadata)` (llama.rs:323-339) → f32 logits [n_seqs, V]."""
        ids = np.ascontiguousarray(ids, np.uint32)
        positions = np.ascontiguousarray(positions, np.int64)
        slot_mapping = np.ascontiguousarray(slot_mapping, np.int64)
        block_tables = np.ascontiguousarray(block_tables, np.uint32)
        context_lens = np.ascontiguousarray(context_lens, np.uint32)
        B, mb = block_tables.shape
        cu = None if cu_q is None else np.ascontiguousarray(cu_q, np.uint32)
        out = np.empty((B, self.mc.vocab_size), np.float32)
        rc = self.L.vra_engine_forward_raw(self.h, ids.ctypes.data_as(C.c_void_p), positions.ctypes.data_as(C.c_void_p),
                                           slot_mapping.ctypes.data_as(C.c_void_p), len(ids), int(cu is not None),
                                           block_tables.ctypes.data_as(C.c_void_p), mb, context_lens.ctypes.data_as(C.c_void_p),
                                           cu.ctypes.data_as(C.c_void_p) if cu is not None else None, B,
                                           out.ctypes.data_as(C.c_void_p))
        self._check(rc, "forward_raw")
        return out

    TP_STAGES = ("q", "k", "v", "attn", "o_partial", "h_after_o", "act", "down_partial", "h_after_down")

    def norm_deferred(self, rows, layer=1):
        """bit 0 / bit 1: the norm + q/k/v / norm + gate/up launch of a step of `rows` rows of layer `layer` applies rstd in its epilogue"""
        return int(self.L.vra_engine_norm_deferred(self.h, int(rows), int(layer)))

    def tp_snapshots(self, on=True, layer=0):
        """parity instrumentation: keep copies of one layer's stages of every forward (vra_engine_debug_tp_snapshots)"""
        self.L.vra_engine_debug_tp_snapshots(self.h, 1 + int(layer) if on else 0)
        return self

    def read_tp_snapshots(self):
        """stage name -> uint16 bit patterns (flat) of the stages layer 0 of the last forward left behind"""
        out = {}
        cap = 64 << 20
        buf = np.empty(cap // 2, np.uint16)
        for i, name in enumerate(self.TP_STAGES):
            n = self.L.vra_engine_debug_read_tp_snapshot(self.h, i, buf.ctypes.data_as(C.c_void_p), cap)
            if n > 0:
                out[name] = buf[:n // 2].copy()
        return out

    def timed_decode(self, steps):
        return self.L.vra_engine_timed_decode(self.h, steps)

    def bench_replay(self, steps):
        """ms per replay of the last step's decode graph, back to back, no host work in between (measurement aid)"""
        return self.L.vra_engine_bench_replay(self.h, steps)

    def bench_gemm(self, which, m, iters):
        return self.L.vra_engine_bench_gemm(self.h, which, m, iters)

    def gemm_bytes(self, which, m):
        return self.L.vra_engine_gemm_bytes(self.h, which, m)

    def close(self):
        if self.h:
            self.L.vra_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostEngine:
    """Host-only engine (vra_engine_config.device = -1): the native scheduler, block manager, prefix cache
    and metadata staging of the real engine with the forward pass replaced by caller-supplied tokens.
    Needs no GPU; used by the CPU parity tests of scheduler.rs / block_manager.rs / runner.rs behaviour."""

    def __init__(self, cfg, *, num_gpu_blocks, block_size=64, max_num_seqs=32, max_model_len=0, prefill_chunk=8192,
                 enable_prefix_cache=False, cpu_mem_fold=0.0, swap_cooling_ms=0, min_tokens_left_for_swap=0):
        self.L = _lib.load()
        self.mc = model_config(cfg)
        self.ec = EngineConfig(block_size=block_size, max_num_seqs=max_num_seqs, max_model_len=max_model_len,
                               num_gpu_blocks=num_gpu_blocks, kv_fraction=0.0, prefill_chunk=prefill_chunk,
                               enable_prefix_cache=int(enable_prefix_cache), prefix_cache_fraction=0.65, use_graph=0,
                               tp_rank=0, tp_world_size=1, device=-1, seed=0, cpu_mem_fold=cpu_mem_fold,
                               swap_cooling_ms=swap_cooling_ms, min_tokens_left_for_swap=min_tokens_left_for_swap)
        self.h = self.L.vra_engine_create(C.byref(self.mc), C.byref(self.ec))
        if not self.h or self.L.vra_engine_finalize_weights(self.h) != 0:
            raise RuntimeError("host engine: " + (self.L.vra_engine_last_error(self.h).decode() if self.h else "create failed"))

    def add_request(self, prompt, max_tokens=16, ignore_eos=False, eos=()):
        p = np.ascontiguousarray(prompt, dtype=np.uint32)
        e = np.ascontiguousarray(list(eos), dtype=np.uint32)
        rid = self.L.vra_engine_add_request(self.h, p.ctypes.data, len(p), max_tokens, int(ignore_eos),
                                            e.ctypes.data if len(e) else None, len(e))
        if rid < 0:
            raise RuntimeError(self.L.vra_engine_last_error(self.h).decode())
        return rid

    def schedule(self):
        """-> None when idle, else dict(is_prefill, ids, positions, slots, block_tables, context_lens, cu_q, requests)."""
        m = _lib.StepMeta()
        pf = C.c_int32(0)
        n = self.L.vra_engine_dry_schedule(self.h, C.byref(pf), C.byref(m))
        if n < 0:
            raise RuntimeError(self.L.vra_engine_last_error(self.h).decode())
        if n == 0:
            return None
        T, B, MB = m.n_tokens, m.n_seqs, m.max_blocks
        arr = lambda ptr, k: np.ctypeslib.as_array(ptr, shape=(k,)).copy()
        return dict(is_prefill=bool(pf.value), n_tokens=T, n_seqs=B, max_blocks=MB, max_seqlen_q=m.max_seqlen_q,
                    max_context_len=m.max_context_len, ids=arr(m.input_ids, T), positions=arr(m.positions, T),
                    slots=arr(m.slot_mapping, T), block_tables=arr(m.block_tables, B * MB).reshape(B, MB),
                    context_lens=arr(m.context_lens, B), cu_q=arr(m.cu_seqlens_q, B + 1) if pf.value else None,
                    requests=[int(m.request_ids[i]) for i in range(min(B, 64))])

    def commit(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.uint32)
        if self.L.vra_engine_dry_commit(self.h, t.ctypes.data, len(t)) < 0:
            raise RuntimeError(self.L.vra_engine_last_error(self.h).decode())

    def has_unfinished(self):
        return bool(self.L.vra_engine_has_unfinished(self.h))

    def swap_stats(self):
        """(cpu blocks, free cpu blocks, blocks swapped out so far, blocks swapped in so far)"""
        out = (C.c_int64 * 4)()
        self.L.vra_engine_swap_stats(self.h, out)
        return tuple(int(x) for x in out)

    def finished(self, rid):
        return bool(self.L.vra_engine_request_finished(self.h, rid))

    def output(self, rid, cap=65536):
        buf = np.zeros(cap, np.uint32)
        n = self.L.vra_engine_request_output(self.h, rid, buf.ctypes.data, cap)
        return buf[:max(n, 0)].tolist()

    def close(self):
        if self.h:
            self.L.vra_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
