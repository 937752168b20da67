"""CPU ORACLE for the whole forward (TEST INFRASTRUCTURE ONLY — see oracle/vra_oracle.c header).

Restates, op by op and rounding by rounding, the reference's
  LLaMaForCausalLM::forward_inner      src/models/llama.rs:269-321
  LLaMaDecoderLayer::forward           src/models/llama.rs:107-131
  Qwen3ForCausalLM (Qwen2 checkpoints) src/models/qwen3.rs:308-371 (qkv bias attention.rs:411-415)
  Attention::forward_ext               src/models/layers/attention.rs:648-839
  MLP::forward                         src/models/layers/mlp.rs:451-469
  WNA16::forward                       src/models/layers/wna16.rs:263-306
on top of the C primitives in vra_oracle.c.  Weights are given in CHECKPOINT format under their
HuggingFace names (qweight/qzeros/scales for gptq|awq, weight for dense).
"""
import numpy as np

from . import oracle as orc

BF16, F16, F32 = 0, 1, 2

# The engine's fused RMSNorm + int4 GEMV launches apply the RMSNorm factor in the GEMV's EPILOGUE where that saves a pass over x:
# the 1..4-row decode kernel always (vllm_rs_amd/csrc/gemv_q4s.cuh), and at 5..32 rows the launches whose producer left them
# ready-made operands (gemv_q4w.cuh, GemvSArgs::pre_*) — rstd commutes with the product; orc.rms_norm_deferred + the row_scale of
# orc.wna16_gemm restate that order.  WHICH launches of a step do so is a shape rule of the engine; GPU parity runs install the
# library here (tests/conftest.py, tests/full_depth.py: `vra_debug_norm_deferred_mask`, checked against the engine's own report by
# tests/test_gpu_engine.py) so that the oracle restates the order the engine runs.  None (CPU runs, no engine to compare with): the
# reference order everywhere (others.rs:11-29).
ENGINE_RULE = None  # the loaded library (ctypes)

# How the int4 product treats the dequantised weight.  "exact" (default): out = rnd(Σ x·s·(q − z)), one rounding at the output — what
# the HIP kernels compute.  "marlin": every weight rounded to the model dtype first, w = rnd((q − z)·s), then the
# product — what the reference's CUDA build computes (Marlin, src/utils/gptq.rs:116-178).  tests/full_depth.py reports the engine
# against BOTH, each with the reference's norm order (ENGINE_RULE None, others.rs:11-29), beside the mirrored order.
WEIGHT_ROUNDING = "exact"


def deferred_norm_mask(cfg, T, tp_world=1, layer=1):
    """bit 0: norm + q/k/v of a T-row step of layer `layer` in the deferred order, bit 1: norm + gate/up (Model::norm_deferred_mask)"""
    if ENGINE_RULE is None or T < 1 or cfg.get("quant_method") not in ("gptq", "awq"):
        return 0
    hq = cfg["num_heads"] // tp_world
    hkv = max(1, cfg["num_kv_heads"] // tp_world)
    return int(ENGINE_RULE.vra_debug_norm_deferred_mask(cfg["hidden_size"], cfg["intermediate_size"] // tp_world, hq, hkv, cfg["head_dim"],
                                                        cfg.get("group_size", 128), 1, int(bool(cfg.get("attention_bias"))), tp_world, T, layer, cfg["dtype"]))


def dense_prefill_rows(cfg, T):
    """True when the engine runs the int4 GEMMs of a T-row step as dequant pass + dense GEMM (vllm_rs_amd/csrc/gemm_dense.cuh): every
    weight rounded to the model dtype first — Marlin's arithmetic (gptq.rs:116-178), WEIGHT_ROUNDING "marlin" for that step.  A shape
    rule of the engine like the deferred norm order: asked of the installed library (`vra_debug_dense_prefill_min_rows`; 0 = never)."""
    if ENGINE_RULE is None or cfg.get("quant_method") not in ("gptq", "awq"):
        return False
    mr = int(ENGINE_RULE.vra_debug_dense_prefill_min_rows())
    return mr > 0 and T >= mr


class Linear:
    def __init__(self, w, prefix, cfg):
        self.dt = cfg["dtype"]
        self.bias = w.get(prefix + ".bias")
        if prefix + ".qweight" in w:
            self.quant = True
            self.gs = cfg["group_size"]
            self.scales = w[prefix + ".scales"]
            G, N = self.scales.shape
            if cfg["quant_method"] == "awq":
                K = w[prefix + ".qweight"].shape[0]
                self.idx = orc.awq_unpack(w[prefix + ".qweight"], K, N)
                self.zeros = orc.awq_unpack_zeros(w[prefix + ".qzeros"], G, N)
            else:  # gptq, symmetric (marlin path, wna16.rs:154-160): zero point 8
                K = w[prefix + ".qweight"].shape[0] * 8
                self.idx = orc.gptq_unpack(w[prefix + ".qweight"], K, N)
                self.zeros = None
        else:
            self.quant = False
            self.w = w[prefix + ".weight"]  # [N, K]

    marlin_step = False  # set per forward by OracleModel.forward (dense_prefill_rows)

    def partial(self, x, k0, k1):
        """row-parallel shard (TensorParallelRowLinear, distributed.rs:438-455): x[:, k0:k1] against rows k0..k1 of the
        weight, no bias, rounded to the model dtype — what ONE rank hands to the all-reduce"""
        x = np.ascontiguousarray(x[:, k0:k1])
        if self.quant:
            g = self.gs if self.gs > 0 else self.idx.shape[0]
            assert k0 % g == 0 and k1 % g == 0
            z = None if self.zeros is None else np.ascontiguousarray(self.zeros[k0 // g:k1 // g])
            return orc.wna16_gemm(x, np.ascontiguousarray(self.idx[k0:k1]), z, np.ascontiguousarray(self.scales[k0 // g:k1 // g]),
                                  self.gs, self.dt, None, None, marlin_rounded=WEIGHT_ROUNDING == "marlin" or Linear.marlin_step)
        return orc.dense_gemm(x, np.ascontiguousarray(self.w[:, k0:k1]), None, self.dt, self.dt)

    def __call__(self, x, residual=None, row_scale=None):
        if self.quant:
            return orc.wna16_gemm(x, self.idx, self.zeros, self.scales, self.gs, self.dt, self.bias, residual, row_scale,
                                  marlin_rounded=WEIGHT_ROUNDING == "marlin" or Linear.marlin_step)
        assert row_scale is None, "the deferred norm order exists for the int4 decode kernel only"
        out = orc.dense_gemm(x, self.w, self.bias, self.dt, self.dt)
        return orc.add(out, residual, self.dt) if residual is not None else out


class OracleModel:
    """tp_world > 1 restates the tensor-parallel arithmetic of the reference in ONE process: column-parallel layers
    produce exact slices of the unsharded result (nothing to simulate); the row-parallel o_proj / down_proj produce
    per-rank partial sums rounded to the model dtype, which the all-reduce adds (here: in rank order, f32, one
    rounding — NCCL's order is unspecified), then bias, then the residual (distributed.rs:438-455, llama.rs:126,130)."""

    def reset_cache(self):
        """empty KV cache: the same weights can replay a request under another arithmetic variant (tests/full_depth.py)"""
        for c in self.kc + self.vc:
            c[...] = 0

    def __init__(self, cfg, weights, num_blocks, block_size=64, tp_world=1, fp8_kvcache=False):
        self.cfg, self.w, self.BS = cfg, weights, block_size
        self.tp = tp_world
        dt = cfg["dtype"]
        self.dt = dt
        # fp8 KV cache (EngineConfig.fp8_kvcache, kvcache_allocator.rs:188-193,776): one E4M3 byte per element, scale 1.0
        self.kv_dt = orc.FP8 if fp8_kvcache else dt
        L, Hkv, D = cfg["num_layers"], cfg["num_kv_heads"], cfg["head_dim"]
        cdt = np.uint8 if fp8_kvcache else np.uint16
        self.kc = [np.zeros((num_blocks, Hkv, block_size, D), cdt) for _ in range(L)]
        self.vc = [np.zeros((num_blocks, Hkv, D, block_size), cdt) for _ in range(L)]
        rs = cfg.get("rope_scaling") or {}
        st = {"": 0, "default": 0, "linear": 1, "llama3": 2, "dynamic": 3, "yarn": 4}[rs.get("rope_type", rs.get("type", ""))]
        # rotary_emb.rs:150-164: the key, else max_position_embeddings / factor, else max_position_embeddings
        omax = rs.get("original_max_position_embeddings") or (cfg["max_position_embeddings"] / rs["factor"] if rs.get("factor") else cfg["max_position_embeddings"])
        if st >= 3:
            cos, sin = orc.rope_tables_ext(D, cfg["rope_theta"], cfg["max_position_embeddings"], st, rs.get("alpha", rs.get("factor", 1.0)), int(omax),
                                           "alpha" in rs, rs.get("beta_fast", 32.0), rs.get("beta_slow", 1.0), rs.get("attn_factor", 1.0),
                                           rs.get("extrapolation_factor", 1.0))
        else:
            cos, sin = orc.rope_tables(D, cfg["rope_theta"], cfg["max_position_embeddings"], st, rs.get("factor", 1.0),
                                       rs.get("low_freq_factor", 1.0), rs.get("high_freq_factor", 4.0), int(omax))
        self.cos, self.sin = orc.to_dt(cos, dt), orc.to_dt(sin, dt)  # tables cast to the model dtype (llama.rs:179-189)
        self.layers = []
        for i in range(L):
            p = f"model.layers.{i}."
            self.layers.append(dict(
                attn_norm=weights[p + "input_layernorm.weight"], ffn_norm=weights[p + "post_attention_layernorm.weight"],
                q=Linear(weights, p + "self_attn.q_proj", cfg), k=Linear(weights, p + "self_attn.k_proj", cfg),
                v=Linear(weights, p + "self_attn.v_proj", cfg), o=Linear(weights, p + "self_attn.o_proj", cfg),
                gate=Linear(weights, p + "mlp.gate_proj", cfg), up=Linear(weights, p + "mlp.up_proj", cfg),
                down=Linear(weights, p + "mlp.down_proj", cfg),
                q_norm=weights.get(p + "self_attn.q_norm.weight"), k_norm=weights.get(p + "self_attn.k_norm.weight")))
        self.embed = weights["model.embed_tokens.weight"]
        self.final_norm = weights["model.norm.weight"]
        self.lm_head = weights.get("lm_head.weight", self.embed)

    def _row_parallel(self, lin, x, residual):
        if self.tp == 1:
            return lin(x, residual=residual)
        K = x.shape[1]
        acc = np.zeros((x.shape[0], residual.shape[1]), np.float32)
        for r in range(self.tp):
            acc += orc.from_dt(lin.partial(x, r * K // self.tp, (r + 1) * K // self.tp), self.dt)
        out = orc.to_dt(acc, self.dt)
        if lin.bias is not None:
            out = orc.add(out, np.broadcast_to(lin.bias, out.shape).copy(), self.dt)
        return orc.add(out, residual, self.dt)

    def qk_norm(self, L, q, k):
        """q_norm / k_norm of Attention::forward_ext (attention.rs:713-735), before the rotary embedding: per head over head_dim
        (weight [head_dim], attention.rs:724-731) or over the whole q / k row (weight [heads * head_dim], `full_dim_qk_norm`,
        attention.rs:714-722).  One RMSNorm each, one rounding (NormX::forward: f32 inside, cast back)."""
        wq, wk = L.get("q_norm"), L.get("k_norm")
        if wq is None or wk is None:
            return q, k
        T, Hq, D = q.shape
        Hkv = k.shape[1]
        eps, dt = self.cfg["rms_norm_eps"], self.dt
        if wq.shape[0] == D:
            q = orc.rms_norm(q.reshape(T * Hq, D), wq, eps, dt).reshape(T, Hq, D)
            k = orc.rms_norm(k.reshape(T * Hkv, D), wk, eps, dt).reshape(T, Hkv, D)
        else:
            q = orc.rms_norm(q.reshape(T, Hq * D), wq, eps, dt).reshape(T, Hq, D)
            k = orc.rms_norm(k.reshape(T, Hkv * D), wk, eps, dt).reshape(T, Hkv, D)
        return q, k

    def forward(self, ids, positions, slot_mapping, block_tables, context_lens, cu_q=None):
        """returns f32 logits [n_seqs, vocab]; cu_q None => decode (one token per sequence)."""
        cfg, dt = self.cfg, self.dt
        Hq, Hkv, D, eps = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"], cfg["rms_norm_eps"]
        ids = np.asarray(ids, np.uint32)
        T = len(ids)
        Linear.marlin_step = dense_prefill_rows(cfg, T)
        try:
            return self._forward(ids, T, positions, slot_mapping, block_tables, context_lens, cu_q)
        finally:
            Linear.marlin_step = False

    def _forward(self, ids, T, positions, slot_mapping, block_tables, context_lens, cu_q):
        cfg, dt = self.cfg, self.dt
        Hq, Hkv, D, eps = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"], cfg["rms_norm_eps"]
        h = orc.embedding(ids, self.embed, dt)
        for li, L in enumerate(self.layers):
            dmask = deferred_norm_mask(cfg, T, self.tp, li)
            if dmask & 1:  # the engine's 1..4-row launch: x staged as round(x * g), rstd on the f32 dot products (gemv_q4s.cuh)
                x, rs = orc.rms_norm_deferred(h, L["attn_norm"], eps, dt)
            else:
                x, rs = orc.rms_norm(h, L["attn_norm"], eps, dt), None
            q = L["q"](x, row_scale=rs).reshape(T, Hq, D)
            k = L["k"](x, row_scale=rs).reshape(T, Hkv, D)
            v = L["v"](x, row_scale=rs).reshape(T, Hkv, D)
            q, k = self.qk_norm(L, q, k)
            q = orc.rope(q, self.cos, self.sin, positions, False, dt, dt)
            k = orc.rope(k, self.cos, self.sin, positions, False, dt, dt)
            orc.reshape_and_cache(k, v, self.kc[li], self.vc[li], slot_mapping, self.BS, dt, self.kv_dt)
            a = orc.paged_attention(q, self.kc[li], self.vc[li], block_tables, context_lens, cu_q, Hkv, self.BS, D ** -0.5, dt, kv_dt=self.kv_dt,
                                    sliding_window=int(cfg.get("sliding_window") or 0))  # llama.rs:46,284
            h = self._row_parallel(L["o"], a.reshape(T, Hq * D), h)           # attn_output + residual
            if dmask & 2:
                x, rs = orc.rms_norm_deferred(h, L["ffn_norm"], eps, dt)
            else:
                x, rs = orc.rms_norm(h, L["ffn_norm"], eps, dt), None
            act = orc.silu_mul(L["gate"](x, row_scale=rs), L["up"](x, row_scale=rs), dt)
            h = self._row_parallel(L["down"], act, h)                         # residual + mlp_output
        if cu_q is not None:  # last token of each sequence (llama.rs:306-310)
            rows = np.asarray(cu_q[1:], np.int64) - 1
            h = np.ascontiguousarray(h[rows])
        x = orc.rms_norm(h, self.final_norm, eps, dt)
        return orc.dense_gemm(x, self.lm_head, None, dt, F32)


def make_random_checkpoint(cfg, seed=0):
    """small random model in checkpoint format (HF names); returns dict name -> numpy array."""
    r = np.random.default_rng(seed)
    dt = cfg["dtype"]
    H, I, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_layers"]
    Hq, Hkv, D = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"]
    qm, gs = cfg.get("quant_method"), cfg.get("group_size", 128)
    w = {}

    def f(shape, std=0.02, mean=0.0):
        return orc.to_dt((mean + std * r.standard_normal(shape)).astype(np.float32), dt)

    def lin(prefix, K, N, bias=False, std=None):
        if qm in ("gptq", "awq"):
            g = gs if gs > 0 else K
            idx = r.integers(0, 16, size=(K, N), dtype=np.uint8)
            # scale so that dequantised weights have std ~ 1/sqrt(K) (keeps activations O(1))
            s = (1.0 / np.sqrt(K) / 4.6) * (0.5 + r.random((K // g, N)))
            w[prefix + ".scales"] = orc.to_dt(s.astype(np.float32), dt)
            if qm == "awq":
                zeros = r.integers(0, 16, size=(K // g, N), dtype=np.uint8)
                w[prefix + ".qweight"] = orc.awq_pack(idx)
                w[prefix + ".qzeros"] = orc.awq_pack(zeros)
            else:
                w[prefix + ".qweight"] = orc.gptq_pack(idx)
                w[prefix + ".qzeros"] = np.full((K // g, N // 8), 0x77777777, np.uint32)
                w[prefix + ".g_idx"] = (np.arange(K) // g).astype(np.uint32)
        else:
            w[prefix + ".weight"] = f((N, K), std or 1.0 / np.sqrt(K))
        if bias:
            w[prefix + ".bias"] = f((N,), 0.1)

    w["model.embed_tokens.weight"] = f((V, H), 1.0)
    w["model.norm.weight"] = f((H,), 0.05, 1.0)
    if not cfg.get("tie_word_embeddings"):
        w["lm_head.weight"] = f((V, H), 1.0 / np.sqrt(H))
    for i in range(L):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = f((H,), 0.05, 1.0)
        w[p + "post_attention_layernorm.weight"] = f((H,), 0.05, 1.0)
        b = bool(cfg.get("attention_bias"))
        lin(p + "self_attn.q_proj", H, Hq * D, b)
        lin(p + "self_attn.k_proj", H, Hkv * D, b)
        lin(p + "self_attn.v_proj", H, Hkv * D, b)
        if cfg.get("qk_norm") or cfg.get("arch") == "qwen3":  # Qwen3-style q_norm / k_norm: per head ("head", the default) or full row ("full")
            full = cfg.get("qk_norm") in ("full", 2)
            w[p + "self_attn.q_norm.weight"] = f((Hq * D if full else D,), 0.1, 1.0)
            w[p + "self_attn.k_norm.weight"] = f((Hkv * D if full else D,), 0.1, 1.0)
        lin(p + "self_attn.o_proj", Hq * D, H)
        lin(p + "mlp.gate_proj", H, I)
        lin(p + "mlp.up_proj", H, I)
        lin(p + "mlp.down_proj", I, H)
    return w
