"""float64 'truth' of the forward pass for the tolerance tests (TEST INFRASTRUCTURE): the same network as oracle/model.py
(llama.rs:107-131,269-321; attention.rs:648-839; mlp.rs:451-469) evaluated in float64 with NO intermediate rounding —
weights and embeddings are the stored 16-bit / int4 values, everything downstream (norms, projections, rotary tables and
angles, softmax, residual stream, logits) stays in float64.  One sequence, contiguous cache.

What it is for: the oracle and the GPU both round to the storage dtype at every op boundary of the reference; each is one
*valid* realisation of that arithmetic, and their distance to each other says little on its own.  Their distances to this
unrounded truth can be compared: the GPU path is acceptable when it is not (materially) further from the truth than the
oracle is (tests/test_gpu_tolerance.py)."""
import numpy as np

from oracle import oracle as orc


def _deq(w, prefix, cfg):
    """float64 weight [K, N] of one linear layer + bias"""
    dt = cfg["dtype"]
    bias = w.get(prefix + ".bias")
    bias = None if bias is None else orc.from_dt(bias, dt).astype(np.float64)
    if prefix + ".qweight" in w:
        sc = orc.from_dt(w[prefix + ".scales"], dt).astype(np.float64)
        G, N = sc.shape
        if cfg["quant_method"] == "awq":
            K = w[prefix + ".qweight"].shape[0]
            idx = orc.awq_unpack(w[prefix + ".qweight"], K, N).astype(np.float64)
            z = orc.awq_unpack_zeros(w[prefix + ".qzeros"], G, N).astype(np.float64)
        else:
            K = w[prefix + ".qweight"].shape[0] * 8
            idx = orc.gptq_unpack(w[prefix + ".qweight"], K, N).astype(np.float64)
            z = np.full((G, N), 8.0)
        g = K // G
        return (idx - np.repeat(z, g, axis=0)) * np.repeat(sc, g, axis=0), bias
    return orc.from_dt(w[prefix + ".weight"], dt).astype(np.float64).T, bias


class TruthModel:
    def __init__(self, cfg, w):
        self.cfg, dt = cfg, cfg["dtype"]
        f = lambda name: orc.from_dt(w[name], dt).astype(np.float64)
        self.embed, self.final_norm = f("model.embed_tokens.weight"), f("model.norm.weight")
        self.lm_head = f("lm_head.weight") if "lm_head.weight" in w else self.embed
        self.layers = []
        for i in range(cfg["num_layers"]):
            p = f"model.layers.{i}."
            L = dict(attn_norm=f(p + "input_layernorm.weight"), ffn_norm=f(p + "post_attention_layernorm.weight"))
            for n, q in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"),
                         ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"), ("down", "mlp.down_proj")):
                L[n] = _deq(w, p + q, cfg)
            self.layers.append(L)
        D = cfg["head_dim"]
        self.inv_freq = 1.0 / cfg["rope_theta"] ** (np.arange(0, D, 2, dtype=np.float64) / D)
        self.k, self.v = [[] for _ in self.layers], [[] for _ in self.layers]

    def _norm(self, x, wt):
        return x / np.sqrt((x * x).mean(-1, keepdims=True) + self.cfg["rms_norm_eps"]) * wt

    def _rope(self, x, pos):  # NeoX pairing: (i, i + D/2)
        ang = pos[:, None].astype(np.float64) * self.inv_freq[None, :]
        c, s = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
        h = x.shape[-1] // 2
        a, b = x[..., :h], x[..., h:]
        return np.concatenate([a * c - b * s, b * c + a * s], -1)

    def forward(self, ids, positions):
        """tokens of ONE sequence at `positions` (continuing the cache); returns float64 logits of the last token"""
        cfg = self.cfg
        Hq, Hkv, D = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"]
        pos = np.asarray(positions, np.int64)
        h = self.embed[np.asarray(ids, np.int64)]
        T = len(pos)
        lin = lambda x, wb: x @ wb[0] + (0.0 if wb[1] is None else wb[1])
        for li, L in enumerate(self.layers):
            x = self._norm(h, L["attn_norm"])
            q = self._rope(lin(x, L["q"]).reshape(T, Hq, D), pos)
            k = self._rope(lin(x, L["k"]).reshape(T, Hkv, D), pos)
            v = lin(x, L["v"]).reshape(T, Hkv, D)
            self.k[li].extend(k)
            self.v[li].extend(v)
            K, V = np.stack(self.k[li]), np.stack(self.v[li])  # [ctx, Hkv, D]
            out = np.empty((T, Hq, D))
            for hq in range(Hq):
                hk = hq // (Hq // Hkv)
                s = q[:, hq] @ K[:, hk].T * D ** -0.5  # [T, ctx]
                ctx = K.shape[0]
                mask = np.arange(ctx)[None, :] > (ctx - T + np.arange(T))[:, None]
                s = np.where(mask, -np.inf, s)
                p = np.exp(s - s.max(-1, keepdims=True))
                out[:, hq] = (p / p.sum(-1, keepdims=True)) @ V[:, hk]
            h = h + lin(out.reshape(T, Hq * D), L["o"])
            x = self._norm(h, L["ffn_norm"])
            g, u = lin(x, L["gate"]), lin(x, L["up"])
            h = h + lin(g / (1.0 + np.exp(-g)) * u, L["down"])
        return self._norm(h[-1:], self.final_norm) @ self.lm_head.T
