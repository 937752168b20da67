"""Per-rank, per-stage oracle of ONE tensor-parallel decoder layer (test infrastructure; oracle/model.py's arithmetic cut open at the
points where the engine's TP snapshots are taken, `Engine.TP_STAGES`): which rank's which stage is the FIRST to deviate from the
oracle tells a wrong GEMM partial (stage o_partial / down_partial of ONE rank) from a wrong exchange (h_after_* wrong with every
partial right) from a wrong input (q / k / v / attn).  Follows src/models/llama.rs:107-131 and distributed.rs:438-455,498-538."""
import numpy as np

from oracle import oracle as orc


def oracle_stages(o, ids, positions, slot_mapping, block_tables, context_lens, cu_q):
    """o: oracle.model.OracleModel with tp_world = W (its KV cache is written as a forward would) -> list over ranks of
    dict stage -> uint16 array (model dtype bit patterns), plus the logits"""
    cfg, dt, W = o.cfg, o.dt, o.tp
    Hq, Hkv, D, eps = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"], cfg["rms_norm_eps"]
    ids = np.asarray(ids, np.uint32)
    T = len(ids)
    L = o.layers[0]
    from oracle.model import deferred_norm_mask
    dmask = deferred_norm_mask(cfg, T, W, 0)  # (layer 0) the order of the engine's 1..4-row launches (oracle/model.py ENGINE_RULE)
    h0 = orc.embedding(ids, o.embed, dt)
    x, rs = orc.rms_norm_deferred(h0, L["attn_norm"], eps, dt) if dmask & 1 else (orc.rms_norm(h0, L["attn_norm"], eps, dt), None)
    q0 = L["q"](x, row_scale=rs).reshape(T, Hq, D)
    k0 = L["k"](x, row_scale=rs).reshape(T, Hkv, D)
    v0 = L["v"](x, row_scale=rs).reshape(T, Hkv, D)
    qn, kn = o.qk_norm(L, q0, k0)
    q = orc.rope(qn, o.cos, o.sin, positions, False, dt, dt)
    k = orc.rope(kn, o.cos, o.sin, positions, False, dt, dt)
    orc.reshape_and_cache(k, v0, o.kc[0], o.vc[0], slot_mapping, o.BS, dt, o.kv_dt)
    a = orc.paged_attention(q, o.kc[0], o.vc[0], block_tables, context_lens, cu_q, Hkv, o.BS, D ** -0.5, dt, kv_dt=o.kv_dt).reshape(T, Hq * D)
    K = a.shape[1]
    po = [L["o"].partial(a, r * K // W, (r + 1) * K // W) for r in range(W)]
    h1 = o._row_parallel(L["o"], a, h0)
    x, rs = orc.rms_norm_deferred(h1, L["ffn_norm"], eps, dt) if dmask & 2 else (orc.rms_norm(h1, L["ffn_norm"], eps, dt), None)
    act = orc.silu_mul(L["gate"](x, row_scale=rs), L["up"](x, row_scale=rs), dt)
    KI = act.shape[1]
    pd = [L["down"].partial(act, r * KI // W, (r + 1) * KI // W) for r in range(W)]
    h2 = o._row_parallel(L["down"], act, h1)
    hq, hkv = Hq // W, max(1, Hkv // W)
    out = []
    for r in range(W):
        kvr = r * Hkv // W if Hkv >= W else r // (W // Hkv)  # kv_head_shard (distributed.rs:498-538)
        out.append(dict(q=q0[:, r * hq:(r + 1) * hq], k=k0[:, kvr * 1:kvr * 1 + hkv] if Hkv < W else k0[:, r * hkv:(r + 1) * hkv],
                        v=v0[:, kvr * 1:kvr * 1 + hkv] if Hkv < W else v0[:, r * hkv:(r + 1) * hkv],
                        attn=a[:, r * K // W:(r + 1) * K // W], o_partial=po[r], h_after_o=h1,
                        act=act[:, r * KI // W:(r + 1) * KI // W], down_partial=pd[r], h_after_down=h2))
    return [{n: np.ascontiguousarray(v).reshape(-1) for n, v in st.items()} for st in out]


STAGE_ORDER = ("q", "k", "v", "attn", "o_partial", "h_after_o", "act", "down_partial", "h_after_down")


def first_deviation(got_ranks, ref_ranks, dt, max_ulps=8.0, common_ulps=2.0, common_frac=0.005):
    """-> text naming, per rank, the first stage that leaves the oracle, or '' when every stage agrees.  A stage agrees when no value
    is off by more than `max_ulps` storage ulps of the stage's own magnitude and at most `common_frac` of its values by more than
    `common_ulps`: single flipped roundings travel (a 1-ulp flip of gate or up moves SiLU(gate)*up by up to ~3 ulp, seen on
    hardware: 1 of 7168 values), a wrong tile / stale slab / missing K slice moves many values by far more."""
    bits = 8 if dt == 0 else 11
    lines = []
    for r, (g, f) in enumerate(zip(got_ranks, ref_ranks)):
        for n in STAGE_ORDER:
            if n not in g or n not in f or g[n].size != f[n].size:
                continue
            a, b = orc.from_dt(g[n], dt).astype(np.float64), orc.from_dt(f[n], dt).astype(np.float64)
            scale = max(float(np.abs(b).max()), 1e-30)
            ulp = 2.0 ** (np.floor(np.log2(scale)) - (bits - 1))
            d = np.abs(a - b) / ulp
            if float(d.max()) > max_ulps or float((d > common_ulps).mean()) > common_frac:
                bad = np.flatnonzero(d > common_ulps)
                lines.append(f"rank {r}: first deviating stage '{n}': {bad.size} of {d.size} values off by up to {float(d.max()):.1f} ulp "
                             f"(flat indices {bad[:6].tolist()}..{int(bad[-1])})")
                break
    return "\n".join(lines)
