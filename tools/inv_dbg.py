#!/usr/bin/env python3
"""Per-row view of one decode case of tools/invariance_sweep.py (same random stream): which sequences deviate, by how much."""
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.test_gpu_engine import BF16, prefill_inputs, simple_tables  # noqa: E402
from tools.repro_sweep import CFGS  # noqa: E402
from vllm_rs_amd.engine import Engine  # noqa: E402

name, want_b = sys.argv[1], int(sys.argv[2])
cfg = CFGS[name]
mp, dt, V = cfg["max_position_embeddings"], cfg["dtype"], cfg["vocab_size"]
eng = Engine(cfg, max_num_seqs=32, max_model_len=mp, num_gpu_blocks=256, use_graph=False, seed=7, fp8_kvcache=False).init_synthetic()
r = np.random.default_rng(9)
bits = 8 if dt == BF16 else 11
for B in (2, 3, 4, 5, 8, 9, 16, 17, 32):
    hi = min(200, mp - 16)
    prompts = [r.integers(0, V, size=int(n)).tolist() for n in r.integers(5, hi, size=B)]
    bt = simple_tables([len(p) + 8 for p in prompts])
    ids = None
    if B == want_b:
        for b0 in range(0, B, 4):
            pi = prefill_inputs(prompts[b0:b0 + 4], bt[b0:b0 + 4])
            eng.forward_raw(pi[0], pi[1], pi[2], bt[b0:b0 + 4], pi[3], pi[4])
    ids = r.integers(0, V, size=B).astype(np.uint32)
    if B != want_b:
        continue
    pos = np.array([len(p) for p in prompts], np.int64)
    slots = np.array([int(bt[b, pos[b] // 64]) * 64 + pos[b] % 64 for b in range(B)], np.int64)
    ctx = (pos + 1).astype(np.uint32)
    tog = [eng.forward_raw(ids, pos, slots, bt, ctx, None) for _ in range(2)]
    alone = np.concatenate([eng.forward_raw(ids[b:b + 1], pos[b:b + 1], slots[b:b + 1], bt[b:b + 1], ctx[b:b + 1], None) for b in range(B)])
    alone2 = np.concatenate([eng.forward_raw(ids[b:b + 1], pos[b:b + 1], slots[b:b + 1], bt[b:b + 1], ctx[b:b + 1], None) for b in range(B)])
    print("together twice identical:", np.array_equal(tog[0], tog[1]), " alone twice identical:", np.array_equal(alone, alone2), flush=True)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(alone).max(axis=-1, keepdims=True), 1.0))) - (bits - 1))
    d = np.abs(tog[0] - alone) / ulp
    for b in range(B):
        print(f"row {b:2d} ctx {int(ctx[b]):4d} bt0 {int(bt[b,0]):3d}  worst {d[b].max():7.1f} ulp  >2ulp: {(d[b] > 2).mean():.4f}  max|logit| {np.abs(alone[b]).max():.2f}", flush=True)
    bad = int(np.argmax(d.max(axis=1)))
    others = [b for b in range(B) if b != bad]
    sub = np.array(others[:4] + [bad])
    eb = 8 if dt == BF16 else 5

    def f32(u16):
        if dt == BF16:
            return (u16.astype(np.uint32) << 16).view(np.float32)
        return u16.view(np.float16).astype(np.float32)

    for layer in range(cfg["num_layers"]):
        eng.tp_snapshots(True, layer)
        eng.forward_raw(ids[bad:bad + 1], pos[bad:bad + 1], slots[bad:bad + 1], bt[bad:bad + 1], ctx[bad:bad + 1], None)
        one = eng.read_tp_snapshots()
        eng.forward_raw(ids[sub], pos[sub], slots[sub], bt[sub], ctx[sub], None)
        five = eng.read_tp_snapshots()
        for st in eng.TP_STAGES:
            if st not in one or st not in five:
                continue
            a = f32(one[st]); w = a.size
            b = f32(five[st]).reshape(len(sub), -1)[-1][:w]
            u = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 1e-30))) - (bits - 1))
            k = int(np.argmax(np.abs(a - b) / u))
            print(f"layer {layer} {st:14s} n {w:6d}  rms {np.sqrt((a * a).mean()):10.4g} max|x| {np.abs(a).max():10.4g}  differing {float((a != b).mean()):.4f}  "
                  f"worst {float((np.abs(a - b) / u).max()):8.1f} own-ulp at [{k}] one {a[k]:.6g} five {b[k]:.6g}   max|d| {np.abs(a - b).max():.4g}", flush=True)
    # the cache of the deviating row: written by a 4-prompt prefill step above; rewrite it by a step of its own and decode again
    def attn_rms(layer):
        eng.tp_snapshots(True, layer)
        lg = eng.forward_raw(ids[bad:bad + 1], pos[bad:bad + 1], slots[bad:bad + 1], bt[bad:bad + 1], ctx[bad:bad + 1], None)
        a = f32(eng.read_tp_snapshots()["attn"])
        return lg, float(np.sqrt((a * a).mean())), float(np.abs(a).max())
    last = cfg["num_layers"] - 1
    lg0, r0, m0 = attn_rms(last)
    pi = prefill_inputs([prompts[bad]], bt[bad:bad + 1])
    eng.forward_raw(pi[0], pi[1], pi[2], bt[bad:bad + 1], pi[3], pi[4])
    lg1, r1, m1 = attn_rms(last)
    print(f"row {bad}: cache from the 4-prompt prefill step: layer {last} attn rms {r0:.3f} max {m0:.3f};  from a prefill step of its own: rms {r1:.3f} max {m1:.3f};  "
          f"logits {float((np.abs(lg1 - lg0) / ulp[bad]).max()):.1f} ulp apart", flush=True)
    g0 = (bad // 4) * 4
    grp = list(range(g0, min(g0 + 4, B)))
    print("its prefill group:", [(b, len(prompts[b])) for b in grp], flush=True)
    eng.tp_snapshots(False)
eng.close()
