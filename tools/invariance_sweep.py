#!/usr/bin/env python3
"""Batch / step-shape invariance sweep (engine only, no oracle): the logits of a sequence must not depend — beyond a few storage
ulps, different kernels sum in different orders — on what else is in the step or on how the prompt was cut into steps:

  * a decode step of B sequences            vs  the same sequences decoded one at a time (kernel E path, the path with the
                                                 tightest oracle parity);
  * a prompt prefilled in ONE step           vs  prefilled up to its last token and finished by a one-token decode step,
                                             vs  prefilled in two chunks (the second attends to the cached first: paged prefix);
  * the same at 16-bit and FP8 KV.

A kernel that is wrong for one (M, K, N) combination shows up as tens of ulps on the rows it touches.

    python tools/invariance_sweep.py [config ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.test_gpu_engine import BF16, prefill_inputs, simple_tables  # noqa: E402
from tools.repro_sweep import CFGS  # noqa: E402
from vllm_rs_amd.engine import Engine  # noqa: E402

MAX_ULPS = 6.0


def ulps(got, ref, dt):
    bits = 8 if dt == BF16 else 11
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1.0))) - (bits - 1))
    return float((np.abs(got - ref) / ulp).max())


def main(names=None):
    names = names or sys.argv[1:] or list(CFGS)
    bad, worst = 0, 0.0
    for name in names:
        cfg = CFGS[name]
        mp, dt, V = cfg["max_position_embeddings"], cfg["dtype"], cfg["vocab_size"]
        for fp8 in (False, True):
            eng = Engine(cfg, max_num_seqs=32, max_model_len=mp, num_gpu_blocks=256, use_graph=False, seed=7, fp8_kvcache=fp8).init_synthetic()
            r = np.random.default_rng(9)
            w_cfg = 0.0
            # ---- decode: B at once vs one at a time
            for B in (2, 3, 4, 5, 8, 9, 16, 17, 32):
                hi = min(200, mp - 16)
                prompts = [r.integers(0, V, size=int(n)).tolist() for n in r.integers(5, hi, size=B)]
                bt = simple_tables([len(p) + 8 for p in prompts])
                for b0 in range(0, B, 4):
                    pi = prefill_inputs(prompts[b0:b0 + 4], bt[b0:b0 + 4])
                    eng.forward_raw(pi[0], pi[1], pi[2], bt[b0:b0 + 4], pi[3], pi[4])
                ids = r.integers(0, V, size=B).astype(np.uint32)
                pos = np.array([len(p) for p in prompts], np.int64)
                slots = np.array([int(bt[b, pos[b] // 64]) * 64 + pos[b] % 64 for b in range(B)], np.int64)
                ctx = (pos + 1).astype(np.uint32)
                together = eng.forward_raw(ids, pos, slots, bt, ctx, None)
                alone = np.concatenate([eng.forward_raw(ids[b:b + 1], pos[b:b + 1], slots[b:b + 1], bt[b:b + 1], ctx[b:b + 1], None) for b in range(B)])
                u = ulps(together, alone, dt)
                w_cfg = max(w_cfg, u)
                if u > MAX_ULPS or not np.isfinite(together).all():
                    bad += 1
                    print(f"VARIANT {name} fp8={fp8} decode B={B}: {u:.1f} ulp from the one-at-a-time logits", flush=True)
            # ---- prefill: one step vs (N-1 tokens + a decode step) vs two chunks
            for n in (2, 9, 33, 64, 65, 130, 257, 700):
                if n + 8 > mp:
                    continue
                p = r.integers(0, V, size=n).tolist()
                bt = simple_tables([n + 8])
                ids, pos, slots, ctx, cu = prefill_inputs([p], bt)
                whole = eng.forward_raw(ids, pos, slots, bt, ctx, cu)
                i2, p2, s2, c2, cu2 = prefill_inputs([p[:-1]], bt)
                eng.forward_raw(i2, p2, s2, bt, c2, cu2)
                dec = eng.forward_raw(np.array(p[-1:], np.uint32), np.array([n - 1], np.int64), np.array([int(bt[0, (n - 1) // 64]) * 64 + (n - 1) % 64], np.int64),
                                      bt, np.array([n], np.uint32), None)
                cut = max(1, n // 2)
                i3, p3, s3, c3, cu3 = prefill_inputs([p[:cut]], bt)
                eng.forward_raw(i3, p3, s3, bt, c3, cu3)
                i4, p4, s4, c4, cu4 = prefill_inputs([p], bt, cached=[cut])
                two = eng.forward_raw(i4, p4, s4, bt, c4, cu4)
                for what, got in (("prefill(n-1) + decode", dec), (f"two chunks ({cut} + {n - cut})", two)):
                    u = ulps(got, whole, dt)
                    w_cfg = max(w_cfg, u)
                    if u > MAX_ULPS or not np.isfinite(got).all():
                        bad += 1
                        print(f"VARIANT {name} fp8={fp8} prompt {n}: {what} is {u:.1f} ulp from the one-step prefill", flush=True)
            eng.close()
            worst = max(worst, w_cfg)
            print(f"{name} fp8={fp8}: worst {w_cfg:.2f} ulp", flush=True)
    print(f"variant cases: {bad} (worst {worst:.2f} ulp, limit {MAX_ULPS})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
