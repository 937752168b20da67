"""debug aid (round 5): the ready-made-operands path of kernel W at the Llama-3-8B widths — per stage of layer 0 against the oracle's stage
values (tests/tp_stages.py with one rank), for a few batch sizes"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import numpy as np
sys.path.insert(0, '.')
from oracle import model as om, oracle as orc
from tests.test_gpu_engine import small_cfg, simple_tables, prefill_inputs, build
from tests.tp_stages import oracle_stages, STAGE_ORDER
from vllm_rs_amd import _lib
L = _lib.load()
om.ENGINE_RULE = L
cfg = small_cfg(hidden_size=4096, intermediate_size=14336, num_layers=2, num_heads=32, num_kv_heads=8, head_dim=128, vocab_size=2048, rope_theta=500000.0, max_position_embeddings=2048)
for B in [int(a) for a in sys.argv[1:]] or [24, 31, 32]:
    r = np.random.default_rng(B)
    lens = [int(n) for n in r.integers(3, 150, size=B)]
    lens[0] = 300
    nblk = sum((n + 8 + 63) // 64 for n in lens) + 2
    eng, oracle = build(cfg, seed=5, max_num_seqs=32, num_gpu_blocks=nblk)
    ost = om.OracleModel(cfg, om.make_random_checkpoint(cfg, 5), num_blocks=nblk)
    eng.tp_snapshots(True)
    prompts = [r.integers(0, cfg["vocab_size"], size=n).tolist() for n in lens]
    bt = simple_tables([len(p) + 8 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    ref = oracle.forward(ids, pos, slots, bt, ctx, cu)
    ost.forward(ids, pos, slots, bt, ctx, cu)
    eng.forward_raw(ids, pos, slots, bt, ctx, cu)
    seqs = [list(p) + [int(t)] for p, t in zip(prompts, orc.argmax_f32(ref))]
    ids = np.array([s[-1] for s in seqs], np.uint32); pos = np.array([len(s) - 1 for s in seqs], np.int64)
    slots = np.array([int(bt[b, (len(s) - 1) // 64]) * 64 + (len(s) - 1) % 64 for b, s in enumerate(seqs)], np.int64); ctx = np.array([len(s) for s in seqs], np.uint32)
    import copy
    oracle_before = copy.deepcopy(oracle)
    got = eng.forward_raw(ids, pos, slots, bt, ctx, None)
    print(flush=True); print(f"== B={B}: engine mask layer0/1 = {eng.norm_deferred(B, 0)}/{eng.norm_deferred(B, 1)}, mirror = {om.deferred_norm_mask(cfg, B, 1, 0)}/{om.deferred_norm_mask(cfg, B, 1, 1)}")
    sts = oracle_stages(ost, ids, pos, slots, bt, ctx, None)
    ref = oracle.forward(ids, pos, slots, bt, ctx, None)
    alone = np.concatenate([eng.forward_raw(ids[b:b + 1], pos[b:b + 1], slots[b:b + 1], bt[b:b + 1], ctx[b:b + 1], None) for b in range(B)])
    ulp = 2.0 ** (np.floor(np.log2(np.abs(ref).max(axis=-1, keepdims=True))) - 7)
    print(f"   contexts {ctx.tolist()[:12]}")
    print(f"   engine step of {B} vs the same rows one at a time (kernel E): per row {np.round((np.abs(got - alone) / ulp).max(axis=1), 1).tolist()[:12]} ulp")
    print(f"   one at a time vs the oracle's step of {B}: per row {np.round((np.abs(alone - ref) / ulp).max(axis=1), 1).tolist()[:12]} ulp")
    for layer in range(1):
        eng.tp_snapshots(True, layer)
        eng.forward_raw(ids, pos, slots, bt, ctx, None)
        snaps = eng.read_tp_snapshots()
        st = sts[layer]
        for n in STAGE_ORDER:
            if n in snaps and n in st and snaps[n].size == st[n].size:
                a, b = orc.from_dt(snaps[n], 0).astype(np.float64), orc.from_dt(st[n], 0).astype(np.float64)
                ulp = 2.0 ** (np.floor(np.log2(max(np.abs(b).max(), 1e-30))) - 7)
                d = (np.abs(a - b) / ulp).reshape(B, -1)
                print(f"   layer {layer} stage {n:12s}: max {d.max():6.2f} ulp of the stage scale (|x| max {np.abs(b).max():.3g}), {100 * (d > 0).mean():5.1f}% differ; per row {np.round(d.max(axis=1), 1).tolist()[:8]}")
    d = np.abs(got - ref); ulp = 2.0 ** (np.floor(np.log2(np.abs(ref).max(axis=-1, keepdims=True))) - 7)
    print(f"   logits: max {float((d / ulp).max()):.2f} ulp; per-row max {np.round((d / ulp).max(axis=1), 1).tolist()}", flush=True)
    # which per-layer norm orders does the engine's output agree with?  (oracle re-run on copies of the cache state of this step)
    import copy
    for name, masks in (("mirror", None), ("all reference", {0: 0, 1: 0}), ("l0 gate/up only", {0: 2, 1: 0}), ("l0 gate/up, l1 both", {0: 2, 1: 3}), ("l1 q/k/v reference", {0: 2, 1: 2}),
                        ("l1 gate/up reference", {0: 2, 1: 1})):
        o2 = copy.deepcopy(oracle_before)
        orig = om.deferred_norm_mask
        if masks is not None:
            om.deferred_norm_mask = lambda cfg_, T_, tp_=1, layer_=1, m=masks: m.get(layer_, 0)
        r2 = o2.forward(ids, pos, slots, bt, ctx, None)
        om.deferred_norm_mask = orig
        d2 = np.abs(got - r2) / ulp
        print(f"   vs oracle [{name:22s}]: max {float(d2.max()):.2f} ulp; per row {np.round(d2.max(axis=1), 1).tolist()[:12]}; mean |d| {float(np.abs(got - r2).mean()):.5f}", flush=True)
    eng.close()
