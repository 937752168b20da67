"""Diagnostic: the B = 31 case of tests/test_gpu_x_frag.py at the Llama-3-8B widths with the dense prefill path on / off: per-row
deviation of the prefill logits and of the first decode step from the oracle (which mirrors the engine's weight rounding), and how far
the oracle itself moves between the two weight roundings on the same rows."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from oracle import model as om
from oracle import oracle as orc
from tests.test_gpu_engine import build, prefill_inputs, simple_tables
from tests.test_gpu_x_frag import CFGS, _decode_inputs
from vllm_rs_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 31
cfg = CFGS["llama3_8b_widths"]
lib = _lib.load()
om.ENGINE_RULE = lib


def ulps(got, ref):
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1.0))) - 7)
    return (np.abs(got - ref) / ulp).max(axis=-1)


for min_rows in (0, 1024):
    lib.vra_debug_set_dense_prefill_min_rows(min_rows)
    r = np.random.default_rng(B)
    lens = [int(n) for n in r.integers(3, 150, size=B)]
    lens[0] = 300
    nblk = sum((n + 8 + 63) // 64 for n in lens) + 2
    eng, oracle = build(cfg, seed=5, max_num_seqs=32, num_gpu_blocks=nblk)
    prompts = [r.integers(0, cfg["vocab_size"], size=n).tolist() for n in lens]
    bt = simple_tables([len(p) + 8 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    print(f"== dense from {min_rows} rows; prefill of {len(ids)} tokens; oracle marlin step: {om.dense_prefill_rows(cfg, len(ids))}", flush=True)
    ref = oracle.forward(ids, pos, slots, bt, ctx, cu)
    got = eng.forward_raw(ids, pos, slots, bt, ctx, cu)
    p = ulps(got, ref)
    print("prefill logits: per-row ulp max %.2f mean %.2f; rows > 4: %s" % (p.max(), p.mean(), np.flatnonzero(p > 4).tolist()), flush=True)
    seqs = [list(q) + [int(t)] for q, t in zip(prompts, orc.argmax_f32(ref))]
    ids, pos, slots, ctx = _decode_inputs(seqs, bt)
    lib.vra_debug_set_x_frag(0)
    rows = eng.forward_raw(ids, pos, slots, bt, ctx, None)
    ref_rows = oracle.forward(ids, pos, slots, bt, ctx, None)
    d = ulps(rows, ref_rows)
    print("decode step 0 : per-row ulp max %.2f mean %.2f; rows > 4: %s" % (d.max(), d.mean(), [(int(i), round(float(d[i]), 1)) for i in np.flatnonzero(d > 4)]), flush=True)
    if min_rows:
        # the oracle's own movement on the decode step when its PREFILL ran with the other weight rounding (exact product)
        import copy
        lib.vra_debug_set_dense_prefill_min_rows(0)
        o2 = om.OracleModel(cfg, oracle.w, num_blocks=nblk)
        i2, p2, s2, c2, cu2 = prefill_inputs(prompts, bt)
        o2.forward(i2, p2, s2, bt, c2, cu2)
        alt = o2.forward(ids, pos, slots, bt, ctx, None)
        m = ulps(alt, ref_rows)
        print("oracle(exact prefill) vs oracle(marlin prefill), decode step 0: max %.2f; row 8: %.2f; engine vs exact-prefill oracle row 8: %.2f" % (m.max(), m[8], ulps(rows, alt)[8]), flush=True)
        print("rows where the oracle moves > 4 ulp:", [(int(i), round(float(m[i]), 1)) for i in np.flatnonzero(m > 4)])
    lib.vra_debug_set_x_frag(1)
    eng.close()
